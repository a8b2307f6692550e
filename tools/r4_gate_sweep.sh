#!/bin/bash
# Faithful C2 run with predicted-accept gating, the recorded verdicts as the prediction (upper bound of what a predictor
# can give).  usage: bash tools/r4_gate_sweep.sh "<queues>:<window>:<behind>:<helpers> ..."
export TMPDIR=/tmp PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r4; mkdir -p $out
cd $root
python - <<'PY'
import numpy as np
d = np.load('tests/golden/c2_incremental_expected.npz')
f = np.zeros(len(d['order']), dtype=np.uint8)
f[d['order']] = d['decision']
open('/tmp/pred_c2.txt', 'w').write(''.join(str(int(x)) for x in f))
PY
for cfg in $1; do
  IFS=: read q w b hp <<< "$cfg"
  echo "== queues $q window $w behind $b helpers $hp"
  GPU_MAX_HW_QUEUES=$q IPC_SPEC_WINDOW=$w IPC_SPEC_BEHIND=$b IPC_PERSIST_HELPERS=$hp IPC_SPEC_PREDICT_FILE=${PRED-/tmp/pred_c2.txt} IPC_SPEC_STATS=1 \
    python tools/lib_incremental.py ipc_amd/libipc_amd.so ${WL:-C2} 2 2>&1 | grep -v amdgpu.ids | cut -c1-420
done
