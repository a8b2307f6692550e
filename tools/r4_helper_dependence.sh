#!/bin/bash
# Where does a wide-window run part from the one-at-a-time run?  (per-candidate records, tools/lib_incremental.py IPC_DUMP_RUN)
mkdir -p gpurun_out/r4
python - <<'PY'
import numpy as np
d = np.load('tests/golden/c2_incremental_expected.npz')
f = np.zeros(len(d['order']), dtype=np.uint8); f[d['order']] = d['decision']
open('/tmp/pred_c2.txt', 'w').write(''.join(str(int(x)) for x in f))
PY
IPC_SPEC_WINDOW=1 IPC_DUMP_RUN=/tmp/run_w1 python tools/lib_incremental.py ipc_amd/libipc_amd.so C2 1 2>&1 | grep -v amdgpu
for r in 1 2 3; do
  GPU_MAX_HW_QUEUES=24 IPC_SPEC_WINDOW=${W:-20} IPC_PERSIST_HELPERS=${HP:-8} IPC_SPEC_PREDICT_FILE=${PRED-/tmp/pred_c2.txt} IPC_DUMP_RUN=/tmp/run_r$r python tools/lib_incremental.py ipc_amd/libipc_amd.so C2 ${REPS:-10} 2>&1 | grep -v amdgpu | grep -v "rep.*3a1abba669d2214c"
done
python - <<'PY'
import numpy as np, glob
a=np.load('/tmp/run_w1.rep0.npy')
for f in sorted(glob.glob('/tmp/run_r*.npy')):
    b=np.load(f)
    d=np.where((a[:,1:5]!=b[:,1:5]).any(axis=1))[0]
    if len(d)==0: continue
    d0=np.where(a[:,8]!=b[:,8])[0]
    print(f,'differs from window 1 at',len(d),'positions, first',d[:6],'; chi2_initial differs at',len(d0),'first',d0[:6])
    for i in d0[:6]: print('   pos',i,'w1',a[i].tolist(),'\n          wN',b[i].tolist())
    acc=np.where(a[:,1]==1)[0]
    print('   first accepts at positions',acc[:5],'their (lo,hi)',a[acc[:5],6:8].tolist())
PY
