#!/bin/bash
# Device-side concurrency of the faithful run (tools/pipeline_concurrency.py over a kernel trace), per window / queue count.
# usage: bash tools/r4_concurrency.sh "<queues>:<window> ..."    (GPU box; writes gpurun_out/r4/conc_q<Q>_w<W>.txt)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r4; mkdir -p $out
cd /tmp
for qw in ${1:-16:10 16:16}; do
  q=${qw%%:*}; w=${qw##*:}
  rm -rf /tmp/tc_$q_$w
  GPU_MAX_HW_QUEUES=$q IPC_SPEC_WINDOW=$w IPC_SPEC_STATS=1 rocprofv3 --kernel-trace -d /tmp/tc_${q}_$w -o t -- python $root/tools/lib_incremental.py $root/ipc_amd/libipc_amd.so ${WL:-C2} 1 > $out/conc_q${q}_w$w.txt 2> $out/conc_q${q}_w$w.err
  python $root/tools/pipeline_concurrency.py $(find /tmp/tc_${q}_$w -name "*.db" | head -1) >> $out/conc_q${q}_w$w.txt
  grep -i speculation $out/conc_q${q}_w$w.err | tail -1 >> $out/conc_q${q}_w$w.txt
  echo "== queues $q window $w"; cat $out/conc_q${q}_w$w.txt
  rm -f $out/conc_q${q}_w$w.err
done
