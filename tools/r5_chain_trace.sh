#!/bin/bash
# Kernel timeline of a faithful prefix run (which kernels and copies lie between two accept solves of the chain).
# usage: bash tools/r5_chain_trace.sh C5 4000
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r5
mkdir -p $out
w=$(echo $1 | tr A-Z a-z)
rm -rf $out/chain_$w
(cd $root && IPC_SPEC_STATS=1 rocprofv3 --kernel-trace --memory-copy-trace -d $out/chain_$w -o t -- python tools/faithful_full.py $1 $2 100000 > $out/chain_$w.log 2>&1)
db=$(find $out/chain_$w -name "*.db" | head -1)
python $root/tools/rocpd_timeline.py $db > /tmp/tl_$w.csv
wc -l /tmp/tl_$w.csv
tail -n 12000 /tmp/tl_$w.csv | gzip > $out/r5_${w}_chain_timeline_tail.csv.gz
python $root/tools/rocpd_summary.py $db | cut -c1-160 | head -14
rm -rf $out/chain_$w
tail -2 $out/chain_$w.log | cut -c1-600
