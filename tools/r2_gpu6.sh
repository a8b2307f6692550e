#!/bin/bash
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_incr
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $root/tools/incremental_bench.py C1 > $out/incr.json 2> $out/trace.log
cd $root
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/kernel_stats.csv
find $out -name "*.db" -delete
cat $out/incr.json
cut -c1-90 $out/kernel_stats.csv | head -40
