export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r5
mkdir -p $out
cd /tmp
rm -rf $out/trace_c5w8
(cd $root && rocprofv3 --kernel-trace --stats -d $out/trace_c5w8 -o t -- python tools/shard_balance.py C5 8 > $out/c5w8.log 2>&1)
python $root/tools/rocpd_summary.py $(find $out/trace_c5w8 -name "*.db" | head -1) > $out/r5_c5_world8_kernel_stats.csv
rm -rf $out/trace_c5w8
cat $out/c5w8.log | tail -3
cut -c1-200 $out/r5_c5_world8_kernel_stats.csv | head -40
