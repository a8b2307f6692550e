export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_c5
rm -rf $out; mkdir -p $out; cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --workload C5 --steps 2 --warmup 1 --no-cpu > $out/bench.json 2> $out/trace.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/kernel_stats.csv
find $out -name "*.db" -delete
cut -c1-60 $out/kernel_stats.csv | paste - <(awk -F, '{print $(NF-12), $(NF-11), $(NF-10)}' $out/kernel_stats.csv) | head -30
