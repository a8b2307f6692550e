#!/bin/bash
# Round-3 profile set (GPU box): kernel trace + stats of the default bench command (C2), SQ / HBM counter passes of a C2
# solve, kernel trace of the faithful incremental run over all C2 candidates, bench lines of C3 / C4 / C5 with CPU legs.
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r3prof
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o c2 -- python $root/bench.py --steps 2 --warmup 1 --no-cpu > $out/bench_trace.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_write.log
rocprofv3 --kernel-trace --stats -d $out/trace_inc -o inc -- python $root/tools/incremental_bench.py C2 > $out/c2_incremental.json 2> $out/trace_inc.log
cd $root
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/r3_c2_kernel_stats.csv
python tools/rocpd_summary.py $(find $out/trace_inc -name "*.db" | head -1) > $out/r3_c2_incremental_kernel_stats.csv
python tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $(find $out/pmc_write -name "*.db" | head -1) > $out/pmc_hbm_C2.csv
find $out -name "*.db" -delete
bash tools/profile_pmc.sh C2 r3_c2 > $out/pmc_c2.log 2>&1
cp gpurun_out/prof_r3_c2/pmc_sq.csv $out/r3_c2_pmc_sq.csv; cp gpurun_out/prof_r3_c2/pmc_sq.meta.json $out/r3_c2_pmc_sq.meta.json
head -8 $out/r3_c2_kernel_stats.csv; head -12 $out/r3_c2_incremental_kernel_stats.csv
python bench.py --workload C5 --steps 3 --warmup 1 > $out/r3_c5_bench.json 2> $out/c5.err; echo "C5 rc=$?"
python bench.py --workload C3 --steps 2 --warmup 1 > $out/r3_c3_bench.json 2> $out/c3.err; echo "C3 rc=$?"
python bench.py --workload C4 --steps 1 --warmup 0 --cpu-seconds 20 > $out/r3_c4_bench.json 2> $out/c4.err; echo "C4 rc=$?"
ls -la $out
