#!/bin/bash
# Round-3 final profile set (GPU box): rocprofv3 kernel trace + stats of the bench command (C2) and of the faithful run over
# all C2 / C1 candidates through the pipeline; leader phase clocks of one-at-a-time C1 / C2 runs.
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r3final
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o c2 -- python $root/bench.py --steps 2 --warmup 1 --no-cpu > $out/bench_trace.json 2> $out/trace.log
rocprofv3 --kernel-trace --stats -d $out/trace_inc -o inc -- python $root/tools/incremental_bench.py C2 > $out/c2_incremental.json 2> $out/trace_inc.log
rocprofv3 --kernel-trace --stats -d $out/trace_inc1 -o inc -- python $root/tools/incremental_bench.py C1 > $out/c1_incremental.json 2> $out/trace_inc1.log
cd $root
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/r3_c2_kernel_stats_final.csv
python tools/rocpd_summary.py $(find $out/trace_inc -name "*.db" | head -1) > $out/r3_c2_pipeline_kernel_stats.csv
python tools/rocpd_summary.py $(find $out/trace_inc1 -name "*.db" | head -1) > $out/r3_c1_pipeline_kernel_stats.csv
find $out -name "*.db" -delete
for c in C1 C2; do
  IPC_SPEC_WINDOW=1 IPC_PERSIST_PROF=1 python tools/incremental_bench.py $c > $out/${c}_w1.json 2> $out/${c}_w1.err
  grep persist_profile $out/${c}_w1.err > $out/r3_${c}_persist_phase_clocks_final.txt; cat $out/${c}_w1.json >> $out/r3_${c}_persist_phase_clocks_final.txt
done
IPC_SPEC_STATS=1 python tools/incremental_bench.py C2 --cpu > $out/c2_pipeline_cpu.json 2> $out/c2_pipeline_cpu.err
head -5 $out/r3_c2_kernel_stats_final.csv; head -6 $out/r3_c2_pipeline_kernel_stats.csv; head -6 $out/r3_c1_pipeline_kernel_stats.csv
cat $out/c2_incremental.json $out/c1_incremental.json | cut -c1-160
ls $out
