#!/bin/bash
# Round-6 measurement set (GPU box): bash tools/r6_measure.sh  -> gpurun_out/r6/
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r6
mkdir -p $out
cd $root
python bench.py > $out/r6_c2_bench.json 2> $out/r6_c2_bench.err
python tools/cpu_full_workload.py C2 > $out/r6_c2_cpu_full.json 2> $out/r6_c2_cpu_full.err
bash tools/r6_profile.sh sq C3 C4 C5 > $out/sq.log 2>&1
bash tools/r6_profile.sh hbm C3 C4 C5 > $out/hbm.log 2>&1
bash tools/r6_profile.sh trace C2 C3 C4 C5 > $out/trace.log 2>&1
cd $root
python bench.py --workload C3 --steps 3 --warmup 1 > $out/r6_c3_bench.json 2> $out/r6_c3_bench.err
python bench.py --workload C4 --steps 2 --warmup 1 --incremental-candidates -1 > $out/r6_c4_bench.json 2> $out/r6_c4_bench.err
python bench.py --workload C5 --steps 5 --warmup 2 --incremental-candidates -1 > $out/r6_c5_bench.json 2> $out/r6_c5_bench.err
python tools/shard_balance.py C2 1 2 4 8 > $out/r6_shard_balance_c2.txt 2>&1
python tools/shard_balance.py C5 1 2 4 8 > $out/r6_shard_balance_c5.txt 2>&1
ls -la $out
