#!/bin/bash
# round 2 GPU call: full -m gpu suite, C2/C4/C5/C3 bench lines, C2 + C4 profiles (trace + SQ PMC + HBM PMC)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $out/gputest.log 2>&1
tail -25 $out/gputest.log
timeout 600 python bench.py --steps 10 --warmup 2 > $out/bench_c2.json 2> $out/bench_c2.log
cat $out/bench_c2.json | head -c 1500; echo
for w in C4 C5 C3; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu > $out/bench_$w.json 2> $out/bench_$w.log
  head -c 600 $out/bench_$w.json; echo
done
timeout 900 bash tools/profile_pmc.sh C2 r2_c2 > $out/prof_c2.log 2>&1
timeout 1500 bash tools/profile_pmc.sh C4 r2_c4 > $out/prof_c4.log 2>&1
tail -3 $out/prof_c2.log $out/prof_c4.log
