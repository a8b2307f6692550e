#!/bin/bash
# round-2 closing run on the GPU box: full -m gpu suite, bench lines, kernel traces + SQ PMC passes of C2 and C4, HBM passes of C2
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $out/gputest.log 2>&1
tail -14 $out/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 2 > $out/bench_c2.json 2> $out/bench_c2.log
head -c 400 $out/bench_c2.json; echo
for w in C4 C5 C3; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu > $out/bench_$w.json 2> $out/bench_$w.log
  head -c 300 $out/bench_$w.json; echo
done
timeout 900 bash tools/profile_pmc.sh C2 r2_c2 > $out/prof_c2.log 2>&1
timeout 1500 bash tools/profile_pmc.sh C4 r2_c4 > $out/prof_c4.log 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_write.log
cd $root
python tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $(find $out/pmc_write -name "*.db" | head -1) > $out/pmc_hbm_C2.csv
find $out -name "*.db" -delete
python tools/incremental_bench.py C1 2>&1 | grep -v amdgpu.ids > $out/incr_c1.json; cut -c1-160 $out/incr_c1.json
