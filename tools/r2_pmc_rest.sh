#!/bin/bash
# SQ PMC passes (executed FP64 flops) of C5 and C3, HBM FETCH/WRITE passes of C4 and C5 -- so that bench.py prints
# frac_executed / traffic for every BASELINE config
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
timeout 600 bash tools/profile_pmc.sh C5 r2_c5 > $out/prof_c5.log 2>&1
timeout 900 bash tools/profile_pmc.sh C3 r2_c3 > $out/prof_c3.log 2>&1
for w in C4 C5; do
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch_$w -o p -- python $root/bench.py --workload $w --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_fetch_$w.log
  rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write_$w -o p -- python $root/bench.py --workload $w --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_write_$w.log
  cd $root
  python tools/rocpd_pmc.py $(find $out/pmc_fetch_$w -name "*.db" | head -1) $(find $out/pmc_write_$w -name "*.db" | head -1) > $out/pmc_hbm_$w.csv
done
find $out -name "*.db" -delete
ls -la $out/prof_r2_c5 $out/prof_r2_c3 $out/pmc_hbm_C4.csv $out/pmc_hbm_C5.csv
