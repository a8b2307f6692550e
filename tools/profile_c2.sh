#!/bin/bash
# Round profile of the default bench command (run on the GPU box through gpurun):
#   kernel trace + stats, then one PMC pass per counter (FETCH_SIZE, WRITE_SIZE), as
#   MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/prof/ (rocpd .db files).
set -e
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $out/bench_trace.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_write.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/kernel_stats.csv
python tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $(find $out/pmc_write -name "*.db" | head -1) > $out/pmc_hbm.csv
python tools/cell_stats.py C2 > $out/cell_stats.json
python bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.log
find $out -name "*.db" -size +30M -delete
ls -la $out
