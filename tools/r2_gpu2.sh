#!/bin/bash
# round 2, GPU call 2: phase split of the SE2 kernels (debug build), cell statistics, HBM PMC passes of C2, incremental mode timing
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
timeout 300 python tools/wave_phase_timing.py C2 > $out/phase_c2.txt 2>&1
cat $out/phase_c2.txt
timeout 300 python tools/cell_stats.py C2 > $out/r2_c2_cell_stats.json 2> $out/cs.log
timeout 300 python tools/dump_matrix.py C2 $out/c2_cells.npz > $out/dump_c2.log 2>&1
timeout 600 python tools/cell_stats.py C4 > $out/r2_c4_cell_stats.json 2>> $out/cs.log
timeout 300 python tools/incremental_bench.py C1 > $out/incr_c1.json 2> $out/incr.log
cat $out/incr_c1.json
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o c2 -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $out/pmc_write.log
cd $root
python tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $(find $out/pmc_write -name "*.db" | head -1) > $out/pmc_hbm_C2.csv
find $out -name "*.db" -delete
head -c 600 $out/r2_c2_cell_stats.json
