#!/bin/bash
# Round-6 profile set (same passes as tools/r6_profile.sh, files named r6_*) (GPU box).  usage: bash tools/r6_profile.sh <what> [workload ...]
#   trace   rocprofv3 --kernel-trace --stats of the bench command (matrix steps only: --no-set-only --no-cpu), per workload
#           -> gpurun_out/r6/r6_<w>_kernel_stats.csv  (avg_ns is per launch of a step; the union row is per step)
#   sq      SQ counter passes of ONE solve (tools/dump_matrix.py) -> r6_<w>_pmc_sq.csv + .meta.json (kernel-source digest)
#   hbm     FETCH_SIZE / WRITE_SIZE passes (separate runs) of ONE solve -> r6_pmc_hbm_<W>.csv
#   inc     faithful run over all of C1 / C2: kernel trace, MFMA counters (SQ_INSTS_VALU_MFMA_MOPS_F64,
#           SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES), leader phase clocks
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r6
mkdir -p $out
what=$1; shift
wls=${@:-C2}
cd /tmp
for wl in $wls; do
  w=$(echo $wl | tr A-Z a-z)
  case $what in
  trace)
    rm -rf $out/trace_$w
    rocprofv3 --kernel-trace --stats -d $out/trace_$w -o t -- python $root/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --no-set-only --incremental-candidates 0 > $out/r6_${w}_bench_traced.json 2> $out/trace_$w.log
    python $root/tools/rocpd_summary.py $(find $out/trace_$w -name "*.db" | head -1) > $out/r6_${w}_kernel_stats.csv
    rm -rf $out/trace_$w
    head -6 $out/r6_${w}_kernel_stats.csv | cut -c1-150; tail -1 $out/r6_${w}_kernel_stats.csv
    ;;
  sq)
    cmd="python $root/tools/dump_matrix.py $wl /tmp/pm_$w.npz"
    rm -rf $out/pmc1_$w $out/pmc2_$w
    rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $out/pmc1_$w -o p -- $cmd > $out/pmc1_$w.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $out/pmc2_$w -o p -- $cmd > $out/pmc2_$w.log 2>&1
    python $root/tools/rocpd_pmc.py $(find $out/pmc1_$w -name "*.db" | head -1) $(find $out/pmc2_$w -name "*.db" | head -1) > $out/r6_${w}_pmc_sq.csv
    (cd $root && python -c "import json, bench; print(json.dumps(dict(kernel_source_digest=bench.kernel_source_digest(), workload='$wl', command='tools/dump_matrix.py $wl (one solve)')))") > $out/r6_${w}_pmc_sq.meta.json
    rm -rf $out/pmc1_$w $out/pmc2_$w
    (cd $root && python tools/pmc_table.py $out/r6_${w}_pmc_sq.csv | head -30)
    ;;
  hbm)
    cmd="python $root/tools/dump_matrix.py $wl /tmp/pm_$w.npz"
    rm -rf $out/hbm1_$w $out/hbm2_$w
    rocprofv3 --pmc FETCH_SIZE -d $out/hbm1_$w -o p -- $cmd > $out/hbm1_$w.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $out/hbm2_$w -o p -- $cmd > $out/hbm2_$w.log 2>&1
    python $root/tools/rocpd_pmc.py $(find $out/hbm1_$w -name "*.db" | head -1) $(find $out/hbm2_$w -name "*.db" | head -1) > $out/r6_pmc_hbm_$wl.csv
    rm -rf $out/hbm1_$w $out/hbm2_$w
    head -8 $out/r6_pmc_hbm_$wl.csv | cut -c1-160
    ;;
  inc)
    rm -rf $out/tinc_$w $out/minc_$w
    rocprofv3 --kernel-trace --stats -d $out/tinc_$w -o t -- python $root/tools/incremental_bench.py $wl > $out/r6_${w}_incremental_traced.json 2> $out/tinc_$w.log
    python $root/tools/rocpd_summary.py $(find $out/tinc_$w -name "*.db" | head -1) > $out/r6_${w}_pipeline_kernel_stats.csv
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU -d $out/minc_$w -o p -- python $root/tools/incremental_bench.py $wl > $out/minc_$w.json 2> $out/minc_$w.log
    python $root/tools/rocpd_pmc.py $(find $out/minc_$w -name "*.db" | head -1) > $out/r6_${w}_pipeline_pmc_mfma.csv
    rm -rf $out/tinc_$w $out/minc_$w
    (cd $root && IPC_SPEC_WINDOW=1 IPC_PERSIST_PROF=1 python tools/incremental_bench.py $wl > $out/w1_$w.json 2> $out/w1_$w.err; grep persist_profile $out/w1_$w.err > $out/r6_${wl}_persist_phase_clocks.txt; cat $out/w1_$w.json >> $out/r6_${wl}_persist_phase_clocks.txt)
    (cd $root && IPC_SPEC_STATS=1 python tools/incremental_bench.py $wl > $out/r6_${w}_incremental_pipeline.json 2> $out/stats_$w.err; grep speculation $out/stats_$w.err >> $out/r6_${w}_incremental_pipeline.json)
    head -4 $out/r6_${w}_pipeline_kernel_stats.csv | cut -c1-160; grep -i mfma $out/r6_${w}_pipeline_pmc_mfma.csv | head -5; cut -c1-300 $out/r6_${w}_incremental_pipeline.json
    ;;
  esac
done
