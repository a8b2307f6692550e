#!/bin/bash
# Kernel trace + PMC passes (SQ instruction mix, FP64 op counts, stalls) of one workload solved once.
# usage (GPU box): bash tools/profile_pmc.sh C4m r2_c4m
set -e
export TMPDIR=/tmp
wl=$1; tag=$2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
cmd="python $root/tools/dump_matrix.py $wl /tmp/pm_$tag.npz"
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $cmd > $out/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $out/pmc1 -o p -- $cmd > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $out/pmc2 -o p -- $cmd > $out/pmc2.log 2>&1
cd $root
python tools/rocpd_summary.py $(find $out/trace -name "*.db" | head -1) > $out/kernel_stats.csv
python tools/rocpd_pmc.py $(find $out/pmc1 -name "*.db" | head -1) $(find $out/pmc2 -name "*.db" | head -1) > $out/pmc_sq.csv
# which build the counters belong to (bench.py marks a pass of other kernel sources as stale)
python -c "import json, bench; print(json.dumps(dict(kernel_source_digest=bench.kernel_source_digest(), workload='$wl', policy_env=dict(IPC_SE2_POLICY='${IPC_SE2_POLICY:-}', IPC_SE3_POLICY='${IPC_SE3_POLICY:-}', IPC_TERMINATE_EPS='${IPC_TERMINATE_EPS:-}'))))" > $out/pmc_sq.meta.json
find $out -name "*.db" -delete
find $out -name "*.csv" -size +2M -delete
head -12 $out/kernel_stats.csv
