#!/bin/bash
# Round 5: kernel trace and FP64-MFMA counters of a faithful-run prefix that goes through the banded large-cluster kernel.
# usage (GPU box): bash tools/r6_band_profile.sh [workload] [candidates]
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r6; mkdir -p $out
wl=${1:-C4}; n=${2:-900}; w=$(echo $wl | tr A-Z a-z)
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/tband -o t -- python $root/tools/faithful_full.py $wl $n $n > $out/r6_${w}_${n}_traced.json 2> $out/tband.log
python $root/tools/rocpd_summary.py $(find $out/tband -name "*.db" | head -1) > $out/r6_${w}_faithful_${n}_kernel_stats.csv
rm -rf $out/tband
head -8 $out/r6_${w}_faithful_${n}_kernel_stats.csv | cut -c1-220
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU -d $out/mband -o p -- python $root/tools/faithful_full.py $wl $n $n > $out/mband.json 2> $out/mband.log
python $root/tools/rocpd_pmc.py $(find $out/mband -name "*.db" | head -1) > $out/r6_${w}_faithful_${n}_pmc_mfma.csv
rm -rf $out/mband
grep -i "band\|persist" $out/r6_${w}_faithful_${n}_pmc_mfma.csv | cut -c1-200 | head -14
