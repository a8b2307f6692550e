#!/bin/bash
# resource usage + instruction mix of one se3_lds_kernel instantiation: tools/se3_regs.sh W M NL
W=$1; M=$2; NL=$3
cd /root/repo/ipc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DTW=$W -DTM=$M -DTNL=$NL -DTOCC=${4:-1} -I. --cuda-device-only -S /tmp/t1.hip -o /tmp/t1_${W}_${M}_${NL}.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Scratch|Spill" | sed 's/.*remark: *//' | tr '\n' ' '
echo " <- W=$W M=$M NL=$NL"
f=/tmp/t1_${W}_${M}_${NL}.s
echo "instrs $(grep -c '^\s*[vsdgb]_' $f)  f64 fma/mul/add $(grep -c 'v_fma_f64\|v_fmac_f64' $f)/$(grep -c 'v_mul_f64' $f)/$(grep -c 'v_add_f64' $f)  accvgpr $(grep -c 'v_accvgpr' $f)  scratch $(grep -c 'scratch_' $f)  rd/wrlane $(grep -c 'v_readlane\|v_writelane' $f)  ds $(grep -c '^\s*ds_' $f)  gload $(grep -c 'global_load' $f)  dpp $(grep -c 'dpp' $f) v_mov $(grep -c 'v_mov_b32' $f)"
