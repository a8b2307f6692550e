#!/bin/bash
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
timeout 600 python tools/uniform_chain_experiment.py ipc_amd/libipc_base.so ipc_amd/libipc_dbg_samerec.so > $out/uce.txt 2>&1
cat $out/uce.txt
timeout 600 python tools/ab_libs.py C2 ipc_amd/libipc_base.so ipc_amd/libipc_amd.so ipc_amd/libipc_nosleep.so > $out/ab_c2.txt 2>&1
cat $out/ab_c2.txt
timeout 600 python tools/ab_libs.py C4m ipc_amd/libipc_base.so ipc_amd/libipc_amd.so > $out/ab_c4m.txt 2>&1
cat $out/ab_c4m.txt
