#!/bin/bash
# round 2, first GPU call: counter inventory + the round-1 SE3 block kernel on full C4 (reference for the rewrite)
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 -L > $out/r2_counters_all.txt 2>&1
grep -i -E "F64|MFMA|INSTS_VALU|FLOP" $out/r2_counters_all.txt | head -100 > $out/r2_counters_f64.txt
timeout 300 python tools/dump_matrix.py C4m $out/c4m_ref.npz > $out/c4m_ref.log 2>&1
timeout 1500 python tools/dump_matrix.py C4 $out/c4_ref.npz > $out/c4_ref.log 2>&1
tail -2 $out/c4m_ref.log $out/c4_ref.log
wc -l $out/r2_counters_all.txt; head -50 $out/r2_counters_f64.txt
