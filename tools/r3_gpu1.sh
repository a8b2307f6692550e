#!/bin/bash
# round 3, first GPU call: persistent dog-leg vs host-driven solver (bitwise), oracle fixtures, C1 timing in both modes
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r3_persist_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_persist_tests.log
tail -30 gpurun_out/r3_persist_tests.log
for mode in persist host; do
  IPC_CLUSTER_MODE=$mode timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_incremental_$mode.json 2> gpurun_out/r3_c1_incremental_$mode.err
  echo "$mode rc=$?"; cat gpurun_out/r3_c1_incremental_$mode.json
done
