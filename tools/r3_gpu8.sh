#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gpu_suite.log
tail -6 gpurun_out/r3_gpu_suite.log
for pol in cyclic cost; do echo "== $pol"; IPC_ROW_BALANCE=$pol timeout 600 python tools/shard_balance.py C2 1 2 4 8 2>/dev/null | tee gpurun_out/r3_shard_balance_$pol.txt; done
