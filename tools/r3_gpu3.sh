#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 ./build/persist_prims_bench 2>&1 | tail -4 > gpurun_out/r3_prims_bench2.txt; cat gpurun_out/r3_prims_bench2.txt
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r3_persist_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_persist_tests.log
tail -5 gpurun_out/r3_persist_tests.log
IPC_PERSIST_PROF=1 timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_prof.json 2> gpurun_out/r3_c1_prof.err
cat gpurun_out/r3_c1_prof.json; grep persist_profile gpurun_out/r3_c1_prof.err
