#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r3_persist_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_persist_tests.log
tail -15 gpurun_out/r3_persist_tests.log
for w in 1 4 8; do
for wl in C1 C2; do
  IPC_SPEC_STATS=1 IPC_SPEC_WINDOW=$w timeout 600 python tools/incremental_bench.py $wl > gpurun_out/r3_${wl}_spec_w$w.json 2> gpurun_out/r3_${wl}_spec_w$w.err
  echo "window=$w $wl rc=$?"; cat gpurun_out/r3_${wl}_spec_w$w.json; grep speculation gpurun_out/r3_${wl}_spec_w$w.err
done; done
