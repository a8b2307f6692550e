#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r3_persist_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_persist_tests.log
tail -5 gpurun_out/r3_persist_tests.log
IPC_SPEC_WINDOW=1 IPC_PERSIST_PROF=1 timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_prof.json 2> gpurun_out/r3_c1_prof.err
cat gpurun_out/r3_c1_prof.json; grep persist_profile gpurun_out/r3_c1_prof.err
for wl in C1 C2; do timeout 600 python tools/incremental_bench.py $wl > gpurun_out/r3_${wl}_inc.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r3_${wl}_inc.json'));print('$wl %.2f s  %.1f checks/s'%(d['gpu_incremental_s'],d['gpu_checks_per_s']))"; done
