#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity_se2.py -x -q -m gpu -k "persistent or dense or speculative or levenberg or fixture or set_only or sharded or final_map or equals" > gpurun_out/r3_persist_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_persist_tests.log
tail -5 gpurun_out/r3_persist_tests.log
IPC_SPEC_WINDOW=1 IPC_PERSIST_PROF=1 timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_prof.json 2> gpurun_out/r3_c1_prof.err
python -c "import json;d=json.load(open('gpurun_out/r3_c1_prof.json'));print('C1 w1 %.2f s'%d['gpu_incremental_s'])"; grep persist_profile gpurun_out/r3_c1_prof.err
for wl in C1 C2; do timeout 600 python tools/incremental_bench.py $wl > gpurun_out/r3_${wl}_inc.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r3_${wl}_inc.json'));print('$wl %.2f s  %.1f checks/s'%(d['gpu_incremental_s'],d['gpu_checks_per_s']))"; done
timeout 600 python bench.py --no-cpu --steps 2 --warmup 1 > gpurun_out/r3_c2_bench_setonly.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r3_c2_bench_setonly.json'));print(d['ms_per_step'], d['set_only_mode'])"
