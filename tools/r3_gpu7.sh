#!/bin/bash
# full gpu suite + default bench + window sweep
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gpu_suite.log
tail -6 gpurun_out/r3_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r3_c2_bench.json 2> gpurun_out/r3_c2_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_c2_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}); print(d.get('incremental'))
PY
run() { name=$1; wl=$2; shift; shift
  env "$@" IPC_SPEC_STATS=1 timeout 600 python tools/incremental_bench.py $wl > gpurun_out/r3_spec_$name.json 2> gpurun_out/r3_spec_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('gpurun_out/r3_spec_$name.json'));print('%.2f s  %.1f checks/s'%(d['gpu_incremental_s'],d['gpu_checks_per_s']))") $(grep speculation gpurun_out/r3_spec_$name.err)"
}
run c2_w6 C2 IPC_SPEC_WINDOW=6
run c2_w8 C2 IPC_SPEC_WINDOW=8
run c2_w12 C2 IPC_SPEC_WINDOW=12
run c2_w16_q32 C2 IPC_SPEC_WINDOW=16 GPU_MAX_HW_QUEUES=32
run c1_w3 C1 IPC_SPEC_WINDOW=3
run c1_w8 C1 IPC_SPEC_WINDOW=8
