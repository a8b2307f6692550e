#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" IPC_SPEC_STATS=1 timeout 600 python tools/incremental_bench.py $wl > gpurun_out/r3_spec_$name.json 2> gpurun_out/r3_spec_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('gpurun_out/r3_spec_$name.json'));print('%.2f s  %.1f checks/s'%(d['gpu_incremental_s'],d['gpu_checks_per_s']))") $(grep speculation gpurun_out/r3_spec_$name.err)"
}
run c2_w1 C2 IPC_SPEC_WINDOW=1
run c2_w2 C2 IPC_SPEC_WINDOW=2
run c2_w4 C2 IPC_SPEC_WINDOW=4
run c2_w8 C2 IPC_SPEC_WINDOW=8
run c2_w8_q16 C2 IPC_SPEC_WINDOW=8 GPU_MAX_HW_QUEUES=16
run c1_w1 C1 IPC_SPEC_WINDOW=1
run c1_w2 C1 IPC_SPEC_WINDOW=2
run c1_w4 C1 IPC_SPEC_WINDOW=4
run c1_w8 C1 IPC_SPEC_WINDOW=8
