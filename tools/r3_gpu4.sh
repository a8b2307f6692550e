#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
IPC_PERSIST_PROF=1 timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_prof.json 2> gpurun_out/r3_c1_prof.err
cat gpurun_out/r3_c1_prof.json; grep persist_profile gpurun_out/r3_c1_prof.err
