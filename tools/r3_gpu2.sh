#!/bin/bash
# persistent dog-leg: leader phase clocks on C1, helper count sweep
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for hp in 39 19 9 3; do
  IPC_PERSIST_PROF=1 IPC_PERSIST_HELPERS=$hp timeout 600 python tools/incremental_bench.py C1 > gpurun_out/r3_c1_prof_h$hp.json 2> gpurun_out/r3_c1_prof_h$hp.err
  echo "helpers=$hp rc=$?"; cat gpurun_out/r3_c1_prof_h$hp.json; grep persist_profile gpurun_out/r3_c1_prof_h$hp.err
done
