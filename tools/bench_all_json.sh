#!/bin/bash
# bench lines (no CPU leg) of the non-default BASELINE configs into gpurun_out/bench_<W>.json
for w in "$@"; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu 2>/dev/null > gpurun_out/bench_$w.json
  python -c "import json; d=json.load(open('gpurun_out/bench_$w.json')); r=d['roofline']; print('$w', round(d['ms_per_step'],1), r['frac'], r['frac_survey_model'], r['frac_executed'], r['traffic'])"
done
