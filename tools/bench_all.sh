#!/bin/bash
# one bench line per workload (no CPU baseline leg), for the tables in DESIGN.md
for w in "$@"; do
  echo -n "$w: "
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f pairs/s  %.1f ms/step  frac %.4f acc %d' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['accepted']))"
done
