#!/bin/bash
# pipeline parameter sweep (round 3): solves in flight x positions ahead, C2 and C1
for c in C2 C1; do
for w in 6 8 10; do for a in 16 64 256; do
  echo -n "$c W=$w AHEAD=$a: "
  IPC_SPEC_WINDOW=$w IPC_SPEC_AHEAD=$a IPC_SPEC_STATS=1 timeout 300 python tools/incremental_bench.py $c 2>&1 | python -c "
import sys, json
t = sys.stdin.read().splitlines()
s = [json.loads(l) for l in t if l.startswith('{')]
sp = [x for x in s if 'speculation' in x][0]['speculation']; r = [x for x in s if 'workload' in x][0]
print('%.3f s  launches %d discarded %d host_launch %.2f s' % (r['gpu_incremental_s'], sp['launches'], sp['discarded'], sp['host_s_launching']))"
done; done; done
