#!/bin/bash
# Workgroups per team of the split band factorisation (IPC_BAND_TEAM) on prefixes of C4 and C5: seconds and digest.
mkdir -p gpurun_out
for t in 0 3 5 7 9 12; do
  for w in "C4 1500" "C5 6000"; do
    set -- $w
    r=$(IPC_BAND_TEAM=$t timeout 300 python tools/faithful_full.py $1 $2 100000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['seconds'], d['digest'], d['oracle_prefix']['decisions_differing'])")
    echo "team $t $1 $2: $r"
  done
done | tee gpurun_out/band_team_sweep.txt
