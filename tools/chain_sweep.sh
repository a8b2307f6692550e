#!/bin/bash
# usage: bash tools/chain_sweep.sh "<env assignments>" ...   -- one faithful prefix run of C4 (2500) and C5 (6000) per argument
for e in "$@"; do
  for w in "C4 2500" "C5 6000"; do
    set -- $w
    echo "[$e] $1 $2: $(env $e IPC_SPEC_STATS=1 timeout 300 python tools/faithful_full.py $1 $2 100000 2>&1 | tail -2 | python -c "
import sys,json
a=json.loads(sys.stdin.readline()); b=json.loads(sys.stdin.readline())['speculation']
print(a['seconds'], a['digest'], a['oracle_prefix']['decisions_differing'], 'launches', b['launches'], 'acc ms', b['accept_ms_per_solve'], 'dev', b['accept_ms_per_solve_on_the_device'], 'rej dev', b['reject_ms_per_solve_on_the_device'], 'timeouts', b['persist_timeouts'])")"
  done
done
