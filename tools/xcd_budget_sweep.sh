for x in 31 27 23 19; do
  for w in "C4 2500" "C5 6000"; do
    set -- $w
    echo "[xcd_cus $x] $1 $2: $(IPC_SPEC_XCD_CUS=$x IPC_PERSIST_PROF=1 timeout 300 python tools/faithful_full.py $1 $2 100000 2>&1 | tail -3 | python -c "
import sys,json
a=json.loads(sys.stdin.readline()); p=json.loads(sys.stdin.readline())['persist_profile_us']
print(a['seconds'], a['digest'], a['oracle_prefix']['decisions_differing'], 'slot 0: total', round(p['total']*1e-6,2), 'start skew', round(p['start_skew']*1e-6,2), 'launches', p['band_launches'])")"
  done
done
