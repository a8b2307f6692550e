for hg in 0 1 2; do
  for w in "C4 2500" "C5 6000"; do
    set -- $w
    echo "hedge $hg $1 $2: $(IPC_SPEC_HEDGE=$hg IPC_SPEC_STATS=1 timeout 300 python tools/faithful_full.py $1 $2 100000 2>&1 | tail -2 | python -c "
import sys,json
a=json.loads(sys.stdin.readline()); b=json.loads(sys.stdin.readline())['speculation']
print(a['seconds'], a['digest'], a['oracle_prefix']['decisions_differing'], 'launches', b['launches'], 'discarded', b['discarded'], 'timeouts', b['persist_timeouts'])")"
  done
done
echo "C1/C2 default:"; for w in C1 C2; do timeout 200 python tools/faithful_full.py $w -1 100000 2>&1 | tail -1 | cut -c1-200; done
