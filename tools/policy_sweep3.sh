wl=$1; shift
for p in "$@"; do
  echo -n "$p => "
  IPC_SE3_POLICY="$p" timeout 900 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f pairs/s  %.1f ms/step  frac %.4f acc %d' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['accepted']))"
done
