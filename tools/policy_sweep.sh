run() { echo "policy $1"; IPC_SE2_POLICY="$1" python bench.py --workload C2 --steps 4 --warmup 1 --no-cpu --no-set-only --incremental-candidates 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms_per_step %.2f kernel %.2f accepted %d' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['accepted']))"; }
run "w1,w3,w5,w7,w9,w11,w13,p7,p9,p11,q7,q9,q11,q13,16x4,16x8,16x16"
run "w1,w3,w5,w7,w9,p5,p7,p9,q7,q9,q11,q13,16x4,16x8,16x16"
run "w1,w3,w5,w7,w9,w11,p7,p9,p11,q7,q9,q11,q13,16x4,16x8,16x16"
run "w1,w3,w5,w7,w9,p5,p7,p9,p11,q7,q9,q11,q13,16x4,16x8,16x16"
run "w1,w3,w5,w7,w9,p7,p9,q7,q9,q11,q13,16x4,16x8,16x16"
run "w1,w3,w5,w7,p5,p7,p9,q7,q9,q11,q13,16x4,16x8,16x16"
