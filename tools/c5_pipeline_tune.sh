# pipeline knobs on prefixes of the large faithful runs (every line must print the same digest per workload)
run() { wl=$1; n=$2; shift 2; echo "== $wl $n $*"; env "$@" IPC_SPEC_STATS=1 timeout 400 python tools/faithful_full.py $wl $n $n 2>&1 | grep -o "\"seconds\": [0-9.]*\|\"digest\": \"[0-9a-f]*\"\|\"launches\": [0-9]*\|\"discarded\": [0-9]*\|\"accept_solves\": [0-9]*\|\"reject_solves\": [0-9]*\|behind_an_expected_accept\": [0-9.]*" | tr '\n' ' '; echo; }
run C5 5000 IPC_SPEC_PREDICT=30
run C5 5000 IPC_SPEC_PREDICT=300
run C5 5000 IPC_SPEC_PREDICT=1000
run C4 1500 IPC_SPEC_PREDICT=10
run C4 1500 IPC_SPEC_PREDICT=100
run C4 1500 IPC_SPEC_PREDICT=1000
run C2 1256 IPC_SPEC_PREDICT=10
run C2 1256 IPC_SPEC_PREDICT=30
run C2 1256 IPC_SPEC_PREDICT=100
