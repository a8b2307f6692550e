#!/bin/bash
# usage: bisect_round.sh lim1 lim2 ...   (builds one debug lib per opt-bisect limit and tests them on the GPU)
cd /root/repo/ipc_amd/csrc
rm -f /root/repo/ipc_amd/libipc_dbg_b*.so
for l in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DIPC_DEBUG_ONE_VARIANT -mllvm -opt-bisect-limit=$l -o /root/repo/ipc_amd/libipc_dbg_b$l.so engine.hip > /dev/null 2>&1 &
done
wait
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 900 -- 'for f in ipc_amd/libipc_dbg_b*.so; do echo -n "$f "; timeout 60 python tools/dbg_side.py $f 2>&1 | tail -1; done' 2>&1 | grep "libipc_dbg_b" | sort -t b -k 4 -n
