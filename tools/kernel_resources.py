"""Registers / LDS / scratch of every kernel in libipc_amd.so (reads the embedded gfx950 code
objects).  Usage: python tools/kernel_resources.py [substring]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = os.environ.get("IPC_LIB", os.path.join(ROOT, "ipc_amd", "libipc_amd.so"))
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    data = open(lib, "rb").read()
    pos = 0
    rows = []
    while True:
        idx = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if idx < 0:
            break
        n = struct.unpack_from("<Q", data, idx + 24)[0]
        off = idx + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            t = data[off:off + tl].decode()
            off += tl
            if "gfx950" in t and s:
                with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                    f.write(data[idx + o:idx + o + s])
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name],
                                     capture_output=True, text=True).stdout
                os.unlink(f.name)
                cur = {}
                for line in txt.splitlines():
                    m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)", line)
                    if not m:
                        continue
                    k, v = m.group(1), m.group(2).strip()
                    if k == "name" and "vgpr_count" in cur:
                        pass
                    cur[k] = v
                    if k == "vgpr_spill_count":
                        rows.append(dict(cur))
                        cur = {}
        pos = idx + 24
    for r in rows:
        name = subprocess.run(["c++filt", r.get("name", "")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if pat in name:
            print("%-46s vgpr %4s agpr %4s sgpr %4s lds %7s scratch %5s vspill %4s" % (
                name[-46:], r.get("vgpr_count"), r.get("agpr_count", "-"), r.get("sgpr_count"),
                r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size"), r.get("vgpr_spill_count")))


if __name__ == "__main__":
    main()
