#!/usr/bin/env python
"""Builds a variant of libipc_amd.so in which only the named translation units get extra compiler flags (all other
objects are the default build's): python tools/build_variant.py <out.so> <unit[,unit...]> <flag> [<flag> ...]
e.g.  python tools/build_variant.py ipc_amd/libipc_maxilp_wave.so se2_wave.hip -mllvm -amdgpu-sched-strategy=max-ilp"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    out, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = list(ge.HIP_FLAGS)
    objdir = os.path.join(ROOT, "build")
    vdir = os.path.join(ROOT, "build", "variant_" + os.path.basename(out).replace(".so", ""))
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for u in ge.UNITS:
        if u in units:
            obj = os.path.join(vdir, u.replace(".hip", ".o"))
            subprocess.check_call([hipcc] + base + flags + ["-c", os.path.join(ge.CSRC, u), "-o", obj])
        else:
            obj = os.path.join(objdir, u.replace(".hip", ".o"))
            assert os.path.exists(obj), "build the default library first"
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print("built", out)


if __name__ == "__main__":
    main()
