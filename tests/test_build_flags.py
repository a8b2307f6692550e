"""The compiler flags and the mailbox idiom the bit-exact parity rests on.

Round 3 found that the SE2 pair kernels built with `-mllvm -amdgpu-sched-strategy=max-ilp` return a different accepted set
on C2.  Round 4 reduced it (tools/maxilp_repro.py): only the kernels with two cooperating waves AND recomputed errors
(M > 8), non-deterministic, gone as soon as the mailbox payload is read with relaxed atomic loads -- which is what ships.
The CPU-box test pins the flags and the idiom; the GPU test builds the one suspect translation unit with both scheduling
strategies and demands bit-identical cells."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_flags_are_pinned():
    import __graft_entry__ as ge
    assert ge.HIP_FLAGS == ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=on"]
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "amdgpu-sched-strategy" not in src.replace("No -mllvm scheduling options", "")
    assert "-ffast-math" not in src and "-Ofast" not in src


def test_mailbox_payload_is_moved_with_atomic_accesses():
    cell = open(os.path.join(ROOT, "ipc_amd", "csrc", "se2_wave_cell.hpp")).read()
    group = open(os.path.join(ROOT, "ipc_amd", "csrc", "se2_group_kernel.hpp")).read()
    # every access to the mailbox payload goes through mb_load / mb_store (relaxed atomics unless -DIPC_MAILBOX_PLAIN)
    for txt in (cell, group):
        body = re.sub(r"//[^\n]*", "", txt)
        for m in re.finditer(r"box->data\[[^;]*;", body):
            stmt = m.group(0)
            ctx = body[max(0, m.start() - 40):m.end()]
            assert "mb_load(" in ctx or "mb_store(" in ctx or "__hip_atomic_load(" in ctx \
                or re.match(r"box->data\[wsub\]\[seq & 1\];", stmt), stmt
    assert "__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)" in cell
    assert "__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)" in cell
    # nothing in the build defines the reproducer's switch
    assert "IPC_MAILBOX_PLAIN" not in open(os.path.join(ROOT, "__graft_entry__.py")).read()


@pytest.mark.gpu
def test_pair_kernel_is_bit_identical_under_both_scheduling_strategies():
    """se2_pair.hip restricted to p9 (two waves per cell, errors recomputed), default strategy against max-ilp, every
    cell of workload T700 (46 512 of them through that kernel): same chi2 bits, same iteration counts."""
    import __graft_entry__ as ge
    ge.build()
    objdir = os.path.join(ROOT, "build")
    if not all(os.path.exists(os.path.join(objdir, u.replace(".hip", ".o"))) for u in ge.UNITS):
        ge._build_lib(ge.LIB)                          # (the objects of the other translation units did not travel)
    tool = os.path.join(ROOT, "tools", "maxilp_repro.py")
    subprocess.check_call([sys.executable, tool, "build", "def=", "ilp=mllvm:-amdgpu-sched-strategy=max-ilp"])
    try:
        out = subprocess.run([sys.executable, tool, "run", "def", "ilp"], capture_output=True, text=True, check=True).stdout
    finally:
        for n in ("def", "ilp"):
            p = os.path.join(ROOT, "ipc_amd", "librepro_%s_m9.so" % n)
            if os.path.exists(p):
                os.remove(p)
    line = [l for l in out.splitlines() if l.startswith("ilp")][0]
    m = re.search(r"\(pM: (\d+)\).*differing from def: (\d+) cells", line)
    assert m, out
    assert int(m.group(1)) > 40000 and int(m.group(2)) == 0, line
