#!/usr/bin/env python
"""Instruction mix of a gfx950 assembly file (hipcc -S): counts by class, static (whole kernel)."""
import re, sys, collections
c = collections.Counter()
ops = collections.Counter()
for line in open(sys.argv[1]):
    m = re.match(r"\s+([vsdgb][a-z0-9_]+)\s", line)
    if not m:
        continue
    op = m.group(1)
    ops[op] += 1
    if op.startswith("v_accvgpr"): k = "accvgpr"
    elif op.endswith("_f64") or "_f64_" in op:
        k = "f64_cmp" if "cmp" in op else ("f64_cvt" if "cvt" in op or "rndne" in op else "f64")
    elif op.startswith("v_mov") and "dpp" in line: k = "dpp_mov"
    elif op.startswith("v_mov"): k = "v_mov"
    elif op.startswith("v_cndmask"): k = "cndmask"
    elif op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): k = "lane"
    elif op.startswith("v_perm"): k = "permlane"
    elif op.startswith("v_"): k = "v_other"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"): k = "vmem"
    elif op.startswith("s_waitcnt"): k = "s_waitcnt"
    elif op.startswith("s_nop"): k = "s_nop"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    if "dpp" in line and k == "f64": k = "f64_dpp"
    c[k] += 1
tot = sum(c.values())
for k, v in c.most_common():
    print("%-10s %6d %5.1f%%" % (k, v, 100.0 * v / tot))
print("total", tot)
if len(sys.argv) > 2:
    for k, v in ops.most_common(int(sys.argv[2])):
        print("   %-28s %d" % (k, v))
