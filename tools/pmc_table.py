#!/usr/bin/env python
"""Per-kernel table from a profiles/*_pmc_sq.csv: share of wave cycles, VALU busy, waits, FP64 mix, executed flops."""
import csv, collections, sys
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['kernel'].split('(')[0].replace('void ipc::', '')
    d[k][r['counter']] = d[k].get(r['counter'], 0.0) + float(r['sum_value'])
tot = collections.Counter()
for k, c in d.items():
    for n, v in c.items():
        tot[n] += v
W = tot['SQ_WAVE_CYCLES']
print("%-32s %6s %6s %6s %6s %6s %7s %7s %7s" % ("kernel", "share", "valu", "wait", "f64/v", "fma/f", "lds/v", "vmem/v", "salu/v"))
rows = sorted(d.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
for k, c in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20] + [("TOTAL", tot)]:
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    g = lambda n: c.get(n, 0.0)
    nf = g('SQ_INSTS_VALU_FMA_F64') + g('SQ_INSTS_VALU_MUL_F64') + g('SQ_INSTS_VALU_ADD_F64') + g('SQ_INSTS_VALU_TRANS_F64')
    v = max(g('SQ_INSTS_VALU'), 1)
    print("%-32s %6.3f %6.3f %6.3f %6.3f %6.3f %7.4f %7.4f %7.4f" % (k[:32], wc / W, g('SQ_ACTIVE_INST_VALU') / wc, g('SQ_WAIT_ANY') / wc, nf / v,
          g('SQ_INSTS_VALU_FMA_F64') / max(nf, 1), g('SQ_INSTS_LDS') / v, g('SQ_INSTS_VMEM_RD') / v, g('SQ_INSTS_SALU') / v))
fl = 64 * (2 * tot['SQ_INSTS_VALU_FMA_F64'] + tot['SQ_INSTS_VALU_MUL_F64'] + tot['SQ_INSTS_VALU_ADD_F64'] + tot['SQ_INSTS_VALU_TRANS_F64'])
print("executed FP64 flops %.4g   MFMA_MOPS_F64 %s   wave-cycles %.4g" % (fl, tot.get('SQ_INSTS_VALU_MFMA_MOPS_F64'), W))
