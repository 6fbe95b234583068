#!/usr/bin/env python3
"""Per-kernel reading of an SQ counter summary (tools/pmc_summary.py output): python tools/sq_table.py summary.csv
busy = share of the wave cycles in which the wavefront issued something; wait = parked on s_waitcnt / barriers; valu = cycles the
vector unit was busy per wave cycle x waves per SIMD is left to the reader (the counters are sums over all SIMDs)."""
import csv, sys, collections
d = collections.defaultdict(dict)
n = {}
for r in csv.DictReader(open(sys.argv[1])):
    d[r["kernel"]][r["counter"]] = float(r["sum"]); n[r["kernel"]] = int(r["dispatches"])
print("%-28s %6s %10s %8s %8s %8s %9s %9s %9s %8s %8s" % ("kernel", "disp", "wavecyc(G)", "wait%", "issue%", "valu%", "valu_i(M)", "lds_i(M)", "vmem_i(M)", "ldsconf%", "cyc/valu"))
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    g = lambda x: c.get(x, 0.0)
    print("%-28s %6d %10.3f %8.1f %8.1f %8.1f %9.1f %9.1f %9.1f %8.1f %8.2f" % (
        k[:28], n[k], wc / 1e9, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc,
        g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / 1e6,
        100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_ACTIVE_INST_LDS"), 1), g("SQ_ACTIVE_INST_VALU") / max(g("SQ_INSTS_VALU"), 1)))
