#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 kernel_trace csv, restricted to launches of at least --min-grid work-items
(bench.py's set-up runs the same kernels on one stream thousands of times; the batched launches are the ones measured).

usage: trace_summary.py <kernel_trace.csv> [--min-grid N] > summary.csv
"""
import csv, sys, argparse, collections

ap = argparse.ArgumentParser()
ap.add_argument("file")
ap.add_argument("--min-grid", type=int, default=1)
a = ap.parse_args()
acc = collections.defaultdict(list)
with open(a.file, newline="") as fh:
    for row in csv.DictReader(fh):
        g = 1
        for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
            try:
                g *= max(1, int(float(row.get(k, 1) or 1)))
            except ValueError:
                pass
        if "Grid_Size" in row and row["Grid_Size"]:
            g = int(float(row["Grid_Size"]))
        if g < a.min_grid:
            continue
        acc[row["Kernel_Name"].split("(")[0]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns"])
for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, len(d), sum(d), "%.1f" % (sum(d) / len(d)), min(d), max(d)])
