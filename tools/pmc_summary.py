#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection csv per kernel: dispatches, mean and sum of each counter.

usage: pmc_summary.py <counter_collection.csv> [<more.csv> ...] > summary.csv
Only launches with a grid of at least --min-grid work-items are counted (default 1), so the one-stream set-up
launches of bench.py can be separated from the batched ones with --min-grid.
"""
import csv, sys, argparse, collections

ap = argparse.ArgumentParser()
ap.add_argument("files", nargs="+")
ap.add_argument("--min-grid", type=int, default=1)
a = ap.parse_args()
acc = collections.defaultdict(lambda: [0, 0.0])
for f in a.files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            try:
                grid = int(float(row.get("Grid_Size", 0) or 0))
            except ValueError:
                grid = 0
            if grid < a.min_grid:
                continue
            name = row["Kernel_Name"].split("(")[0]
            k = (name, row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "sum"])
for (name, cn), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, cn, n, "%.3f" % (s / n), "%.3f" % s])
