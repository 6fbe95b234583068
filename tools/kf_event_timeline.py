#!/usr/bin/env python3
"""Timeline of one keyframe event from a rocprofv3 kernel_trace csv: the kernels between the k_add_keyframe launch that did
work and the following k_ba_assemble, with start offsets and durations in microseconds (usage: kf_event_timeline.py trace.csv)."""
import csv, sys
rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", "")))
rows.sort()
adds = [i for i, r in enumerate(rows) if r[2] == "k_add_keyframe"]
best = max(adds, key=lambda i: rows[i][1] - rows[i][0])
t0 = rows[best][0]
for s, e, n, st in rows[best:]:
    if s - t0 > 80e6:
        break
    if (e - s) > 20000 or n in ("k_ba_assemble", "k_add_keyframe"):
        print("%9.1f us  +%8.1f us  stream %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n[:40]))
    if n == "k_ba_assemble" and s > t0:
        break
