#!/usr/bin/env python3
"""Occupancy of the timed region of a bench.py run from a rocprofv3 kernel_trace csv: the window is the last --steps launches of
k_pose with the batched grid; per kernel the summed duration inside the window, and per queue the busy time.

usage: timeline.py <kernel_trace.csv> --steps 20 --min-grid 512"""
import csv, sys, argparse, collections
ap = argparse.ArgumentParser()
ap.add_argument("file"); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--min-grid", type=int, default=512)
a = ap.parse_args()
rows = []
with open(a.file, newline="") as fh:
    for r in csv.DictReader(fh):
        g = 1
        for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
            try: g *= max(1, int(float(r.get(k, 1) or 1)))
            except ValueError: pass
        if r.get("Grid_Size"): g = int(float(r["Grid_Size"]))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), g))
rows.sort()
poses = [x for x in rows if x[2].startswith("k_pose") and x[4] >= a.min_grid]
# two k_pose launches per step at most (coarse + fine); take the window covering the last `steps` fine launches = every launch
ends = [x[1] for x in poses]
per = 2 if len(poses) >= 2 * a.steps + 2 and (poses[-1][0] - poses[-2][0]) < (poses[-2][0] - poses[-3][0]) else 1
first = poses[-a.steps * per - 1][1] if len(poses) > a.steps * per else rows[0][0]
# (between the warm-up and the timed region bench.py reads every stream's state: a run of small copies that is not part of the window)
rows = [x for x in rows if not x[2].startswith("__amd_rocclr_copyBuffer")]
t0 = min([x[0] for x in rows if x[0] >= first] or [first])
t1 = poses[-1][1]
# pauses of the whole device longer than 5 ms are the host's (bench.py reading every stream's state around the timed region): they
# are taken out of the window
_iv = sorted((max(s, t0), min(e, t1)) for s, e, n, qu, g in rows if min(e, t1) > max(s, t0))
pauses = []; _ce = t0
for s_, e_ in _iv:
    if s_ - _ce > 5_000_000: pauses.append((_ce, s_))
    _ce = max(_ce, e_)
paused = sum(b - a_ for a_, b in pauses)
WIN = (t1 - t0) - paused
print("window %.3f ms (%d host pauses of %.1f ms in all taken out), %d steps -> %.3f ms/step" % (WIN / 1e6, len(pauses), paused / 1e6, a.steps, WIN / 1e6 / a.steps))
acc = collections.defaultdict(lambda: [0, 0]); q = collections.defaultdict(list)
ev = []
for s, e, n, qu, g in rows:
    s2, e2 = max(s, t0), min(e, t1)
    if e2 <= s2: continue
    acc[n][0] += e2 - s2; acc[n][1] += 1; q[qu].append((s2, e2)); ev.append((s2, 1)); ev.append((e2, -1))
ev.sort()
busy = 0; depth = 0; last = t0; hist = collections.defaultdict(int)
for t, d in ev:
    if depth > 0: busy += t - last
    hist[depth] += t - last
    depth += d; last = t
hist[0] += t1 - last
hist[0] -= paused
print("some kernel running: %.1f %% of the window; concurrency histogram (kernels in flight: %% of time): %s" % (100.0 * busy / WIN, {k: round(100.0 * v / WIN, 1) for k, v in sorted(hist.items())}))
tot = sum(v[0] for v in acc.values())
print("sum of kernel durations / window = %.2f" % (tot / WIN))
for n, (d, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-44s %6d launches %9.3f ms/step %6.1f %% of window" % (n[:44], c, d / 1e6 / a.steps, 100.0 * d / WIN))
for qu, iv in q.items():
    print("queue %s busy %.1f %%" % (qu, 100.0 * sum(e - s for s, e in iv) / WIN))
# the idle gaps of the window (no kernel in flight): how many, how long, and what ran before / after the longest ones
inw = sorted((max(s, t0), min(e, t1), n, qu) for s, e, n, qu, g in rows if min(e, t1) > max(s, t0))
gaps = []; cur_end = t0; last_n = ("-", "-")
for s, e, n, qu in inw:
    if s > cur_end and s - cur_end <= 5_000_000: gaps.append((s - cur_end, cur_end - t0, last_n, (n, qu)))
    if e > cur_end: cur_end = e; last_n = (n, qu)
tot_gap = sum(g[0] for g in gaps)
print("idle gaps: %d, %.3f ms in all (%.1f %% of the window), median %.1f us" % (len(gaps), tot_gap / 1e6, 100.0 * tot_gap / WIN, sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0))
byk = collections.defaultdict(lambda: [0, 0])
for d, at, a_, b_ in gaps: byk[(a_[0][:28], b_[0][:28])][0] += d; byk[(a_[0][:28], b_[0][:28])][1] += 1
for (ka, kb), (d, c) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  after %-28s before %-28s %5d gaps %8.3f ms" % (ka, kb, c, d / 1e6))
# per queue: where its own idle time goes (the kernel before / after each gap of that queue, pauses excluded)
for qu, iv in sorted(q.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:2]:
    mine = sorted((max(s, t0), min(e, t1), n) for s, e, n, qq, g in rows if qq == qu and min(e, t1) > max(s, t0))
    pair = collections.defaultdict(lambda: [0, 0]); ce = None; ln = "-"
    for s, e, n in mine:
        if ce is not None and s > ce and s - ce <= 5_000_000: pair[(ln[:26], n[:26])][0] += s - ce; pair[(ln[:26], n[:26])][1] += 1
        if ce is None or e > ce: ce = e; ln = n
    tot_q = sum(v[0] for v in pair.values())
    print("queue %s idle between its own kernels: %.3f ms (%.1f %% of the window)" % (qu, tot_q / 1e6, 100.0 * tot_q / WIN))
    for (ka, kb), (d, c) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:10]:
        print("  after %-26s before %-26s %5d gaps %8.3f ms" % (ka, kb, c, d / 1e6))
