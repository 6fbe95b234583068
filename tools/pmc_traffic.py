#!/usr/bin/env python3
"""HBM traffic per unit of work of the path's kernels from a PMC summary of bench.py's OWN launches (tools/collect_pmc_bench.sh):
   python3 tools/pmc_traffic.py <summary.csv> <bench line of the FETCH pass> <bench line of the WRITE pass> <out prefix>
Bytes = 2 x FETCH_SIZE + WRITE_SIZE (kilobytes in the counters; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes for
wide coalesced reads -- other access widths are uncalibrated there, so the figure is an upper estimate for the gather-heavy kernels).
Units: k_ba_compute per LM trial (the trials every launch of the run counted on the device); the tracker's kernels per frame of one
stream (both launches of a kernel that runs twice per frame, e.g. coarse + fine, together)."""
import csv, json, sys
summ, f_line, w_line, prefix = sys.argv[1:5]
rows = {}
for r in csv.DictReader(open(summ)):
    rows[(r["kernel"].replace("void ", "").strip(), r["counter"])] = (int(r["dispatches"]), float(r["sum"]))
lines = [json.loads(open(p).read().strip().splitlines()[-1]) for p in (f_line, w_line)]
S = lines[0]["config"]["streams_per_gpu"]
T = lines[0]["frames_whole_run"]
def kb(kernel, counter):
    return rows.get((kernel, counter), (0, 0.0))
def emit(name, kernels, unit, units_f, units_w, alg_bytes, what):
    f = sum(kb(k, "FETCH_SIZE")[1] for k in kernels) * 1024.0
    w = sum(kb(k, "WRITE_SIZE")[1] for k in kernels) * 1024.0
    bpu = 2.0 * f / units_f + w / units_w
    out = {"kernel": name, "kernels": kernels, "unit": unit, "bytes_per_unit": round(bpu, 1), "fetch_bytes_per_unit_as_counted": round(f / units_f, 1), "write_bytes_per_unit": round(w / units_w, 1),
           "algorithmic_bytes_per_unit": round(alg_bytes, 1), "ratio_to_algorithmic": round(bpu / alg_bytes, 3) if alg_bytes else None,
           "dispatches": {k: kb(k, "FETCH_SIZE")[0] for k in kernels}, "streams": S, "frames": T, "source": what}
    json.dump(out, open("%s_%s.json" % (prefix, name.replace("k_", "")), "w"), indent=1)
    print(name, out["bytes_per_unit"], "B per", unit, "ratio", out["ratio_to_algorithmic"])
what = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --streams %d ...: the bench's own launches (tools/collect_pmc_bench.sh)" % S
tf, tw = (l["ba_launch_totals_whole_run"] for l in lines)
alg = lambda t: (t["trials_x_meas"] * 176.0 + t["trials_x_cams"] * 312.0 + t["trials_x_points"] * 168.0) / max(1, t["trials"])
emit("k_ba_compute", ["k_ba_compute"], "LM trial of one problem", max(1, tf["trials"]), max(1, tw["trials"]), alg(tf), what)
c = lines[0]["config"]
P, W, H = c["patch_size"], 640, 480
frames = float(S * T)
hsum = sum(H >> l for l in range(4))
emit("k_front_end", ["k_fast_slide<true>", "k_fast_slide<false>", "k_compact"], "frame of one stream", frames, frames, W * H * (1 + 21.0 / 64.0) + 4 * c["corners_l0_per_frame"] + 4 * hsum, what)
emit("search_fine", ['"k_searchN<8, 8>"'.strip('"'), 'k_subpixN<8, 8>'], "frame of one stream", frames, frames, P * P * (c["patches_attempted_per_frame"] + c["zmssd_evals_per_frame"]) + 48 * c["patches_attempted_per_frame"], what)
emit("pose_fine", ["k_pose"], "frame of one stream", frames, frames, 10 * (c["patches_found_per_frame"] * 120 + 216), what)
emit("k_pvs", ["k_pvs"], "frame of one stream", frames, frames, c.get("map_points_per_stream", 2250.0) * 220.0, what)
