#!/usr/bin/env python3
"""Benchmark of the MI355X PTAM hot path: frames/sec of Tracker::TrackFrame + local bundle adjustment.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one vslam_track_frame over one batch of frames: every one of the S independent sequences (streams) on
this GPU advances by one 640x480 frame -- MakeKeyFrame_Lite (pyramid + FAST-10), TrackMap (PVS, patch search, 10+10
Gauss-Newton iterations), and, whenever the tracker asks for a keyframe (every ~21 frames per stream),
MapMaker::AddKeyFrame + one BundleAdjustRecent, all on device.  Frames are synthetic (seeded feeder) and resident in
HBM before the timed region.  Sequences shard across GPUs with no data-path collective (weak scaling); RCCL is used only
to gather the per-rank statistics and the max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` for the dominant kernel (HIP-event time measured live over the timed
region on the library's own stream) and `cpu_baseline` (the oracle's TrackFrame+BA on one host core, bounded sample).
"""
import argparse
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector (= fp64 matrix) peak, AMD product brief; the guide lists no fp64 figure: 256 CUs x 128 FMA/clk x 2.4 GHz


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def aggregate(elapsed_s, stats, world):
    """max-over-ranks time and gathered per-rank stats (RCCL all_gather of a small fp64 vector; gloo in CPU tests)."""
    import torch
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return elapsed_s, [list(stats)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = torch.tensor(list(stats), dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return float(t.item()), [o.cpu().tolist() for o in out]


def algorithmic_bytes(stage, S, W, H, P, per_stream):
    """SURVEY.md 8(d) algorithmic bytes of ONE launch of `stage` over S streams (per-stream averages in per_stream)."""
    ncorn, npatch, nzm, nfound, ba_m, ba_c, ba_p, ba_trials_per_launch = (per_stream[k] for k in
        ("corners", "patches", "zmssd", "found", "ba_meas", "ba_cams", "ba_pts", "ba_trials_per_launch"))
    hsum = sum(H >> l for l in range(4))
    b_fast = W * H * (1 + 21.0 / 64.0) + 4 * ncorn + 4 * hsum
    b_patch = P * P * (npatch + nzm) + 48 * npatch
    b_pose = 10 * (nfound * 120 + 216)
    b_ba = ba_trials_per_launch * (ba_m * 176 + ba_c * 312 + ba_p * 168)
    table = {"pyr_fast0": W * H * (1 + 21.0 / 64.0) + (W * H) / 8.0, "fast_lvl": W * H * (21.0 / 64.0) * (1 + 1 / 8.0),
             "compact": 4 * ncorn + 4 * hsum + (W * H * (1 + 21.0 / 64.0)) / 8.0,
             "search_fine": b_patch, "search_coarse": 0.0, "pose_fine": b_pose, "pose_coarse": 0.0, "ba_compute": b_ba,
             "fast_stage": b_fast}
    return S * table.get(stage, 0.0)


def baseline_metric():
    """The headline metric exactly as BASELINE.json names it (the file sits beside bench.py and travels with the repository)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, ValueError, KeyError):
        return "frames/sec (TrackFrame+local BA) on 640\u00d7480 synthetic, 1/2/4/8 GPU"


def pmc_traffic(kernel, cfg):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*traffic.json), or None.

    bench.py cannot run rocprofv3 on itself; the counters were collected with this same command and configuration
    (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, kilobytes; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).
    """
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        if t.get("kernel") == kernel and all(cfg.get(k) == v for k, v in t.get("config", {}).items()):
            return {"bytes_per_launch": round((2.0 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024.0),
                    "source": os.path.relpath(f, os.path.dirname(os.path.abspath(__file__)))}
    return None


def ba_flops(per_stream):
    """SURVEY.md 8(d) fp64 flops of ONE k_ba_compute launch per stream: F_ba = M*780 + sum_pts C(n_free,2)*216 + (6n)^3/3 per trial."""
    n = per_stream["ba_free"]
    per_trial = per_stream["ba_meas"] * 780.0 + per_stream["ba_pts"] * (n * (n - 1) / 2.0) * 216.0 + (6.0 * n) ** 3 / 3.0
    return per_stream["ba_trials_per_launch"] * per_trial


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VSLAM_BENCH_STREAMS", 1024)), help="sequences per GPU")
    ap.add_argument("--systems", type=int, default=int(os.environ.get("VSLAM_BENCH_SYSTEMS", 1)),
                    help="split the streams over this many vslam_system handles (one HIP stream each) so their kernels overlap")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--patch", type=int, default=8, help="PatchFinder template side (BASELINE configs: 8; reference default 11)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--ba-delay", type=int, default=int(os.environ.get("VSLAM_BENCH_BA_DELAY", 16)),
                    help="vslam_params.ba_delay_frames: 0 = synchronous map-maker; D > 0 = Bundle::Compute on its own HIP stream, applied D frames later")
    ap.add_argument("--use-sbi", type=int, default=int(os.environ.get("VSLAM_BENCH_USE_SBI", 0)),
                    help="vslam_params.use_sbi: 1 = SmallBlurryImage rotation prior in the motion model (the reference's gvnUseSBI)")
    ap.add_argument("--grow-map", type=int, default=int(os.environ.get("VSLAM_BENCH_GROW_MAP", 0)),
                    help="vslam_params.grow_map bit flags: 1 = AddSomeMapPoints (epipolar search), 2 = ReFindInSingleKeyFrame, 3 = both (the reference)")
    ap.add_argument("--diag-kf-dist-mult", type=float, default=None,
                    help="DIAGNOSTIC ONLY (not the metric): overrides vslam_params.max_kf_dist_wiggle_mult, e.g. 1e9 = no keyframes, no BA")
    ap.add_argument("--no-events", action="store_true", help="skip the per-stage HIP events in the timed region")
    args = ap.parse_args()
    rank, world, local_rank = dist_env()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from visualslam_android_amd import capi, feeder
    if os.environ.get("VSLAM_LIB"):            # diagnostic builds from tools/build_variant.sh (A/B runs on one box)
        capi.load_library(os.environ["VSLAM_LIB"])
    S, W, H, K, Wm = args.streams, args.width, args.height, args.steps, args.warmup
    T = Wm + K
    nthreads = max(2, min(16, (os.cpu_count() or 8) // max(1, world)))   # set-up threads per rank: the ranks of a node share its cores

    # ---- synthetic scenes: one seeded feeder, trajectory and ground-truth map per stream -----------------------------
    t_setup = time.time()
    seeds = [1234 + rank * S + s for s in range(S)]
    with ThreadPoolExecutor(nthreads) as ex:
        feeders = list(ex.map(lambda sd: feeder.Feeder(W, H, seed=sd), seeds))
    vp = capi.default_params(W, H, S, patch_size=args.patch, device=local_rank)
    fe = capi.System(capi.default_params(W, H, 1, patch_size=args.patch, device=local_rank))   # front-end used to pick map corners

    fe_lock = threading.Lock()

    def corner_fn(gray):
        with fe_lock:                          # one front-end system; the renders and the numpy of build_map run in parallel
            fe.make_keyframe_lite(gray[None])
            fe.fast_nonmax()
            return [fe.read_max_corners(0, l)[0] for l in range(4)]

    with ThreadPoolExecutor(nthreads) as ex:
        maps = list(ex.map(lambda f: feeder.build_map(f, corner_fn), feeders))
    fe.close()
    NS = max(1, min(args.systems, S))
    assert S % NS == 0, "--streams must be a multiple of --systems"
    Sk = S // NS
    vpk = capi.default_params(W, H, Sk, patch_size=args.patch, device=local_rank, ba_delay_frames=args.ba_delay, use_sbi=args.use_sbi, grow_map=args.grow_map)
    if args.diag_kf_dist_mult is not None:
        vpk.max_kf_dist_wiggle_mult = args.diag_kf_dist_mult
    systems = [capi.System(vpk) for _ in range(NS)]

    def sys_of(s):
        return systems[s // Sk], s % Sk

    for s in range(S):
        sy, ls = sys_of(s)
        sy.load_map(ls, maps[s])
        sy.set_pose(ls, feeders[s].pose(-1))
    frames_dev = torch.empty((T, S, H, W), dtype=torch.uint8, device="cuda")
    host_frames0 = None
    with ThreadPoolExecutor(max(1, nthreads // 4)) as ex:
        for s, fr in enumerate(ex.map(lambda f: f.render(0, T, threads=4), feeders)):
            frames_dev[:, s].copy_(torch.from_numpy(fr))
            if s == 0:
                host_frames0 = fr
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    base = frames_dev.data_ptr()
    fstride = S * H * W

    def step(t):
        for k, sy in enumerate(systems):
            sy.track_frame_device(base + t * fstride + k * Sk * H * W, W, H * W)

    def sync_all():
        for sy in systems:
            sy.synchronize()

    def state(s):
        sy, ls = sys_of(s)
        return sy.state(ls)

    # ---- warm-up, then the timed region ---------------------------------------------------------------------------------
    for t in range(Wm):
        step(t)
    sync_all()
    st0 = [state(s) for s in range(S)]
    if not args.no_events:
        for sy in systems:
            sy.profile_begin(K)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, T):
        step(t)
    sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    stage_ms, nprof = {}, 0
    if not args.no_events:
        for sy in systems:                      # per-launch durations: summed over systems, averaged below over launches
            ms, n = sy.profile_end()
            nprof += n
            for k_, v_ in ms.items():
                stage_ms[k_] = stage_ms.get(k_, 0.0) + v_
    st1 = [state(s) for s in range(S)]

    # ---- SURVEY 8(f) row 1, outside the timed region: MakeKeyFrame_Rest (non-max + Shi-Tomasi candidates) and
    #      ThinCandidates of the last frame, all streams; not part of `value` (nothing on the built path consumes them yet)
    rest_ms = None
    try:
        sync_all()
        t_r = time.perf_counter()
        reps = 5
        for _ in range(reps):
            for sy in systems:
                sy.make_keyframe_rest(70.0)
                sy.thin_candidates(-1)
        sync_all()
        rest_ms = 1e3 * (time.perf_counter() - t_r) / reps
        n_cand = [len(systems[0].read_candidates(0, l)[0]) for l in range(4)]
    except Exception as e:                     # noqa: BLE001 - extra information only
        n_cand = str(e)

    # ---- per-stream workload statistics (for the algorithmic-byte formulas) ---------------------------------------------
    zm = float(np.mean([(b.n_zmssd - a.n_zmssd) / K for a, b in zip(st0, st1)]))
    att = float(np.mean([sum(b.attempted) for b in st1]))
    fnd = float(np.mean([sum(b.found) for b in st1]))
    kf_adds = float(np.mean([b.n_keyframes - a.n_keyframes for a, b in zip(st0, st1)]))
    ba_trials = float(np.mean([b.n_ba_trials - a.n_ba_trials for a, b in zip(st0, st1)]))
    good = int(sum(1 for b in st1 if b.quality == 2))
    ncorn = float(len(systems[0].read_corners(0, 0)))
    bs = systems[0].bundle_stats(0)           # sizes of the last bundle-adjustment problem of stream 0 (all streams alike)
    per_stream = {"corners": ncorn, "patches": att, "zmssd": zm, "found": fnd, "ba_meas": float(bs["meas"]), "ba_cams": float(bs["cams"]),
                  "ba_free": float(bs["free_cams"]), "ba_pts": float(bs["points"]), "ba_trials_per_launch": ba_trials / K}

    total_t, gathered = aggregate(elapsed, [elapsed, S * K, zm, att, fnd, kf_adds, ba_trials, good], world)
    frames_total = sum(g[1] for g in gathered)
    value = frames_total / total_t

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events over the timed region) -----------------------------------------
        roof = None
        stages = {}
        if stage_ms:
            for name, ms in stage_ms.items():
                per_launch_ms = ms / max(1, nprof)
                ab = algorithmic_bytes(name, Sk, W, H, args.patch, per_stream)
                stages[name] = {"ms_per_launch": round(per_launch_ms, 5), "algorithmic_GBps": round(ab / (per_launch_ms * 1e-3) / 1e9, 3) if per_launch_ms > 0 and ab > 0 else None}
            cfg_key = {"streams_per_gpu": S, "patch_size": args.patch, "ba_delay_frames": args.ba_delay, "width": W, "height": H}

            def roofline_of(name):
                ms_ = stage_ms[name] / max(1, nprof)
                ab_ = algorithmic_bytes(name, Sk, W, H, args.patch, per_stream)
                ach_ = ab_ / (ms_ * 1e-3) / 1e9
                r_ = {"kernel": name, "bound": "hbm", "achieved": round(ach_, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(ach_ / HBM_PEAK_GBS, 6), "traffic": None, "ms_per_launch": round(ms_, 5)}
                if name == "ba_compute":        # supplementary: the same launch against the fp64 vector/matrix peak (78.6 TFLOP/s)
                    tf = Sk * ba_flops(per_stream) / (ms_ * 1e-3) / 1e12
                    r_["fp64"] = {"achieved": round(tf, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 5)}
                tr = pmc_traffic(name, cfg_key)
                if tr:
                    r_["traffic"] = tr["bytes_per_launch"]; r_["traffic_source"] = tr["source"]
                r_["algorithmic_bytes"] = round(ab_)
                return r_

            # the dominant kernel = the largest summed HIP-event time over the timed region among the kernels with a byte model;
            # k_ba_compute and k_pyr_fast0 are within a few per cent of each other, so the runners-up are listed as well
            ranked = sorted((n for n in stage_ms if algorithmic_bytes(n, Sk, W, H, args.patch, per_stream) > 0), key=lambda n: -stage_ms[n])
            roof = roofline_of(ranked[0])
            roof["others"] = [roofline_of(n) for n in ranked[1:]]
            roof["problem"] = {k: per_stream[k] for k in ("ba_cams", "ba_free", "ba_pts", "ba_meas", "ba_trials_per_launch")}
        # ---- CPU baseline: the oracle's TrackFrame + BA on one core over a bounded sample of the same frames -------------
        from oracle import binding as orc
        o = orc.OracleSystem(orc.params_from_vslam(vpk))
        o.load_map(maps[0])
        o.set_pose(feeders[0].pose(-1))
        tc = time.perf_counter()
        n_cpu = 0
        for t in range(T):
            o.track_frame(host_frames0[t])
            n_cpu += 1
            if time.perf_counter() - tc > args.cpu_seconds:
                break
        cpu_s = time.perf_counter() - tc
        pose_diff = None
        if n_cpu == T:
            pose_diff = float(np.abs(np.array(o.state().pose[:]) - np.array(st1[0].pose[:])).max())
        out = {
            "metric": baseline_metric(), "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(1e3 * total_t / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %dx%d 4-level FAST-10 + %dx%d PatchFinder ZMSSD search, TrackMap pose update, "
                                   "AddKeyFrame + BundleAdjustRecent on keyframe frames" % (W, H, args.patch, args.patch),
                       "streams_per_gpu": S, "systems_per_gpu": NS, "ba_delay_frames": args.ba_delay, "use_sbi": args.use_sbi, "grow_map": args.grow_map, "diagnostic_kf_dist_mult": args.diag_kf_dist_mult, "patch_size": args.patch, "corners_l0_per_frame": ncorn,
                       "patches_attempted_per_frame": round(att, 1), "patches_found_per_frame": round(fnd, 1),
                       "zmssd_evals_per_frame": round(zm, 1), "keyframes_added_per_stream": kf_adds,
                       "ba_trials_per_stream": ba_trials, "streams_tracking_good": good, "setup_seconds": round(setup_s, 1)},
            "roofline": roof,
            "cpu_baseline": {"value": round(n_cpu / cpu_s, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                             "sample": "oracle TrackFrame+BA on the first %d frames of stream 0 (same frames, same map)" % n_cpu},
            "stages": stages,
            "parity_pose_maxdiff_stream0": pose_diff,
            "next_rows": {"make_keyframe_rest_plus_thin_ms_per_launch": None if rest_ms is None else round(rest_ms, 4),
                          "candidates_left_stream0": n_cand},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
