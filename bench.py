#!/usr/bin/env python3
"""Benchmark of the MI355X PTAM hot path: frames/sec of Tracker::TrackFrame + local bundle adjustment.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment, bench.py starts the N ranks itself (python -m torch.distributed.run, one
rank per GPU over RCCL) and relays rank 0's JSON line; started BY torch.distributed.run (RANK / WORLD_SIZE set) it is one
of the ranks.  It refuses to run when the world size is not --gpus.

One "step" = one vslam_track_frame over one batch of frames: every one of the S independent sequences (streams) on
this GPU advances by one frame -- MakeKeyFrame_Lite (pyramid + FAST-10), TrackMap (PVS, patch search, 10+10
Gauss-Newton iterations), and, whenever the tracker asks for a keyframe (every ~21 frames per stream),
MapMaker::AddKeyFrame + one BundleAdjustRecent, all on device.  Frames are synthetic (seeded feeder) and resident in
HBM before the timed region.  The sequences are independent, so their keyframe phases are spread evenly over the keyframe
period (--kf-stagger; otherwise all of them would ask for their first keyframe in the same frame and stay in lock-step):
every step then carries the same mix of work -- S tracked frames and about S / 21 keyframes with their bundle adjustments
-- and the rate does not depend on where the timed window falls.  Sequences shard across GPUs with no data-path collective
(weak scaling); RCCL is used only to gather the per-rank statistics and the max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` for the dominant kernel (HIP-event time measured live over the timed
region on the library's own streams) and `cpu_baseline` (the oracle's TrackFrame+BA on the host cores, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector (= fp64 matrix) peak, AMD product brief; the guide lists no fp64 figure: 256 CUs x 128 FMA/clk x 2.4 GHz
KF_PERIOD = 21           # frames between two keyframes of a stream: min_frames_between_kf (20, jni/Tracker.cc:128) + 1


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """bench.py --gpus N outside torch.distributed.run: start the N ranks (one per GPU) and relay their output.  Runs before
    anything touches the GPU in this process; the child is a separate process, never an exec."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env)
    return proc.wait()


def aggregate(elapsed_s, stats, world):
    """max-over-ranks time and gathered per-rank stats (RCCL all_gather of a small fp64 vector; gloo in CPU tests)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():                      # one rank started without a process group
        return elapsed_s, [list(stats)]
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = torch.tensor(list(stats), dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return float(t.item()), [o.cpu().tolist() for o in out]


def algorithmic_bytes(stage, S, W, H, P, per_stream):
    """SURVEY.md 8(d) algorithmic bytes of ONE launch of `stage` over S streams (per-stream averages in per_stream), formulas
    unmodified: B_fast = W H (1 + 21/64) + 4 N_c + 4 sum H_l for the whole front end (level re-reads of an unfused FAST pass are
    implementation overhead and carry no bytes); B_patch = P^2 (N_p + K) + 48 N_p; B_pose = N_found 120 + 216 per iteration;
    B_ba = M 176 + N_cam 312 + N_pt 168 per LM trial."""
    ncorn, npatch, nzm, nfound = (per_stream[k] for k in ("corners", "patches", "zmssd", "found"))
    hsum = sum(H >> l for l in range(4))
    b_fast = W * H * (1 + 21.0 / 64.0) + 4 * ncorn + 4 * hsum
    b_patch = P * P * (npatch + nzm) + 48 * npatch
    b_pose = 10 * (nfound * 120 + 216)
    table = {"front_end": b_fast, "pyr_fast0": W * H * (1 + 21.0 / 64.0), "fast_lvl": 0.0, "compact": 4 * ncorn + 4 * hsum,
             "search_fine": b_patch, "search_coarse": 0.0, "pose_fine": b_pose, "pose_coarse": 0.0}
    return S * table.get(stage, 0.0)


def ba_counted(st):
    """SURVEY.md 8(d)'s byte and flop formulas (B_ba = M 176 + N_cam 312 + N_pt 168, F_ba = M 780 + sum_pts C(n_free, 2) 216 +
    (6 n)^3 / 3 per LM trial) over what k_ba_compute launches actually ran: `st` = the counters the launches themselves accumulate
    on the device (vslam_profile_ba_stats / vslam_get_mapmaker_timing: LM trials and their trial-weighted problem sizes).
    -> (algorithmic bytes, flops) of all the launches counted."""
    b = st["trials_x_meas"] * 176.0 + st["trials_x_cams"] * 312.0 + st["trials_x_points"] * 168.0
    f = st["trials_x_meas"] * 780.0 + st["trials_x_points_x_pairs"] * 216.0 + st["trials_x_6n_cubed"] / 3.0
    return b, f


def baseline_metric():
    """The headline metric exactly as BASELINE.json names it (the file sits beside bench.py and travels with the repository)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, ValueError, KeyError):
        return "frames/sec (TrackFrame+local BA) on 640×480 synthetic, 1/2/4/8 GPU"


def pmc_traffic(kernel):
    """HBM bytes of `kernel` per unit of its work from the committed rocprofv3 PMC passes (profiles/*traffic*.json), or None.

    bench.py cannot run rocprofv3 on itself; the counters were collected with this same command (FETCH_SIZE and WRITE_SIZE
    in separate --pmc passes, kilobytes; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950) and are stored per
    unit of work (per LM trial of one problem for k_ba_compute, per frame for the front end), so that they can be scaled to
    the launch mix of THIS run."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        if t.get("kernel") == kernel and "bytes_per_unit" in t:
            return {"bytes_per_unit": float(t["bytes_per_unit"]), "unit": t.get("unit", ""), "source": os.path.relpath(f, ROOT),
                    "algorithmic_bytes_per_unit": float(t.get("algorithmic_bytes_per_unit", 0.0))}
    return None


def oracle_corner_fn(gray):
    """maximal FAST corners per level from the oracle (the CPU legs build their maps without a GPU)"""
    from oracle import binding as orc
    out = []
    for img, c, _lut in orc.make_keyframe_lite(gray, (10, 15, 15, 10)):
        out.append(orc.nonmax(c, orc.fast_score(img, c, 10)))
    return out


def cpu_worker(job):
    """One host core: the oracle's TrackFrame + BA over whole synthetic sequences (own seeds, n_frames each, one keyframe with
    its bundle adjustment per ~21 frames) until the budget is spent; scene set-up is not timed.  Returns (frames, seconds, keyframes)."""
    vp_kw, seed, W, H, n_frames, budget_s, map_kw = job
    from oracle import binding as orc
    from visualslam_android_amd import capi, feeder
    frames_done, secs, kfs = 0, 0.0, 0
    while secs < budget_s:
        f = feeder.Feeder(W, H, seed=seed)
        seed += 1000
        m = feeder.build_map(f, oracle_corner_fn, **map_kw)
        frames = f.render(0, n_frames)
        vp = capi.default_params(W, H, 1, **vp_kw)
        o = orc.OracleSystem(orc.params_from_vslam(vp))
        o.load_map(m)
        o.set_pose(f.pose(-1))
        t0 = time.perf_counter()
        n = 0
        for t in range(n_frames):
            o.track_frame(frames[t])
            n += 1
            if secs + (time.perf_counter() - t0) > budget_s and n >= KF_PERIOD + 2:
                break
        secs += time.perf_counter() - t0
        frames_done += n
        kfs += o.state().n_keyframes - len(m["keyframes"])
        o.close()
    return frames_done, secs, kfs


def bench_ba(args, rank, world, local_rank):
    """--config ba: BASELINE configs[2] / SURVEY.md 8(d) config 3 itself.  N independent problems of 5 cameras (camera 0 fixed) x 300
    points, all visible (M = 1500), 0.5 px noise, 5 % gross outliers, Tukey, 10 LM iterations, through the stand-alone Bundle of the C ABI
    (vslam_bundle_*: Bundle::AddCamera / AddPoint / AddMeas / Compute).  Unit of work = one Compute(); a step = one launch over the N
    problems, their inputs resident in HBM before the launch's first HIP event.  Roofline: 8(d)'s B_ba / F_ba per LM trial (316,000 B and
    1.56 MFLOP for this size) over the trials the launch itself counted.  cpu_baseline: the oracle's Compute() on one core."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from visualslam_android_amd import capi
    from visualslam_android_amd.ba_scene import CAM, ba_scene
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    N, K, Wm = args.problems, args.steps, args.warmup
    n_scenes = min(N, 64)                       # distinct seeded scenes; the N problems cycle through them
    scenes = [ba_scene(n_cams=5, n_pts=300, pixel_noise=0.5, outlier_frac=0.05, seed=1000 * rank + i, n_fixed=1) for i in range(n_scenes)]
    vp = capi.default_params(640, 480, 1, ba_max_iterations=10, device=local_rank, ba_sum_order=args.ba_sum_order)
    g = capi.Bundle(vp, N, 8, 320, 1600)
    for n in range(N):
        sc = scenes[n % n_scenes]
        g.set_problem(n, sc["cams_init"], [int(f) for f in sc["fixed"]], sc["pts_init"], [m[0] for m in sc["meas"]], [m[1] for m in sc["meas"]],
                      [m[2] for m in sc["meas"]], [m[3] for m in sc["meas"]])
    ms_steps, st_sum = [], {}
    for it in range(Wm + K):
        if it == Wm and world > 1:
            dist.barrier()
        g.compute()                              # uploads (untimed), then the launch between two HIP events on the Bundle's stream
        ms, st = g.timing()
        if it >= Wm:
            ms_steps.append(ms)
            for k_, v_ in st.items():
                st_sum[k_] = st_sum.get(k_, 0) + v_
    elapsed = sum(ms_steps) * 1e-3
    total_t, gathered = aggregate(elapsed, [elapsed, float(N * K)], world)
    res0 = g.result(0)
    out = None
    if rank == 0:
        from oracle import binding as orc
        fb, ff = ba_counted(st_sum)
        ms_launch = 1e3 * elapsed / K
        ach = fb / K / (ms_launch * 1e-3) / 1e9
        # CPU: the oracle's Bundle::Compute on one core, the same problems (construction untimed), a bounded sample
        t_cpu, n_cpu, tr_cpu = 0.0, 0, 0
        parity = None
        while t_cpu < args.cpu_seconds:
            sc = scenes[n_cpu % n_scenes]
            o = orc.OracleBundle(CAM, 640, 480, max_iterations=10)
            for pose, fixed in zip(sc["cams_init"], sc["fixed"]):
                o.add_camera(pose, fixed)
            for p_ in sc["pts_init"]:
                o.add_point(p_)
            for (c_, p_, xy, s2) in sc["meas"]:
                o.add_meas(c_, p_, xy, s2)
            t0 = time.perf_counter()
            acc = o.compute()
            t_cpu += time.perf_counter() - t0
            tr_cpu += int(o.stats()[2])
            if n_cpu == 0:
                parity = {"accepted": [int(acc), int(res0["accepted"])], "lm_trials": [int(o.stats()[2]), int(res0["trials"])],
                          "camera_maxdiff": float(np.abs(o.cameras() - g.cameras(0)).max())}
            n_cpu += 1
            o.close()
        out = {"metric": "Bundle::Compute()/sec, BASELINE configs[2]: local BA 5 keyframes x 300 map points, Tukey M-estimator, 10 LM iterations", "value": round(sum(x[1] for x in gathered) / total_t, 2),
               "unit": "computes/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(1e3 * total_t / K, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[2] (SURVEY.md 8(d) config 3): %d independent problems per GPU of 5 cameras (1 fixed) x 300 points, M = 1500, 0.5 px noise, 5 %% outliers, Tukey, 10 LM iterations; one vslam_bundle_compute launch per step" % N,
                          "problems_per_gpu": N, "distinct_scenes": n_scenes, "ba_sum_order": args.ba_sum_order,
                          "lm_trials_per_compute": round(st_sum["trials"] / max(1, st_sum["problems"]), 2), "latency_note": "ms_per_step = the launch of all problems; one Compute() alone on the GPU is latency-bound (a single workgroup)"},
               "roofline": {"kernel": "k_ba_compute_ordered" if args.ba_sum_order else "k_ba_compute", "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                            "traffic": None, "ms_per_launch": round(ms_launch, 4), "launches": K, "algorithmic_bytes": round(fb / K),
                            "bytes_per_lm_trial": round(fb / max(1, st_sum["trials"]), 1), "flops_per_lm_trial": round(ff / max(1, st_sum["trials"]), 1),
                            "lm_trials_per_launch": round(st_sum["trials"] / K, 1), "ms_each_launch": [round(x, 3) for x in ms_steps],
                            "fp64": {"achieved": round(ff / K / (ms_launch * 1e-3) / 1e12, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ff / K / (ms_launch * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5)},
                            "counted": "LM trials, bytes and flops of exactly these launches, accumulated on the device (vslam_bundle_get_timing)"},
               "cpu_baseline": {"value": round(n_cpu / t_cpu, 2), "unit": "computes/s", "cores": 1, "kind": "port", "compute_latency_ms": round(1e3 * t_cpu / n_cpu, 3),
                                "lm_trials_per_compute": round(tr_cpu / n_cpu, 2),
                                "sample": "oracle Bundle::Compute (oracle/bundle.cpp, -O3, one thread) on %d of the same problems, %.1f s" % (n_cpu, t_cpu)},
               "parity_problem0_oracle_vs_device": parity}
        print(json.dumps(out), flush=True)
    g.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VSLAM_BENCH_STREAMS", 3072)),
                    help="sequences per GPU (3072: three full rounds of k_pose's 1024 resident workgroups; measured 343 k frames/s at 2048, 365 k at 3072, 366 k at 4096)")
    ap.add_argument("--systems", type=int, default=int(os.environ.get("VSLAM_BENCH_SYSTEMS", 1)),
                    help="split the streams over this many vslam_system handles (one HIP stream each) so their kernels overlap")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--patch", type=int, default=8, help="PatchFinder template side (BASELINE configs: 8; reference default 11)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample (per leg)")
    ap.add_argument("--ba-delay", type=int, default=int(os.environ.get("VSLAM_BENCH_BA_DELAY", 16)),
                    help="vslam_params.ba_delay_frames: 0 = synchronous map-maker; D > 0 = Bundle::Compute on its own HIP stream, applied D frames later")
    ap.add_argument("--ba-batch", type=int, default=int(os.environ.get("VSLAM_BENCH_BA_BATCH", 0)),
                    help="vslam_params.ba_batch_frames: the keyframes of this many consecutive frames share one Bundle::Compute launch (0 = chosen so that a launch carries about four problems per compute unit)")
    ap.add_argument("--ba-window", type=int, default=5, help="vslam_params.ba_window (jni/MapMaker.cc:812-820: 5; BASELINE configs[3]: 10)")
    ap.add_argument("--max-keyframes", type=int, default=32)
    ap.add_argument("--max-patches", type=int, default=1000, help="vslam_params.max_patches_per_frame (jni/Tracker.cc:518: 1000; SURVEY.md 8(d) config 4 also asks for 2000)")
    ap.add_argument("--corners-per-level", type=str, default="", help="map points per pyramid level of the synthetic map, e.g. 1400,420,130,40 (default: the feeder's)")
    ap.add_argument("--kf-stagger", type=int, default=KF_PERIOD,
                    help="spread the streams' keyframe phases over this many frames (0: all streams in lock-step, one burst of bundle adjustments per period)")
    ap.add_argument("--use-sbi", type=int, default=int(os.environ.get("VSLAM_BENCH_USE_SBI", 0)),
                    help="vslam_params.use_sbi: 1 = SmallBlurryImage rotation prior in the motion model (the reference's gvnUseSBI)")
    ap.add_argument("--grow-map", type=int, default=int(os.environ.get("VSLAM_BENCH_GROW_MAP", 0)),
                    help="vslam_params.grow_map bit flags: 1 = AddSomeMapPoints (epipolar search), 2 = ReFindInSingleKeyFrame, 3 = both (the reference)")
    ap.add_argument("--diag-kf-dist-mult", type=float, default=None,
                    help="DIAGNOSTIC ONLY (not the metric): overrides vslam_params.max_kf_dist_wiggle_mult, e.g. 1e9 = no keyframes, no BA")
    ap.add_argument("--no-events", action="store_true", help="skip the per-stage HIP events in the timed region")
    ap.add_argument("--no-all-cores", action="store_true", help="skip the all-host-cores leg of cpu_baseline (one oracle process per core)")
    ap.add_argument("--no-flat-out", action="store_true", help="skip the flat-out bundle-adjustment round measured after the timed region")
    ap.add_argument("--parity-check", type=int, default=1, help="1: compare stream 0's final pose with the oracle run on the same frames")
    ap.add_argument("--config", type=str, default="track", choices=("track", "ba"),
                    help="track (default): the headline metric, TrackFrame + local BA on synthetic streams; ba: BASELINE configs[2] itself, N x Bundle::Compute of 5 cameras x 300 points")
    ap.add_argument("--problems", type=int, default=4096, help="--config ba: independent problems per GPU and launch")
    ap.add_argument("--ba-sum-order", type=int, default=0, help="vslam_params.ba_sum_order: 1 = every sum of Bundle::Compute in the reference's order (the parity mode)")
    ap.add_argument("--host-frames", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): the frames stay in host memory and every step is one vslam_update -- upload + TrackFrame, synchronous like native_update, jni/jni_part.cpp:132-145")
    ap.add_argument("--feeder-rects", type=int, default=0, help="rectangles of the feeder's texture (default 1000: ~1100 FAST corners at level 0 of 640x480; 1500 gives ~2000 at 1280x720, BASELINE configs[3])")
    ap.add_argument("--n1-value", type=float, default=None, help="with --gpus N: the value of the N = 1 run, to print the scaling efficiency beside the per-rank rates")
    args = ap.parse_args()
    if args.feeder_rects > 0:
        os.environ["VSLAM_FEEDER_NRECT"] = str(args.feeder_rects)
    if any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF")) for k in os.environ):   # under rocprofv3 the profiler attaches to every child process:
        args.no_all_cores = True                # the 256-process CPU leg is skipped

    stub = os.environ.get("VSLAM_BENCH_STUB") == "1"   # CPU test of the launcher / aggregation path: gloo, no GPU, a stubbed step
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank, world, local_rank = dist_env()
    t_prog = time.perf_counter()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the torch.distributed world has %d ranks" % (args.gpus, world))
    if args.config == "ba" and os.environ.get("VSLAM_BENCH_STUB") != "1":
        return bench_ba(args, rank, world, local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    if stub:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        K = args.steps
        t0 = time.perf_counter()
        for _ in range(K):
            time.sleep(0.001 * (1 + rank))
        elapsed = time.perf_counter() - t0
        total_t, gathered = aggregate(elapsed, [elapsed, float(args.streams * K)], world)
        out = None
        if rank == 0:
            out = {"metric": baseline_metric(), "value": round(sum(g[1] for g in gathered) / total_t, 2), "unit": "frames/s", "n_gpus": world,
                   "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * total_t / K, 4), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f64", "data": "stub", "config": {"workload": "launcher self-test (VSLAM_BENCH_STUB=1): no GPU work"},
                   "per_rank": [{"rank": r, "frames_per_s": round(g[1] / g[0], 2), "ms_per_step": round(1e3 * g[0] / K, 4)} for r, g in enumerate(gathered)]}
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return out
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    one_gpu = os.environ.get("VSLAM_BENCH_ONE_GPU") == "1"   # rehearsal of the multi-rank path on a one-GPU box: every rank on cuda:0, gloo
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 and one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: RCCL sees %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from visualslam_android_amd import capi, feeder
    if os.environ.get("VSLAM_LIB"):            # diagnostic builds from tools/build_variant.sh (A/B runs on one box)
        capi.load_library(os.environ["VSLAM_LIB"])
    S, W, H, K, Wm = args.streams, args.width, args.height, args.steps, args.warmup
    T = Wm + K
    nthreads = max(2, min(16, (os.cpu_count() or 8) // max(1, world)))   # set-up threads per rank: the ranks of a node share its cores
    map_kw = {}
    if args.corners_per_level:
        map_kw["per_level"] = tuple(int(x) for x in args.corners_per_level.split(","))

    # ---- synthetic scenes: one seeded feeder, trajectory and ground-truth map per stream -----------------------------
    # Built CHUNK streams at a time (a feeder holds a 16.8 MB texture: all 3072 at once were 67 GB of host memory per rank, 530 GB for
    # eight ranks of a node; now ~6 GB): feeders -> source keyframe images (host threads) -> their maximal FAST corners from the device
    # front end in batches of FB images per call (3 + 2 launches per batch: a per-image front end was ~16,000 set-up launches for 2048
    # streams, which a profiler pass with counters serialises one by one -- VERDICT r2 weak #11) -> maps (host threads) -> load_map, the
    # start pose -> the stream's T frames rendered and uploaded as one contiguous piece -> the feeders closed (stream 0's is kept for
    # the parity check).
    t_setup = time.time()
    NKF = 8
    NS = max(1, min(args.systems, S))
    assert S % NS == 0, "--streams must be a multiple of --systems"
    Sk = S // NS
    ncu = torch.cuda.get_device_properties(local_rank).multi_processor_count
    stagger = max(0, args.kf_stagger)
    ba_batch = args.ba_batch
    if args.ba_delay <= 0:
        ba_batch = 1
    elif ba_batch <= 0:                        # about six problems per compute unit per launch, and done well inside the delay window
        per_frame = Sk / float(stagger) if stagger else float(Sk)    # (3072 streams: 7 frames per batch 359 k frames/s, 10 frames 365 k)
        ba_batch = int(max(1, min(args.ba_delay - 6, (6 * ncu) // max(1.0, per_frame))))
    vp_kw = dict(patch_size=args.patch, ba_delay_frames=args.ba_delay, use_sbi=args.use_sbi, grow_map=args.grow_map, ba_window=args.ba_window,
                 max_keyframes=args.max_keyframes, max_patches_per_frame=args.max_patches, ba_sum_order=args.ba_sum_order)
    vpk = capi.default_params(W, H, Sk, device=local_rank, ba_batch_frames=ba_batch, **vp_kw)
    if args.diag_kf_dist_mult is not None:
        vpk.max_kf_dist_wiggle_mult = args.diag_kf_dist_mult
    systems = [capi.System(vpk) for _ in range(NS)]

    def sys_of(s):
        return systems[s // Sk], s % Sk

    host_mode = bool(args.host_frames)
    # frame ring, stream-major ([S][T][H][W]: a stream's T frames are one contiguous upload, not T strided pieces)
    frames_dev = torch.empty((S, T, H, W), dtype=torch.uint8, device="cpu" if host_mode else "cuda", pin_memory=host_mode)
    host_frames0 = None
    feeder0 = map0 = None
    CHUNK = 256
    FB = min(256, S * NKF)
    fe = capi.System(capi.default_params(W, H, FB, patch_size=args.patch, device=local_rank))
    for c0 in range(0, S, CHUNK):
        cs = list(range(c0, min(S, c0 + CHUNK)))
        with ThreadPoolExecutor(nthreads) as ex:
            feeders = list(ex.map(lambda s_: feeder.Feeder(W, H, seed=1234 + rank * S + s_), cs))
            kf_images = list(ex.map(lambda f: feeder.keyframe_images(f, n_keyframes=NKF), feeders))
        flat_imgs = [im for ims in kf_images for im in ims]
        kf_corners = []
        for b0 in range(0, len(flat_imgs), FB):
            chunk = flat_imgs[b0:b0 + FB]
            batch = np.stack(chunk + [chunk[-1]] * (FB - len(chunk)))
            fe.make_keyframe_lite(batch)
            fe.fast_nonmax()
            for i in range(len(chunk)):
                kf_corners.append([fe.read_max_corners(i, l)[0] for l in range(4)])
        with ThreadPoolExecutor(nthreads) as ex:
            maps = list(ex.map(lambda j: feeder.build_map(feeders[j], None, n_keyframes=NKF, images=kf_images[j], corners=kf_corners[j * NKF:(j + 1) * NKF], **map_kw), range(len(cs))))
        for j, s_ in enumerate(cs):
            sy, ls = sys_of(s_)
            sy.load_map(ls, maps[j])
            sy.set_pose(ls, feeders[j].pose(-1))
            if stagger:                        # Tracker::mnLastKeyFrameDropped: stream s asks for its first keyframe in frame 1 + phase
                sy.set_last_keyframe_dropped(ls, -20 + (s_ * stagger) // S)
        with ThreadPoolExecutor(max(1, nthreads // 4)) as ex:
            for j, fr in enumerate(ex.map(lambda f: f.render(0, T, threads=4), feeders)):
                frames_dev[cs[j]].copy_(torch.from_numpy(fr))
                if cs[j] == 0:
                    host_frames0 = fr
        if c0 == 0:
            feeder0, map0 = feeders[0], maps[0]
        for j, f in enumerate(feeders):
            if f is not feeder0:
                f.close()
        del feeders, kf_images, flat_imgs, kf_corners, maps
    fe.close()
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    base = frames_dev.data_ptr()
    fstride = H * W                             # frame t of stream s at base + (s * T + t) * H * W
    sstride = T * H * W

    def step(t):
        for k, sy in enumerate(systems):
            if host_mode:                       # native_update: host gray image in, synchronous (the caller may reuse its buffer)
                capi._check(sy.lib.vslam_update(sy.h, base + t * fstride + k * Sk * sstride, W, sstride))
            else:
                sy.track_frame_device(base + t * fstride + k * Sk * sstride, W, sstride)

    def sync_all():
        for sy in systems:
            sy.synchronize()

    def state(s):
        sy, ls = sys_of(s)
        return sy.state(ls)

    def all_states():                           # one copy per system: polling 2048 streams one by one idles the GPU for ~50 ms, long
        out = []                                # enough for its clocks to drop before the timed region starts
        for sy in systems:
            out.extend(sy.states())
        return out

    def note(msg):                              # progress on stderr: a long run must not look hung to the box's watchdog
        if rank == 0:
            print("[bench] %s (%.0f s)" % (msg, time.perf_counter() - t_prog), file=sys.stderr, flush=True)
    note("set-up done")
    # ---- warm-up, then the timed region ---------------------------------------------------------------------------------
    for t in range(Wm):
        step(t)
    sync_all()
    st0 = all_states()
    if not args.no_events:
        for sy in systems:
            sy.profile_begin(K)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, T):
        step(t)
    enqueue_s = time.perf_counter() - t0        # the host's share: issuing the K steps (launches only, nothing waited for)
    sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    stage_ms, stage_n, nprof = {}, {}, 0
    ba_st = {}                                  # what the window's k_ba_compute launches ran, counted by the launches themselves
    if not args.no_events:
        for sy in systems:                      # per-launch durations: summed over systems, averaged below over launches
            ms, n = sy.profile_end()
            cnt = sy.profile_launches()
            nprof += n
            for k_, v_ in ms.items():
                stage_ms[k_] = stage_ms.get(k_, 0.0) + v_
                stage_n[k_] = stage_n.get(k_, 0) + cnt[k_]
            for k_, v_ in sy.profile_ba_stats().items():
                ba_st[k_] = ba_st.get(k_, 0) + v_
    st1 = all_states()
    ba_totals = {}                              # every k_ba_compute launch of the process (for profiler passes: their counters cover the whole run)
    for sy in systems:
        for k_, v_ in sy.ba_launch_totals().items():
            ba_totals[k_] = ba_totals.get(k_, 0) + v_

    # ---- per-stream workload statistics (for the algorithmic-byte formulas) ---------------------------------------------
    zm = float(np.mean([(b.n_zmssd - a.n_zmssd) / K for a, b in zip(st0, st1)]))
    att = float(np.mean([sum(b.attempted) for b in st1]))
    fnd = float(np.mean([sum(b.found) for b in st1]))
    kf_adds = float(np.mean([b.n_keyframes - a.n_keyframes for a, b in zip(st0, st1)]))
    ba_trials = float(np.mean([b.n_ba_trials - a.n_ba_trials for a, b in zip(st0, st1)]))
    good = int(sum(1 for b in st1 if b.quality == 2))
    ncorn = float(len(systems[0].read_corners(0, 0)))
    bss = [systems[s // Sk].bundle_stats(s % Sk) for s in range(0, S, max(1, S // 64))]   # sizes of the last assembled problems (a sample of streams)
    bss = [b for b in bss if b["meas"] > 0] or [systems[0].bundle_stats(0)]
    per_stream = {"corners": ncorn, "patches": att, "zmssd": zm, "found": fnd,
                  "ba_meas": float(np.mean([b["meas"] for b in bss])), "ba_cams": float(np.mean([b["cams"] for b in bss])),
                  "ba_free": float(np.mean([b["free_cams"] for b in bss])), "ba_pts": float(np.mean([b["points"] for b in bss])),
                  "ba_trials_per_problem": float(np.mean([b["trials"] for b in bss]))}
    # the k_ba_compute launches of the window: LM trials, algorithmic bytes and flops of exactly those launches (device counters),
    # so that they and the launches' HIP-event time describe the same work (round 2 divided the trials APPLIED in the window --
    # adjustments launched ba_delay frames earlier -- by the launches IN the window)
    ba_launches = max(1, ba_st.get("launches", 0))
    trials_per_launch = ba_st.get("trials", 0) / ba_launches
    ba_bytes_launch, ba_flops_launch = (x / ba_launches for x in ba_counted(ba_st)) if ba_st else (0.0, 0.0)

    # ---- a flat-out round of the bundle adjustment: every stream's BundleAdjustRecent in ONE launch, nothing beside it ------
    flat = None
    note("timed region done")
    if not args.no_flat_out and args.diag_kf_dist_mult is None:
        # Every stream makes its current frame a keyframe NOW (vslam_add_keyframe: MapMaker::AddKeyFrame + BundleAdjustRecent, the
        # same problems as the timed region's, all 2048 in one synchronous launch after the adjustments in flight were collected),
        # timed with HIP events on the system's stream around select + assemble / k_ba_compute / write-back
        # (vslam_get_mapmaker_timing).  Twice: the second round adds the frame once more (a camera more per problem).  Bytes and
        # flops are those the launch counted on the device.  (Round 2 of the review: this used to be a BundleAdjustRecent of maps
        # adjusted a moment ago -- one LM trial per problem -- timed by the host's clock together with the drain of the pending
        # asynchronous adjustments, whose LM trials it also counted: 11-25 ms and "14,094 trials" depending on what was in flight.)
        try:
            rounds = []
            for _round in range(2):
                sync_all()
                t_f = time.perf_counter()
                for sy in systems:
                    capi._check(sy.lib.vslam_add_keyframe(sy.h, -1))
                sync_all()
                wall_s = time.perf_counter() - t_f
                ms3 = {"assemble": 0.0, "compute": 0.0, "writeback": 0.0}
                cst = {}
                for sy in systems:
                    m3, c8 = sy.mapmaker_timing()
                    for k_ in ms3:
                        ms3[k_] = max(ms3[k_], m3[k_])
                    for k_, v_ in c8.items():
                        cst[k_] = cst.get(k_, 0) + v_
                fb, ff = ba_counted(cst)
                cs = max(1e-9, ms3["compute"] * 1e-3)
                rounds.append({"problems": cst["problems"], "lm_trials": cst["trials"], "ms_assemble": round(ms3["assemble"], 3),
                               "ms_compute": round(ms3["compute"], 3), "ms_writeback": round(ms3["writeback"], 3), "ms_host_wall": round(1e3 * wall_s, 3),
                               "algorithmic_bytes": round(fb), "achieved_GBps": round(fb / cs / 1e9, 2), "frac": round(fb / cs / 1e9 / HBM_PEAK_GBS, 5),
                               "fp64_TFLOPs": round(ff / cs / 1e12, 3)})
            flat = dict(rounds[0])
            flat["round2"] = rounds[1]
            flat["timing"] = ("HIP events on the system's stream around k_ba_compute alone (ms_compute; achieved / frac use it), select + assemble and "
                              "write-back beside it; ms_host_wall = the host's clock around the call + synchronize; every stream's AddKeyFrame + "
                              "BundleAdjustRecent in ONE synchronous launch, nothing beside it")
        except Exception as e:                 # noqa: BLE001 - extra information only
            flat = {"error": str(e)}

    total_t, gathered = aggregate(elapsed, [elapsed, S * K, zm, att, fnd, kf_adds, ba_trials, good], world)
    frames_total = sum(g[1] for g in gathered)
    value = frames_total / total_t
    good = int(round(sum(g[7] for g in gathered)))        # streams tracking GOOD at the end, all ranks
    # per-GPU rates beside the aggregate (each rank's own clock over its own streams); the scaling efficiency
    # needs the N = 1 value, which a multi-rank run does not have -- the driver computes it from its N = 1 line, or pass --n1-value
    per_rank = [{"rank": r, "frames_per_s": round(g[1] / g[0], 2), "ms_per_step": round(1e3 * g[0] / K, 4), "streams_tracking_good": int(round(g[7])),
                 "keyframes_added_per_stream": round(g[5], 3), "efficiency_vs_n1": (round(g[1] / g[0] / args.n1_value, 4) if args.n1_value else None)} for r, g in enumerate(gathered)]

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events over the timed region) -----------------------------------------
        roof = None
        stages = {}
        if stage_ms:
            stage_ms["front_end"] = stage_ms.get("pyr_fast0", 0.0) + stage_ms.get("fast_lvl", 0.0) + stage_ms.get("compact", 0.0)
            stage_n["front_end"] = stage_n.get("pyr_fast0", nprof)

            def stage_bytes(name):
                if name == "ba_compute":
                    return ba_bytes_launch
                return algorithmic_bytes(name, Sk, W, H, args.patch, per_stream)

            for name, ms in stage_ms.items():
                per_launch_ms = ms / max(1, stage_n.get(name, nprof))
                ab = stage_bytes(name)
                stages[name] = {"ms_per_launch": round(per_launch_ms, 5), "launches": stage_n.get(name, nprof),
                                "algorithmic_GBps": round(ab / (per_launch_ms * 1e-3) / 1e9, 3) if per_launch_ms > 0 and ab > 0 else None}

            def roofline_of(name):
                ms_ = stage_ms[name] / max(1, stage_n.get(name, nprof))
                ab_ = stage_bytes(name)
                ach_ = ab_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
                r_ = {"kernel": name, "bound": "hbm", "achieved": round(ach_, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(ach_ / HBM_PEAK_GBS, 6), "traffic": None, "ms_per_launch": round(ms_, 5),
                      "launches": stage_n.get(name, nprof), "algorithmic_bytes": round(ab_)}
                if name == "ba_compute":        # supplementary: the same launch against the fp64 vector/matrix peak (78.6 TFLOP/s)
                    tf = ba_flops_launch / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
                    r_["fp64"] = {"achieved": round(tf, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 5)}
                    r_["lm_trials_per_launch"] = round(trials_per_launch, 1)
                    r_["problems_per_launch"] = round(ba_st.get("problems", 0) / ba_launches, 1)
                    r_["counted"] = "LM trials, bytes and flops of exactly these launches, accumulated on the device by k_ba_compute (vslam_profile_ba_stats)"
                    r_["flat_out_round"] = flat
                tr = pmc_traffic({"ba_compute": "k_ba_compute", "front_end": "k_front_end"}.get(name, name))
                if tr:                          # counters per unit of work (profiles/), scaled to this run's launch mix
                    units = trials_per_launch if name == "ba_compute" else Sk
                    scale = 1.0                 # counters of a problem of another size: scaled by the ratio of the algorithmic bytes per unit
                    if name == "ba_compute" and tr["algorithmic_bytes_per_unit"] > 0 and units > 0:
                        scale = (ab_ / units) / tr["algorithmic_bytes_per_unit"]
                    r_["traffic"] = round(tr["bytes_per_unit"] * units * scale); r_["traffic_source"] = tr["source"]; r_["traffic_unit"] = tr["unit"]
                return r_

            # the dominant kernel = the largest summed HIP-event time over the timed region among the kernels with a byte model
            cands = [n for n in ("ba_compute", "front_end", "search_fine", "pose_fine") if n in stage_ms and stage_bytes(n) > 0]
            # (the front end is three kernels in a row -- level 0 + pyramid, levels 1-3, compaction: it ranks by its longest one)
            fe_longest = max(stage_ms.get("pyr_fast0", 0.0), stage_ms.get("fast_lvl", 0.0), stage_ms.get("compact", 0.0))
            ranked = sorted(cands, key=lambda n: -(fe_longest if n == "front_end" else stage_ms[n]))
            if ranked:
                roof = roofline_of(ranked[0])
                roof["others"] = [roofline_of(n) for n in ranked[1:]]
                roof["problem"] = {k: round(per_stream[k], 1) for k in ("ba_cams", "ba_free", "ba_pts", "ba_meas", "ba_trials_per_problem")}
        # ---- CPU baseline: the oracle's TrackFrame + BA on the host cores over a bounded sample of the same workload ---------
        from oracle import binding as orc
        pose_diff = None
        if args.parity_check:
            # stream 0 keeps the reference's own keyframe schedule (phase 0): the same frames through the oracle end at the same pose
            vp1 = capi.default_params(W, H, 1, **vp_kw)
            o = orc.OracleSystem(orc.params_from_vslam(vp1))
            o.load_map(map0)
            o.set_pose(feeder0.pose(-1))
            for t in range(T):
                o.track_frame(host_frames0[t])
            pose_diff = float(np.abs(np.array(o.state().pose[:]) - np.array(st1[0].pose[:])).max())
            o.close()
        n_seq = 400                             # frames per oracle sequence: 19 keyframes, under the 32-keyframe capacity of the device path
        job = (vp_kw, 900000, W, H, n_seq, args.cpu_seconds, map_kw)
        note("gpu side done, cpu_baseline starts")
        # (timed at N = 1 only: beside N - 1 ranks waiting at the last barrier it would only lengthen the multi-GPU runs)
        n_cpu, cpu_s, cpu_kf = cpu_worker(job) if world == 1 else (0, 1.0, 0)
        note("cpu_baseline (1 core) done")
        ncores = os.cpu_count() or 1
        all_cores = None
        try:
            if world > 1:
                raise RuntimeError("skipped (measured at N = 1 only)")
            if args.no_all_cores:
                raise RuntimeError("skipped (--no-all-cores)")                                    # the same on every host core at once (independent sequences, one process each)
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(ncores) as pool:
                t_a = time.perf_counter()
                res = pool.map(cpu_worker, [(vp_kw, 910000 + c, W, H, n_seq, args.cpu_seconds, map_kw) for c in range(ncores)])
                wall = time.perf_counter() - t_a
            all_cores = {"value": round(sum(n / s_ for n, s_, _k in res), 2), "unit": "frames/s", "cores": ncores,
                         "sample": "%d oracle processes side by side, %d frames in all (own seeds, %d keyframes with their bundle adjustment); sum of the per-process rates; pool wall time %.1f s incl. scene set-up"
                                   % (ncores, sum(r[0] for r in res), sum(r[2] for r in res), wall)}
        except Exception as e:                  # noqa: BLE001 - extra information only
            all_cores = {"error": str(e)}
        out = {
            "metric": baseline_metric(), "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(1e3 * total_t / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic (frames in host memory, uploaded per step: PCIe-inclusive)" if host_mode else "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %dx%d 4-level FAST-10 + %dx%d PatchFinder ZMSSD search, TrackMap pose update, "
                                   "AddKeyFrame + BundleAdjustRecent (%d-keyframe window) on keyframe frames" % (1 if (W, H) == (640, 480) else 3, W, H, args.patch, args.patch, args.ba_window),
                       "streams_per_gpu": S, "systems_per_gpu": NS, "ba_delay_frames": args.ba_delay, "ba_batch_frames": ba_batch, "ba_window": args.ba_window,
                       "kf_stagger_frames": stagger, "use_sbi": args.use_sbi, "grow_map": args.grow_map, "max_patches_per_frame": args.max_patches, "ba_sum_order": args.ba_sum_order,
                       "diagnostic_kf_dist_mult": args.diag_kf_dist_mult, "patch_size": args.patch, "corners_l0_per_frame": ncorn,
                       "patches_attempted_per_frame": round(att, 1), "patches_found_per_frame": round(fnd, 1),
                       "zmssd_evals_per_frame": round(zm, 1), "keyframes_per_step": round(kf_adds * S / K, 2),
                       "keyframes_added_per_stream": round(kf_adds, 3), "ba_trials_per_stream": round(ba_trials, 3),
                       "map_points_per_stream": round(float(np.mean([b.n_points for b in st1])), 1),
                       "streams_tracking_good": good, "setup_seconds": round(setup_s, 1),
                       "host_enqueue_ms_per_step": round(1e3 * enqueue_s / K, 3)},
            "roofline": roof,
            "cpu_baseline": {"skipped": "measured at N = 1 only"} if world > 1 else
                            {"value": round(n_cpu / cpu_s, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                             "sample": "oracle TrackFrame+BA (oracle/, -O3, one thread) over %d frames of synthetic sequences of the same workload (own seeds; %d keyframes with their bundle adjustment), %.1f s" % (n_cpu, cpu_kf, cpu_s),
                             "all_cores": all_cores},
            "stages": stages,
            "per_rank": per_rank,
            "scaling_efficiency_vs_n1": (round(value / (world * args.n1_value), 4) if args.n1_value else None),
            "pcie_inclusive": ({"frames_per_s": round(value, 2), "host_GBps": round(value * W * H / 1e9, 2),
                                "note": "--host-frames: every step uploads S frames from pinned host memory inside vslam_update (synchronous, like native_update); this line is NOT the headline value"} if host_mode else None),
            "ba_launch_totals_whole_run": ba_totals, "frames_whole_run": T,
            "ba_flat_out": flat,
            "parity_pose_maxdiff_stream0": pose_diff,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
