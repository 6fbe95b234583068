"""CPU test of the N > 1 path of bench.py: max-over-ranks time + stats gather with torch.distributed (gloo, 2 ranks)."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed = 1.0 + rank                      # rank 1 is the slow one
    t, gathered = bench.aggregate(elapsed, [elapsed, 100.0 * (rank + 1), 7.0 + rank], world)
    q.put((rank, t, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregate_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, gathered in res:
        assert t == 2.0                                           # max over ranks
        assert gathered == [[1.0, 100.0, 7.0], [2.0, 200.0, 8.0]]  # every rank sees every rank's stats, in rank order
    frames = sum(g[1] for g in res[0][2])
    assert frames / res[0][1] == 150.0                            # whole-job value = all ranks' units / max time


def test_aggregate_single_process():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(0.5, [1, 2], 1) == (0.5, [[1, 2]])
    ps = {"corners": 1000, "patches": 1000, "zmssd": 8000, "found": 500, "ba_meas": 1500, "ba_cams": 5, "ba_free": 4, "ba_pts": 300}
    assert bench.algorithmic_bytes("front_end", 2, 640, 480, 8, ps) == 2 * 415600     # 408,000 + 4,000 + 3,600 per frame
    assert bench.algorithmic_bytes("search_fine", 1, 640, 480, 8, ps) == 624000      # 8(d) example: P = 8, N_p = 1000, K = 8 per patch
    # 8(d) config 3 through the device counters' form: one problem, one LM trial, 1500 measurements, 5 cameras (4 adjustable), 300 points
    st = {"trials": 1, "trials_x_meas": 1500, "trials_x_cams": 5, "trials_x_points": 300, "trials_x_points_x_pairs": 300 * 6, "trials_x_6n_cubed": 24 ** 3}
    b, f = bench.ba_counted(st)
    assert b == 315960                                                                # 1500 * 176 + 5 * 312 + 300 * 168 (quoted there as 316,000)
    assert abs(f - 1.56e6) < 0.01e6                                                   # ~1.56 MFLOP per trial


def _run_bench(args, env_extra):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, [json.loads(l) for l in lines]


def test_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment: bench.py starts the two ranks (torch.distributed.run),
    they rendezvous (gloo here: VSLAM_BENCH_STUB=1 replaces the GPU step by a sleep), and rank 0's ONE line says n_gpus 2 with
    the whole-job value = all ranks' frames / the slowest rank's time."""
    out, lines = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "1", "--streams", "10"], {"VSLAM_BENCH_STUB": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["data"] == "stub"
    assert abs(d["value"] - 2 * 10 * 20 / (d["ms_per_step"] * 20e-3)) < 0.01 * d["value"]
    assert d["ms_per_step"] >= 2.0                                  # rank 1 sleeps 2 ms per step: the max over ranks counts
    pr = d["per_rank"]                                              # the per-GPU rates beside the aggregate
    assert [x["rank"] for x in pr] == [0, 1] and pr[0]["frames_per_s"] > 1.5 * pr[1]["frames_per_s"] > 0    # rank 0 sleeps 1 ms per step


def test_world_size_must_equal_gpus():
    """Started with a world of one while --gpus says 2 (what a broken launcher would do): refuse, do not report n_gpus 1."""
    out, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"VSLAM_BENCH_STUB": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and not lines
    assert "--gpus 2" in (out.stderr + out.stdout)
