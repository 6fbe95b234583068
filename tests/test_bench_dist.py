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
    b = bench.algorithmic_bytes("fast_stage", 2, 640, 480, 8, {"corners": 1000, "patches": 0, "zmssd": 0, "found": 0, "ba_meas": 0,
                                                               "ba_cams": 0, "ba_pts": 0, "ba_trials_per_launch": 0})
    assert b == 2 * 415600                                        # 408,000 + 4,000 + 3,600 per frame
