"""bench.py --gpus 2 end to end on a one-GPU box: the launcher starts two ranks (torch.distributed.run, 127.0.0.1), both on cuda:0
with gloo for the barrier and the gather (VSLAM_BENCH_ONE_GPU=1 -- the multi-GPU runs themselves use RCCL, one rank per GPU), each
rank tracks its own streams through the HIP library, rank 0 prints ONE line with n_gpus = 2 and the frames of both ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, VSLAM_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "64", "--steps", "12", "--warmup", "3",
           "--no-flat-out", "--cpu-seconds", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak"
    assert d["config"]["streams_per_gpu"] == 64 and d["config"]["streams_tracking_good"] == 128     # both ranks' streams, all tracking
    assert abs(d["value"] - 2 * 64 * 12 / (d["ms_per_step"] * 12e-3)) <= 1e-3 * d["value"]          # whole-job frames / max-over-ranks time
    assert "skipped" in d["cpu_baseline"]                                                           # the CPU legs are timed at N = 1 only
    assert d["roofline"] is not None and d["roofline"]["kernel"]
