"""bench.py's cross-rank aggregation through RCCL itself (backend "nccl") on the one GPU of the box: a one-rank process group runs the
very collectives of the N-GPU line -- all_reduce(MAX) of the fp64 time, all_gather of the fp64 per-rank stats, the barriers around the
timed region -- on device tensors.  (Two ranks cannot share one GPU under RCCL; the two-rank path is rehearsed over gloo in
test_gpu_bench_ranks.py and test_bench_dist.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
dist.barrier()
torch.cuda.synchronize()
t, per = bench.aggregate(1.25, [3.0, 4.5, 1e-13, 2.0 ** 53 + 2], 1)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"t": t, "per": per}))
"""


def test_aggregate_over_rccl_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["t"] == 1.25 and d["per"] == [[3.0, 4.5, 1e-13, 2.0 ** 53 + 2]]          # fp64 end to end: nothing rounded through fp32
