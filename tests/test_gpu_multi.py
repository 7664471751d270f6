"""GPU, two or more devices on the node (skipped on the one-GPU boxes the round's tests run on; the first multi-GPU box
exercises it): the C-ABI collective with TWO ranks — lde_comm_unique_id on rank 0, lde_comm_init on both, an in-place f32 sum of
rank-dependent data, checked against the closed form on every rank — and bench.py's own launcher at --gpus 2 on the metric
workload and on c3 (whose dW goes through lde_comm_allreduce_f32)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch
    return torch.cuda.device_count()


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["LDE_ROOT"])
import torch, torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("gloo")                      # only carries the 128-byte id: the sum itself is lde_comm over RCCL
from latentdiffeq_amd.dist import LdeComm
from latentdiffeq_amd import _lib as L
c = LdeComm(rank, world)
lib = L.load()
assert lib.lde_comm_nranks(c.handle) == world and lib.lde_comm_rank(c.handle) == rank
n = 24864                                            # c4's weight count
x = (torch.arange(n, device="cuda", dtype=torch.float32) % 97) * (rank + 1) + rank
for _ in range(3):                                   # repeated use of one communicator
    y = x.clone()
    c.allreduce_(y)
    torch.cuda.synchronize()
    want = (torch.arange(n, device="cuda", dtype=torch.float32) % 97) * (world * (world + 1) / 2) + world * (world - 1) / 2
    assert torch.equal(y, want), (rank, float((y - want).abs().max()))
c.close()
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs on the node")
def test_lde_comm_two_ranks(tmp_path):
    w = tmp_path / "comm_worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, LDE_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", str(w)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs on the node")
@pytest.mark.parametrize("workload", ["goku_pendulum", "c3"])
def test_bench_two_gpus(workload):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--workload", workload],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0          # every line carries the CPU baseline, N > 1 too
    if workload == "c3":
        assert "lde_comm_allreduce_f32" in d["config"]["parallelism"]
    else:   # the metric line: the strong-scaling figure (global batch 256 split over the GPUs) rides beside the weak one
        assert d["strong_scaling"]["global_batch"] == 256 and d["strong_scaling"]["batch_per_gpu"] == 128 and d["strong_scaling"]["value"] > 0


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs on the node")
def test_bench_two_gpus_c4_strong_scaling():
    """BASELINE.json configs[3] as it is stated — ONE batch sharded over the GPUs (512 per GPU at 8; here 512 split over two), the shared
    RHS-MLP gradient all-reduced once per step. Each rank's coupled solve adapts on its own columns (SURVEY.md §8e option (i))."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", "c4",
                        "--scaling", "strong", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 512 and d["config"]["batch_per_gpu"] == 256 and d["value"] > 0


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs on the node")
@pytest.mark.parametrize("dtype", ["f32", "mixed"])
def test_bench_two_gpus_whole_training_step(dtype):
    """goku_step at two ranks: the split-graph step (two hipGraph replays around the eager flat gradient all-reduce) with its RCCL
    communicator — the path BASELINE.json configs[4] takes at 8 GPUs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--workload", "goku_step",
                        "--dtype", dtype], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["loss"] == d["loss"]
    assert "two hipGraph replays" in d["config"]["submission"] or "eager" in d["config"]["submission"]
