"""GPU: bench.py keeps the driver's contract — one JSON line with the agreed keys, for the metric workload and the others."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # ONE JSON line on stdout
    return json.loads(lines[0])


def test_metric_line():
    d = _run("--steps", "20", "--warmup", "3")
    assert KEYS <= set(d), KEYS - set(d)
    assert d["metric"].startswith("trajectories/sec (fwd+adjoint) GOKU pendulum, batch=256") and d["unit"] == "trajectories/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("goku_pendulum") and d["config"]["global_batch"] == 256
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert d["value"] > c["value"]
    assert d["solver_stats"]["forward"]["nfailed"] == 0 and d["solver_stats"]["adjoint"]["nfailed"] == 0
    # the headline runs the reference's own default gradient: ForwardDiffSensitivity = LDE_SENSE_DISCRETE [REF pendulum.jl:8-11]
    assert d["sensealg"].startswith("LDE_SENSE_DISCRETE") and "reference's default" in d["sensealg"]
    assert d["other_sensealg"]["sensealg"] == "continuous" and d["other_sensealg"]["value"] > 0
    # counter bytes are attached from a committed summary ONLY when it holds the kernel this run launched (lde_last_kernel): a stale summary
    # must not survive a kernel change
    lk = r["launched_kernels"]
    assert lk["lde_forward"].startswith("k_pend_forward") and lk["lde_adjoint"].startswith("k_pend_adjoint")
    # … and the committed tree must hold that summary: a kernel renamed or re-dispatched without re-collecting its profile FAILS here
    # (bench.py then writes traffic = null and a traffic_source that starts with "stale")
    assert r["traffic"] is not None and not r["traffic_source"].startswith("stale"), r.get("traffic_source")
    dom = "lde_forward" if r["kernel"] == "lde_forward" else "lde_adjoint"
    assert r["traffic_kernel"].startswith(lk[dom]) and r["traffic_source"].startswith("profiles/"), (r["traffic_kernel"], lk)
    # the CPU side runs the same definition of the gradient, by the algorithm the reference executes (the solve on dual numbers)
    assert "dual numbers" in c["algorithm"] and c["reverse_sweep"]["value"] > 0 and c["continuous_adjoint"]["value"] > 0
    assert abs(d["vs_cpu_baseline"]["ratio"] - d["value"] / c["value"]) <= 1e-9 * d["vs_cpu_baseline"]["ratio"]


@pytest.mark.parametrize("workload", ["goku_pendulum", "c3"])
def test_two_ranks_on_one_gpu_run_the_n_gt_1_path(workload):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), on a box with ONE GPU: LDE_BENCH_SHARE_GPU=1
    puts both ranks on device 0 and lets them meet over gloo (RCCL refuses two ranks on one device). Everything of the N > 1 path but RCCL
    itself runs — sharding, the barriers, MAX over ranks, the strong-scaling rider, rank 0's single line — which tests/test_gpu_multi.py can
    only run on a multi-GPU node."""
    env = dict(os.environ, LDE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
                        "--workload", workload], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "shared_gpu_test" in d["config"]
    B = 256 if workload == "goku_pendulum" else 1024
    assert d["config"]["batch_per_gpu"] == B and d["config"]["global_batch"] == 2 * B
    assert abs(d["value"] - 2 * B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]     # whole-job aggregate over both ranks
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0
    if workload == "goku_pendulum":
        assert d["strong_scaling"]["global_batch"] == 256 and d["strong_scaling"]["batch_per_gpu"] == 128 and d["strong_scaling"]["value"] > 0
        assert d["sensealg"].startswith("LDE_SENSE_DISCRETE")
    else:
        assert "all_reduce" in d["config"]["parallelism"]


@pytest.mark.parametrize("workload,bound", [("c2", "mfma"), ("goku_decoder", "mfma"), ("goku_step", "mfma")])
def test_other_workloads(workload, bound):
    d = _run("--workload", workload, "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["roofline"]["bound"] == bound and d["value"] > 0 and d["cpu_baseline"] is None
    assert workload in d["config"]["workload"] and "model" not in d["config"]
