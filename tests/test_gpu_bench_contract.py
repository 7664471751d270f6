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
    if r["traffic"] is not None:
        dom = "lde_forward" if r["kernel"] == "lde_forward" else "lde_adjoint"
        assert r["traffic_kernel"].startswith(lk[dom]) and r["traffic_source"].startswith("profiles/"), (r["traffic_kernel"], lk)
    else:
        assert "stale" in r.get("traffic_source", "stale") or "traffic_source" not in r
    # the CPU side runs the same definition of the gradient, by the algorithm the reference executes (the solve on dual numbers)
    assert "dual numbers" in c["algorithm"] and c["reverse_sweep"]["value"] > 0 and c["continuous_adjoint"]["value"] > 0
    assert abs(d["vs_cpu_baseline"]["ratio"] - d["value"] / c["value"]) <= 1e-9 * d["vs_cpu_baseline"]["ratio"]


@pytest.mark.parametrize("workload,bound", [("c2", "mfma"), ("goku_decoder", "mfma"), ("goku_step", "mfma")])
def test_other_workloads(workload, bound):
    d = _run("--workload", workload, "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["roofline"]["bound"] == bound and d["value"] > 0 and d["cpu_baseline"] is None
    assert workload in d["config"]["workload"] and "model" not in d["config"]
