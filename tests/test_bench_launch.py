"""CPU: `python bench.py --gpus N` starts the N ranks itself (child processes, before anything touches a GPU) and relays
ONE JSON line; a failing rank makes the launcher exit non-zero. `--dry-launch` runs the same launch path on gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)


@pytest.mark.timeout(900)
def test_gpus_2_spawns_two_ranks_and_emits_one_line():
    r = _bench("--gpus", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective_ranks"] == 2 and d["dry_launch"] is True


@pytest.mark.timeout(900)
def test_gpus_8_dry_launch_is_one_node_of_eight_ranks():
    """The shape of the driver's scaling run (N = 8 on one node): eight ranks rendezvous on 127.0.0.1, one collective, ONE line."""
    r = _bench("--gpus", "8", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["collective_ranks"] == 8


@pytest.mark.timeout(900)
def test_launcher_refuses_more_ranks_than_gpus_and_propagates_failure():
    # no GPU in this container: the non-dry launcher must refuse instead of printing an n_gpus: 1 line
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _bench("--gpus", "2", "--dry-launch", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    # a torchrun environment is taken as is (no self-launch): the dry path reports what the environment says
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1
