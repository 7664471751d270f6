"""GPU: the cross-PROCESS half of lde_set_global_sum_peers on a one-GPU box (hipIpcOpenMemHandle takes its
handle by value).

RCCL cannot put two ranks on one device; HIP IPC can: two processes on GPU 0 allocate their mailboxes (fine-grained device memory), exchange
the IPC handles over a gloo group (`dist.GlobalSumMailboxes`), map each other's mailbox and run the two shards of ONE coupled c4-shaped solve
(32-128-128-32, Tsit5, BASELINE.json configs[3]'s network) under LDE_BATCH_COUPLED_GLOBAL — the step control's sums written by each
process's kernel into the other's mailbox while both kernels run [REF src/models/LatentODE.jl:70-72: the reference's norm runs over the
whole [D'×B] state]. Checked against the unsharded solve in this process: same accepted steps, ẑ ≤ 1e-6·scale, gradients ≤ 1e-5, for the
continuous adjoint and for LDE_SENSE_DISCRETE (whose pullback exchanges nothing); and a rank that leaves before the solve poisons the
other (NaN blocks, retcode ≠ 0) after the bounded spin instead of hanging it. What a one-GPU box cannot show is the mapping across DEVICES
(peer access over xGMI): tests/test_gpu_multi.py keeps those cases for a multi-GPU box.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS = [(0, 80), (80, 192)]
B = 192


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _launch(tmp, mode, sense):
    port = _port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py"), str(r), "2", port, str(tmp), mode, sense,
                               str(lo), str(hi), str(B)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r, (lo, hi) in enumerate(BOUNDS)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o))
    return outs


@pytest.mark.parametrize("sense", ["continuous", "discrete"])
def test_two_processes_on_one_gpu_share_one_coupled_solve(tmp_path, sense):
    from tests.test_gpu_coupled_global import KW, LAYERS, _inputs, _setup
    W = O.mlp_weights(LAYERS, seed=3)
    z0, ts, dz = _inputs(B)
    z0[80:] *= 3.0
    more = dict(sensealg=O.SENSE_DISCRETE) if sense == "discrete" else {}
    ref = _setup(O.BATCH_COUPLED, W, **more)
    zf, _, sf = ref.forward(z0, None, ts)
    f0, _, fW, sbf = ref.adjoint(zf, None, ts, dz)
    ref.close()
    outs = _launch(tmp_path, "solve", sense)
    assert all(rc == 0 for rc, _ in outs), outs
    scale = max(1.0, np.abs(zf).max())
    gW = 0
    for r, (lo, hi) in enumerate(BOUNDS):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert (d["ret"] == 0).all() and int(d["nfailed"]) == 0 and int(d["adj_failed"]) == 0
        assert int(d["naccept"]) == sf["naccept"] and int(d["badj"]) == sbf["naccept"]      # the SAME step sequence on both ranks
        assert np.abs(d["z"] - zf[:, lo:hi]).max() <= 1e-6 * scale
        assert np.abs(d["g0"] - f0[lo:hi]).max() <= 1e-5 * np.abs(f0).max()
        assert (d["ret2"] == 0).all() and np.array_equal(d["z2"], d["z"])                  # second launch: the other word sets
        gW = gW + d["gW"]
    assert np.abs(gW - fW).max() <= 1e-5 * np.abs(fW).max()       # Σ over ranks: the path's one data collective


def test_a_rank_that_leaves_poisons_the_other_without_hanging(tmp_path):
    outs = _launch(tmp_path, "leave", "continuous")
    assert all(rc == 0 for rc, _ in outs), outs
    d = np.load(tmp_path / "rank0.npz")
    assert (d["ret"] != 0).all() and int(d["nfailed"]) == 80 and np.isnan(d["z"][1:]).all()    # NaN blocks, as every failed solve
    assert not (tmp_path / "rank1.npz").exists()
