"""CPU: the oracle reproduces the committed golden fixtures (tests/golden/*.npz, made by make_golden.py).

Guards the checker itself against drift (compiler, libm, RNG) — the same fixtures are what the HIP kernels are
compared with on the GPU box (tests/test_gpu_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.golden import make_golden as G

FIX = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
             if not os.path.basename(f).startswith("chain_"))   # chain fixtures: tests/test_oracle_chain.py


def test_fixture_set_is_complete():
    assert {os.path.basename(f)[:-4] for f in FIX} == set(G.CONFIGS)
    assert all(os.path.getsize(f) < 256 * 1024 for f in FIX), "fixtures stay small"


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(f)[:-4] for f in FIX])
def test_oracle_reproduces_fixture(o32, path):
    name = os.path.basename(path)[:-4]
    cfg = G.CONFIGS[name]
    fx = np.load(path)
    ts, z0, theta, W, dz = G.inputs(cfg)
    k = cfg["keep"]
    cs = fx["checksums"]
    assert np.array_equal(ts, fx["ts"]) and np.array_equal(z0[:k], fx["z0"]) and np.array_equal(dz[:, :k], fx["dz"])
    assert abs(z0.astype(np.float64).sum() - cs[0]) < 1e-9 and abs(dz.astype(np.float64).sum() - cs[1]) < 1e-12
    if W is not None:
        assert abs(W.astype(np.float64).sum() - cs[4]) < 1e-9, "weight RNG stream drifted"
    d = G.desc(cfg)
    z, ret, info = o32.forward(d, z0, theta, ts, W=W)
    g0, gth, gW, binfo = o32.adjoint(d, z, theta, ts, dz, W=W)
    # same compiler flags (no FMA contraction, generic x86-64) ⇒ bit-reproducible; allow 1e-6 for libm drift
    assert np.abs(z[:, :k] - fx["z"]).max() <= 1e-6
    assert np.array_equal(ret[:k], fx["retcode"])
    assert list(fx["fwd_stats"]) == [info[n] for n in ("nfe", "naccept", "nreject", "nfailed", "max_steps")]
    assert list(fx["bwd_stats"])[1:3] == [binfo["naccept"], binfo["nreject"]]
    assert np.allclose(info["dt_trace"][:64], fx["dt_trace"], rtol=1e-5, atol=0)
    assert np.abs(g0[:k] - fx["dz0"]).max() <= 1e-6 * np.abs(fx["dz0"]).max() + 1e-12
    if theta is not None:
        assert np.abs(gth[:k] - fx["dtheta"]).max() <= 1e-6 * np.abs(fx["dtheta"]).max() + 1e-12
    if W is not None:
        assert np.abs(gW[fx["dW_idx"]] - fx["dW"]).max() <= 2e-6 * np.abs(fx["dW"]).max()
    # fp32 result vs the stored float64 truth: solver accuracy at the config's tolerance
    tight = cfg.get("reltol", 1e-3) < 1e-4 or not cfg.get("adaptive", True)
    assert np.abs(fx["z"] - fx["z64"]).max() <= (2e-5 if tight else 2e-3)
    assert np.abs(fx["dz0"] - fx["dz0_64"]).max() <= (1e-3 if tight else 2e-2) * np.abs(fx["dz0_64"]).max()
