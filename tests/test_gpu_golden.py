"""GPU: HIP path (through the C ABI) against the committed golden fixtures of tests/golden/ — the data that is
guaranteed identical on the build container and on the GPU box.

Tolerances: tight-tolerance / fixed-step fixtures ≤ 2e-5 on ẑ (fp32 round-off through ≤ 200 RHS evaluations);
default-tolerance fixtures follow tests/test_gpu_pendulum.py (99 % ≤ 1e-4, max ≤ 3e-4, and no farther from the
stored float64 truth than 1.5× the fixture's own error + 1e-5). Gradients: ≤ 1e-3 relative (default tol 5e-3)
against the fixture, ≤ the fixture's own distance ×1.5 + 1e-3 against the float64 adjoint."""
import glob
import os

import numpy as np
import pytest

from tests.golden import make_golden as G

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(f)[:-4] for f in FIX])
def test_hip_matches_golden_fixture(path):
    import ctypes as C
    from tests.gpu_util import Native
    from latentdiffeq_amd import _lib as L
    name = os.path.basename(path)[:-4]
    cfg = G.CONFIGS[name]
    fx = np.load(path)
    ts, z0, theta, W, dz = G.inputs(cfg)
    k = cfg["keep"]
    od = G.desc(cfg)
    d = L.ProblemDesc()
    C.memmove(C.byref(d), C.byref(od), C.sizeof(d))
    nat = Native(d)
    if W is not None:
        nat.set_weights(W)
    z, ret, st = nat.forward(z0, theta, ts)
    assert np.array_equal(ret[:k], fx["retcode"]) and (ret == 0).all()
    tight = cfg.get("reltol", 1e-3) < 1e-4 or not cfg.get("adaptive", True)
    per = np.abs(z[:, :k] - fx["z"]).max(axis=(0, 2))
    e_k, e_o = np.abs(z[:, :k] - fx["z64"]).max(), np.abs(fx["z"] - fx["z64"]).max()
    if tight:
        assert per.max() <= 2e-5
    else:
        assert per.max() <= 3e-4 and np.quantile(per, 0.99) <= 1e-4 or per.max() <= 1e-4
        assert e_k <= 1.5 * e_o + 1e-5
    assert abs(st["naccept"] - fx["fwd_stats"][1]) <= 0.03 * fx["fwd_stats"][1] + 1
    g0, gth, gW, sb = nat.adjoint(z, theta, ts, dz)
    lim = 1e-3 if tight else 5e-3
    s0 = np.abs(fx["dz0"]).max()
    assert np.abs(g0[:k] - fx["dz0"]).max() <= lim * s0
    assert np.abs(g0[:k] - fx["dz0_64"]).max() <= 1.5 * np.abs(fx["dz0"] - fx["dz0_64"]).max() + 1e-3 * s0
    if theta is not None:
        assert np.abs(gth[:k] - fx["dtheta"]).max() <= lim * np.abs(fx["dtheta"]).max()
    if W is not None:
        sw = np.abs(fx["dW"]).max()
        assert np.abs(gW[fx["dW_idx"]] - fx["dW"]).max() <= lim * sw
        assert abs(np.linalg.norm(gW.astype(np.float64)) - fx["dW_norm"][0]) <= lim * fx["dW_norm"][0]
        assert np.abs(gW[fx["dW_idx"]] - fx["dW_64"]).max() <= 1.5 * np.abs(fx["dW"] - fx["dW_64"]).max() + 1e-3 * sw
