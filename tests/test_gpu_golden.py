"""GPU: HIP path (through the C ABI) against the committed golden fixtures of tests/golden/ — the data that is
guaranteed identical on the build container and on the GPU box.

Tolerances: tight-tolerance / fixed-step fixtures ≤ 2e-5 on ẑ (fp32 round-off through ≤ 200 RHS evaluations);
default-tolerance fixtures: worst column within 3× max(3e-4, the fixture's own float64 error), median within
max(1e-4, half of it), and no farther from the stored float64 truth than 2.5× the fixture's own error + 1e-5. Gradients: ≤ 1e-3 relative (default tol 1e-2)
against the fixture, ≤ the fixture's own distance ×3.5 + 1e-3 against the float64 adjoint."""
import glob
import os

import numpy as np
import pytest

from tests.golden import make_golden as G

pytestmark = pytest.mark.gpu
FIX = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
             if not os.path.basename(f).startswith("chain_"))   # chain fixtures: tests/test_oracle_chain.py


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
        # two correct fp32 solves differ by at most about the solver's own error at this tolerance
        # (relu right-hand sides are the worst case: a kink crossed at a slightly different step size moves a
        #  trajectory by a multiple of the local error — hence the factor on the worst column, with the bulk bounded)
        assert per.max() <= 3 * max(3e-4, e_o) and np.median(per) <= max(1e-4, 0.5 * e_o)
        assert e_k <= 2.5 * e_o + 1e-5
    assert abs(st["naccept"] - fx["fwd_stats"][1]) <= 0.10 * fx["fwd_stats"][1] + 2
    g0, gth, gW, sb = nat.adjoint(z, theta, ts, dz)
    lim = 1e-3 if tight else 1e-2   # default tolerance (esp. relu right-hand sides): ~1 % between two correct fp32 solves
    s0 = np.abs(fx["dz0"]).max()

    def close(g, ref, ref64):   # within lim of the fixture, or within 3× the fixture's own float64 error (≈1 % for relu at reltol=1e-3)
        return np.abs(g - ref).max() <= max(lim * np.abs(ref).max(), 3 * np.abs(ref - ref64).max())
    assert close(g0[:k], fx["dz0"], fx["dz0_64"])
    assert np.abs(g0[:k] - fx["dz0_64"]).max() <= 3.5 * np.abs(fx["dz0"] - fx["dz0_64"]).max() + 1e-3 * s0
    if theta is not None:
        assert close(gth[:k], fx["dtheta"], fx["dtheta_64"])
    if W is not None:
        sw = np.abs(fx["dW"]).max()
        assert close(gW[fx["dW_idx"]], fx["dW"], fx["dW_64"])
        assert abs(np.linalg.norm(gW.astype(np.float64)) - fx["dW_norm"][0]) <= 4 * lim * fx["dW_norm"][0]
        assert np.abs(gW[fx["dW_idx"]] - fx["dW_64"]).max() <= 3.5 * np.abs(fx["dW"] - fx["dW_64"]).max() + 1e-3 * sw
