"""Measured margins of the HIP path against the golden fixtures (what tests/test_gpu_golden.py gates on)."""
import ctypes as C, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden import make_golden as G
from tests.gpu_util import Native
from latentdiffeq_amd import _lib as L
for path in sorted(glob.glob("tests/golden/*.npz")):
    name = os.path.basename(path)[:-4]
    if name.startswith("chain_"):
        continue
    cfg = G.CONFIGS[name]; fx = np.load(path)
    ts, z0, theta, W, dz = G.inputs(cfg); k = cfg["keep"]
    d = L.ProblemDesc(); C.memmove(C.byref(d), C.byref(G.desc(cfg)), C.sizeof(d))
    nat = Native(d)
    if W is not None: nat.set_weights(W)
    z, ret, st = nat.forward(z0, theta, ts)
    g0, gth, gW, sb = nat.adjoint(z, theta, ts, dz)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    out = dict(z=float(np.abs(z[:, :k] - fx["z"]).max()), z_own64=float(np.abs(fx["z"] - fx["z64"]).max()),
               dz0=rel(g0[:k], fx["dz0"]), dz0_own64=rel(fx["dz0"], fx["dz0_64"]))
    if theta is not None: out.update(dth=rel(gth[:k], fx["dtheta"]), dth_own64=rel(fx["dtheta"], fx["dtheta_64"]))
    if W is not None: out.update(dW=rel(gW[fx["dW_idx"]], fx["dW"]), dW_own64=rel(fx["dW"], fx["dW_64"]))
    print(name, {k_: float("%.2e" % v) for k_, v in out.items()})
