"""c3 / c4 at tight tolerances, full batch: distances kernel ↔ f32 oracle ↔ float64 (which of the two f32 solves is off, and by how much),
for the relu networks of the configs and for the same shapes with tanh (smooth: the solver's order holds and the f32 solves converge)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle

NT = 32
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()


def run(name, kw, B, D, pend, tol):
    layers = kw["layers"]
    W = O.mlp_weights(layers, seed=3)
    kw = {**kw, "abstol": tol, "reltol": tol}
    d = make_desc(**kw)
    nat = Native(d); nat.set_weights(W); od = copy_desc_to_oracle(d)
    T = 50
    if pend:
        z0, L = O.pendulum_inputs(B)
    else:
        z0, L = (0.5 * np.random.default_rng(1).standard_normal((B, D))).astype(np.float32), None
    ts = O.time_grid(T); dz = O.cotangent(T, B, D)
    z, ret, st = nat.forward(z0, L, ts)
    g0, gL, gW, sb = nat.adjoint(z, L, ts, dz)
    o32, o64 = O.Oracle("f32"), O.Oracle("f64")
    zr, _, info = o32.forward(od, z0, L, ts, W=W, nthreads=NT)
    r0, rL, rW, infob = o32.adjoint(od, z, L, ts, dz, W=W, nthreads=NT)
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    z64, _, _ = o64.forward(d64, z0, L, ts, W=W.astype(np.float64), nthreads=NT)
    t0, tL, tW, _ = o64.adjoint(d64, z64, L, ts, dz, W=W.astype(np.float64), nthreads=NT)
    print(f"{name} tol {tol}: steps kernel {st['naccept']}+{st['nreject']} / {sb['naccept']}+{sb['nreject']}  oracle {info['naccept']} / {infob['naccept']}")
    print("  |z-zr| %.2e  |z-z64| %.2e  |zr-z64| %.2e  scale %.2f" % (np.abs(z - zr).max(), np.abs(z - z64).max(), np.abs(zr - z64).max(), np.abs(zr).max()))
    print("  grads vs f32 oracle: dz0 %.2e dW %.2e ; kernel vs f64: dz0 %.2e dW %.2e ; oracle vs f64: dz0 %.2e dW %.2e" % (
        rel(g0, r0), rel(gW, rW), rel(g0, t0), rel(gW, tW), rel(r0, t0), rel(rW, tW)))


c3 = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 64, 64, 2))
c4 = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=(32, 128, 128, 32), batching=O.BATCH_COUPLED)
which = sys.argv[1:] or ["c3relu", "c3tanh", "c4relu", "c4tanh"]
if "c3relu" in which: run("c3 relu", c3, 1024, 2, True, 1e-6)
if "c3tanh" in which: run("c3 tanh", {**c3, "activation": O.ACT_TANH}, 1024, 2, True, 1e-6)
if "c4relu" in which: run("c4 relu", c4, 512, 32, False, 1e-6)
if "c4tanh" in which: run("c4 tanh", {**c4, "activation": O.ACT_TANH}, 512, 32, False, 1e-6)
