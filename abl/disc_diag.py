"""Diagnostic (GPU box): per-trajectory error statistics of the discrete adjoint against the oracle on the kernel's steps — relu vs tanh at
the BASELINE sizes (are the outliers relu-kink flips?)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from tests.test_gpu_discrete import MLP_CASES, _mlp_inputs, _native
o32, o64 = O.Oracle("f32"), O.Oracle("f64")
NT = 32
for name, B in (("c3_pend_plus_mlp_relu", 1024), ("latentode_ref_relu_coupled", 64), ("c4_relu_coupled", 512)):
    for act in (O.ACT_RELU, O.ACT_TANH):
        kw, _ = MLP_CASES[name]
        kw = {**kw, "activation": act}
        W, z0, theta, ts, dz = _mlp_inputs(kw, B, seed=3)
        nat, od = _native(W, **kw)
        z, ret, st = nat.forward(z0, theta, ts)
        rec = nat.step_record(0, B)
        zr, _, _, _ = o32.forward_steps(od, z0, theta, ts, W=W, rec=rec, nthreads=NT)
        g0, gth, gW, sb = nat.adjoint(z, theta, ts, dz)
        r0, rth, rW, _ = o32.adjoint_discrete(od, z, theta, ts, dz, rec, W=W, nthreads=NT)
        per = np.abs(g0 - r0).max(axis=1) / np.abs(r0).max()
        perz = np.abs(z - zr).max(axis=(0, 2))
        print(name, "relu" if act == O.ACT_RELU else "tanh", "B", B, "steps", st["naccept"], "nfe_adj", sb["nfe"],
              "| dz max", perz.max(), "n>2e-5:", int((perz > 2e-5).sum()),
              "| dz0 rel max", per.max(), "n>1e-4:", int((per > 1e-4).sum()), "median", np.median(per),
              "| dW rel", np.abs(gW - rW).max() / np.abs(rW).max(), flush=True)
