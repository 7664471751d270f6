"""Step counts / timing of the small-batch MLP kernels against the tile kernels (LDE_MLPV=0) on the coupled c4 shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
layers = (32, 128, 128, 32)
W = O.mlp_weights(layers, seed=3)
kw = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=O.BATCH_COUPLED)
o32 = O.Oracle("f32")
for B in (16, 72, 200):
    z0 = (0.5 * np.random.default_rng(1).standard_normal((B, 32))).astype(np.float32)
    ts = O.time_grid(50); dz = O.cotangent(50, B, 32)
    d = make_desc(**kw); od = copy_desc_to_oracle(d)
    for flag in ("1", "0"):
        os.environ["LDE_MLPV"] = flag
        nat = Native(d); nat.set_weights(W)
        z, ret, st = nat.forward(z0, None, ts)
        g0, _, gW, sb = nat.adjoint(z, None, ts, dz)
        print("B", B, "mlpv", flag, "fwd acc/rej", st["naccept"], st["nreject"], "adj acc/rej", sb["naccept"], sb["nreject"], "nfe", sb["nfe"], "|g0|", float(np.abs(g0).max()))
    zr, _, info = o32.forward(od, z0, None, ts, W=W, nthreads=16)
    r0, _, rW, infob = o32.adjoint(od, zr, None, ts, dz, W=W, nthreads=16)
    print("   oracle fwd", info["naccept"], info["nreject"], "adj", infob["naccept"], infob["nreject"], "|r0|", float(np.abs(r0).max()))
