"""latentode_ref's adjoint with two builds of the library (LDE_LIB_PATH): step counts and bitwise comparison of the results.
    LDE_LIB_PATH=…/liblde_old.so python abl/spec_ab.py gpurun_out/ab_old.npz;  python abl/spec_ab.py gpurun_out/ab_new.npz;  python abl/spec_ab.py --cmp a b"""
import sys
import numpy as np
sys.path.insert(0, ".")
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        print(k, "equal" if np.array_equal(a[k], b[k]) else ("max abs diff %.3e (scale %.3e)" % (np.abs(a[k].astype(np.float64) - b[k]).max(), np.abs(a[k]).max())))
    sys.exit(0)
from oracle import oracle as O
from tests.gpu_util import Native, make_desc
layers = (16, 200, 200, 16)
W = O.mlp_weights(layers, seed=3)
nat = Native(make_desc(rhs_kind=O.RHS_MLP, state_dim=16, param_dim=0, layers=layers, batching=O.BATCH_COUPLED))
nat.set_weights(W)
B, T = 64, 50
z0 = (0.5 * np.random.default_rng(1).standard_normal((B, 16))).astype(np.float32)
ts = O.time_grid(T)
dz = O.cotangent(T, B, 16)
z, ret, st = nat.forward(z0, None, ts)
g0, _, gW, sb = nat.adjoint(z, None, ts, dz)
print("forward", st, "adjoint", sb)
np.savez(sys.argv[1], z=z, g0=g0, gW=gW, stats=np.array([sb["nfe"], sb["naccept"], sb["nreject"]]))
