"""Parity margins of the GOKU path at the tight tolerance (gate: 1e-5 from the oracle and from float64 truth) and the default one."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
from oracle import oracle as O
o32, o64 = O.Oracle("f32"), O.Oracle("f64")
for kind in (O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION):
    for tol in ((1e-6, 1e-6), (1e-6, 1e-3)):
        for B in (256, 1000):
            d = make_desc(rhs_kind=kind, abstol=tol[0], reltol=tol[1]); od = copy_desc_to_oracle(d)
            z0, L = O.pendulum_inputs(B); ts = O.time_grid(50)
            z, ret, _ = Native(d).forward(z0, L, ts)
            zr, _, _ = o32.forward(od, z0, L, ts)
            zt, _, _ = o64.forward(O.make_desc(rhs_kind=kind, abstol=1e-11, reltol=1e-11), z0, L, ts)
            dz = O.cotangent(50, B, 2)
            g0, gL, _, _ = Native(d).adjoint(z, L, ts, dz)
            r0, rL, _, _ = o32.adjoint(od, z, L, ts, dz)
            print("kind %d tol %s B %4d: |z-oracle| %.2e |z-truth| %.2e |oracle-truth| %.2e  adj rel dz0 %.2e dL %.2e" % (
                kind, tol, B, np.abs(z - zr).max(), np.abs(z - zt).max(), np.abs(zr - zt).max(),
                np.abs(g0 - r0).max() / np.abs(r0).max(), np.abs(gL - rL).max() / np.abs(rL).max()))
