"""k_pend_forward_lp against k_pend_forward_sh on the SAME recorded steps: ẑ of each kernel against the oracle's f32 and f64 replays of that
kernel's own step record (tight tolerances, long records — where round-off accumulates over hundreds of steps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle

o32, o64 = O.Oracle("f32"), O.Oracle("f64")
NT = 16
cases = [("tight 1e-8, T=50, B=200", dict(abstol=1e-8, reltol=1e-8), 200, O.time_grid(50), 9),
         ("ragged 1e-8, T=150, B=70", dict(abstol=1e-8, reltol=1e-8), 70, None, 11),
         ("default, T=50, B=256", dict(), 256, O.time_grid(50), 1),
         ("1e-6, T=50, B=256", dict(abstol=1e-6, reltol=1e-6), 256, O.time_grid(50), 1)]
for name, kw, B, ts, seed in cases:
    if ts is None:
        rng = np.random.default_rng(11)
        ts = np.concatenate([[0.0], np.cumsum(rng.uniform(0.002, 0.05, 149) * rng.choice([1.0, 1.0, 4.0], 149))])
    z0, L = O.pendulum_inputs(B, seed=seed)
    for lp in (1, 0):
        d = make_desc(sensealg=O.SENSE_DISCRETE, **kw)
        nat = Native(d)
        od = copy_desc_to_oracle(d)
        nat.set_option("record_capacity", 2048)
        nat.set_option("pend_lp", lp)
        z, ret, st = nat.forward(z0, L, ts)
        rec = nat.step_record(0, B, cap=2048)
        z32, _, _, _ = o32.forward_steps(od, z0, L, ts, rec=rec, nthreads=NT)
        z64, _, _, _ = o64.forward_steps(od, z0, L, ts, rec=rec, nthreads=NT)
        e = np.abs(z - z64)
        print(f"{name:28s} {'lp' if lp else 'sh'}: steps {st['naccept'] / B:6.1f}/traj  |z-z32| {np.abs(z - z32).max():.2e}  |z-z64| {e.max():.2e}  "
              f"|z32-z64| {np.abs(z32 - z64).max():.2e}   at t_end: x {e[-1, :, 0].max():.2e} v {e[-1, :, 1].max():.2e}   interior: x {e[:-1, :, 0].max():.2e} v {e[:-1, :, 1].max():.2e}")
