import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
o64 = O.Oracle("f64")
rng = np.random.default_rng(11)
ts = np.concatenate([[0.0], np.cumsum(rng.uniform(0.002, 0.05, 149) * rng.choice([1.0, 1.0, 4.0], 149))])
B = 70
z0, L = O.pendulum_inputs(B, seed=11)
for lp in (1, 0):
    d = make_desc(sensealg=O.SENSE_DISCRETE, abstol=1e-8, reltol=1e-8)
    nat = Native(d); od = copy_desc_to_oracle(d)
    nat.set_option("record_capacity", 2048); nat.set_option("pend_lp", lp)
    z, ret, st = nat.forward(z0, L, ts)
    rec = nat.step_record(0, B, cap=2048)
    z64, _, _, _ = o64.forward_steps(od, z0, L, ts, rec=rec, nthreads=16)
    e = z - z64
    b = int(np.abs(e).max(axis=(0, 2)).argmax())
    n = int(rec["n"][b]); tt = rec["t"][b, :n]; dd = rec["dt"][b, :n]
    # is a save time a step end?
    ends = tt + dd
    is_end = np.array([np.any(np.abs(ends - s) < 1e-12) for s in ts])
    print("lp" if lp else "sh", "worst trajectory", b, "steps", n, "L", L[b], "z0", z0[b])
    for j in list(range(1, 12)) + list(range(128, 150)):
        k = int(np.searchsorted(tt, ts[j], side="left")) - 1      # the step with tt[k] < ts[j] <= tt[k] + dd[k]
        th = (ts[j] - tt[k]) / dd[k]
        print(f"  j={j:3d} t={ts[j]:8.5f} step {k:3d} h={dd[k]:.5f} theta={th:6.4f} ex={e[j, b, 0]: .2e} ev={e[j, b, 1]: .2e}   x={z64[j, b, 0]: .4f} v={z64[j, b, 1]: .4f}")
