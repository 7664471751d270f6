import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.gpu_util import Native, make_desc
rng = np.random.default_rng(11)
ts = np.concatenate([[0.0], np.cumsum(rng.uniform(0.002, 0.05, 149) * rng.choice([1.0, 1.0, 4.0], 149))])
B = 70
z0, L = O.pendulum_inputs(B, seed=11)
out = dict(ts=ts, z0=z0, L=L)
for lp in (1, 0):
    d = make_desc(sensealg=O.SENSE_DISCRETE, abstol=1e-8, reltol=1e-8)
    nat = Native(d)
    nat.set_option("record_capacity", 2048); nat.set_option("pend_lp", lp)
    z, ret, st = nat.forward(z0, L, ts)
    rec = nat.step_record(0, B, cap=2048)
    out[f"z{lp}"] = z; out[f"t{lp}"] = rec["t"]; out[f"dt{lp}"] = rec["dt"]; out[f"n{lp}"] = rec["n"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "lp_dump.npz"), **out)
