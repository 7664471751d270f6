"""Large-batch GOKU forward: k_pend_forward (LDE_PEND_LB=0) against the one-trajectory-per-lane kernel with the LDS row ring
(LDE_PEND_LB = ring rows, LDE_PEND_LB_HOLD = hold margin; LDE_PEND_LB_MIN_B=0 to force it at every size)."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
T = 50
ts = O.time_grid(T); tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for B in (32768, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20):
    z0, L = O.pendulum_inputs(B)
    nat = Native(make_desc())
    z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
    zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
    f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
    for _ in range(5): assert f() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20): f()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("LB=%s B=%8d  forward %.1f us  %.2f TB/s algorithmic" % (os.environ.get("LDE_PEND_LB", "16"), B, ms * 1e3, 412 * B / (ms * 1e-3) / 1e12))
