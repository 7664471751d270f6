"""Forward GOKU kernel time vs batch size around the small-batch kernel's limit (LDE_PEND_WS_MAX_B)."""
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
for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    z0, L = O.pendulum_inputs(B)
    nat = Native(make_desc())
    z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
    zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
    f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
    for _ in range(10): assert f() == 0
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in evs:
        a.record(s); f(); b.record(s)
    torch.cuda.synchronize()
    print("B=%6d  forward %.2f us" % (B, np.median([a.elapsed_time(b) for a, b in evs]) * 1e3))
