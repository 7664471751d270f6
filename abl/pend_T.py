"""Forward pendulum kernel time vs number of save points (same adaptive steps): what do the saves cost?
HIP events around each single launch (the host is not faster than this kernel, so a loop of launches would time the host)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
B = int(os.environ.get("PB", 256))
z0, L = O.pendulum_inputs(B)
nat = Native(make_desc())
import ctypes as C
from latentdiffeq_amd import _lib as LL
lib = LL.load()
z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for T in (2, 3, 6, 11, 26, 50, 99):
    ts = np.linspace(0.0, 2.45, T)
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
    f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
    for _ in range(10): assert f() == 0
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a, b in evs:
        a.record(s); f(); b.record(s)
    torch.cuda.synchronize()
    print("T=%3d  forward kernel %.2f us" % (T, np.median([a.elapsed_time(b) for a, b in evs]) * 1e3))
