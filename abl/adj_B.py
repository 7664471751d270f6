"""GOKU adjoint time vs batch size (the fused time-parallel kernel ≤ 32768, the streaming form above)."""
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
for B in [int(x) for x in os.environ.get("BS", "256,4096,32768,65536,262144,1048576").split(",")]:
    z0, L = O.pendulum_inputs(B)
    nat = Native(make_desc())
    z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
    zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
    dz = torch.tensor(O.cotangent(T, B, 2), device="cuda"); g0 = torch.empty((B, 2), device="cuda"); gt = torch.empty((B, 1), device="cuda")
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp) == 0
    f = lambda: lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dz), p(g0), p(gt), C.c_void_p(), sp)
    for _ in range(5): assert f() == 0
    n = 30
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(n): f()
    b.record(s); torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print("B=%8d  adjoint %9.2f us   %7.1f GB/s algorithmic (412 B per trajectory)" % (B, us, 412.0 * B / us * 1e-3))
