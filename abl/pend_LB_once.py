"""one large-batch forward + adjoint (for rocprofv3 --pmc passes)"""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
T, B = 50, 1 << 20
ts = O.time_grid(T); tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
z0, L = O.pendulum_inputs(B)
nat = Native(make_desc())
z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
dz = torch.randn((T, B, 2), device="cuda"); g0 = torch.empty((B, 2), device="cuda"); gt = torch.empty((B, 1), device="cuda")
for _ in range(3):
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp) == 0
    assert lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dz), p(g0), p(gt), None, sp) == 0
torch.cuda.synchronize()
