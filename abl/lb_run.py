"""B = 2^20 (or argv[1]) forward + pullback of the GOKU pendulum path, N launches each — the program abl/lb_traffic.sh runs under rocprofv3."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
sense = LL.SENSE_DISCRETE if (len(sys.argv) < 3 or sys.argv[2] == "discrete") else LL.SENSE_PARALLEL_CHECKPOINTED
N = 10
T = 50
ts = O.time_grid(T); tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
z0, L = O.pendulum_inputs(B); dz = O.cotangent(T, B, 2)
nat = Native(make_desc(sensealg=sense))
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda"); dzd = torch.tensor(dz, device="cuda")
zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
g0 = torch.empty((B, 2), device="cuda"); gt = torch.empty((B, 1), device="cuda")
for _ in range(N):
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp) == 0
    assert lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dzd), p(g0), p(gt), C.c_void_p(), sp) == 0
torch.cuda.synchronize()
print("ran", B, N, lib.lde_last_kernel(nat.h, 0).decode(), lib.lde_last_kernel(nat.h, 1).decode(), nat.stats(0)["naccept"] / B)
