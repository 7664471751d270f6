"""k_pend_forward_lp beyond one workgroup per CU: forward µs per launch by batch (option pend_sh_max_b = B), discrete sensealg."""
import ctypes as C, os, sys
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
def timeit(f, n=200):
    for _ in range(20): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(s)
    for _ in range(n): f()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for B in (256, 512, 768, 1024, 1536, 2048):
    z0, L = O.pendulum_inputs(B)
    z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
    nat = Native(make_desc(sensealg=LL.SENSE_DISCRETE)); nat.set_option("pend_sh_max_b", B)
    zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
    f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
    assert f() == 0
    out.append("B=%d %.1f" % (B, timeit(f)))
print(os.path.basename(os.environ.get("LDE_LIB_PATH", "product")), "  ".join(out))
