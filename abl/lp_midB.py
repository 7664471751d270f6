"""Which forward mapping serves the batches between the metric's (256: a trajectory per workgroup, k_pend_forward_lp) and the large ones?
HIP-event time of lde_forward (recording, the default sensealg) and lde_adjoint per launch, by option "pend_sh_max_b" / "pend_tl_max_b"."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
T = 50
ts = O.time_grid(T)
tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def timeit(f, n=200):
    for _ in range(20): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(s)
    for _ in range(n): f()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
SENSE = LL.SENSE_DISCRETE if (len(sys.argv) < 2 or sys.argv[1] == "discrete") else LL.SENSE_PARALLEL_CHECKPOINTED
print("sensealg", SENSE)
for B in (256, 384, 512, 768, 1024, 1536, 2048, 4096):
    z0, L = O.pendulum_inputs(B)
    dz = O.cotangent(T, B, 2)
    z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda"); dzd = torch.tensor(dz, device="cuda")
    row = []
    for name, opts in (("default", {}), ("lp", {"pend_sh_max_b": B}), ("sh", {"pend_sh_max_b": B, "pend_lp": 0}), ("tl", {"pend_sh_max_b": 0, "pend_tl_max_b": B}), ("ws", {"pend_sh_max_b": 0, "pend_tl_max_b": 0})):
        nat = Native(make_desc(sensealg=SENSE))
        for k, v in opts.items(): nat.set_option(k, v)
        zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
        g0 = torch.empty((B, 2), device="cuda"); gt = torch.empty((B, 1), device="cuda")
        f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
        assert f() == 0
        g = lambda: lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dzd), p(g0), p(gt), C.c_void_p(), sp)
        assert g() == 0
        kn = lib.lde_last_kernel(nat.h, 0).decode()
        row.append("%s %s %.1f + %.1f us" % (name, kn.replace("k_pend_forward_", ""), timeit(f), timeit(g)))
    print("B=%5d  " % B + "   ".join(row), flush=True)
