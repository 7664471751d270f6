"""Large-batch recording forward (B = 2^20, the reference-default gradient's forward): µs per launch by the ring's hold margin (option "pend_lb_hold")."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
B, T = 1 << 20, 50
ts = O.time_grid(T); tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
z0, L = O.pendulum_inputs(B)
s = torch.cuda.current_stream(); sp = C.c_void_p(s.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
for sense, name in ((LL.SENSE_DISCRETE, "recording"), (LL.SENSE_PARALLEL_CHECKPOINTED, "plain")):
    row = []
    for hold in (-1, 0, 2, 4, 6, 8, 10, 12, 15):
        nat = Native(make_desc(sensealg=sense)); nat.set_option("pend_lb_hold", hold)
        f = lambda: lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp)
        for _ in range(3): assert f() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(s)
        for _ in range(10): f()
        b.record(s); torch.cuda.synchronize()
        row.append("%d: %.0f" % (hold, a.elapsed_time(b) / 10 * 1e3))
    print(name, "  ".join(row), flush=True)
