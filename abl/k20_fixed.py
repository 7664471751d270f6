"""Where the fixed ≈ 25 µs of a K = 20 timed region go: host clock around hipGraphLaunch and the synchronise, device events around the replay."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
B, T = 256, 50
z0, L = O.pendulum_inputs(B); ts = O.time_grid(T); dz = O.cotangent(T, B, 2)
tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
nat = Native(make_desc(sensealg=LL.SENSE_DISCRETE))
dev = "cuda"
z0d = torch.tensor(z0, device=dev); thd = torch.tensor(L, device=dev); dzd = torch.tensor(dz, device=dev)
zout = torch.empty((T, B, 2), device=dev); ret = torch.empty((B,), device=dev, dtype=torch.int32)
g0 = torch.empty((B, 2), device=dev); gt = torch.empty((B, 1), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
gs = torch.cuda.Stream(); gsp = C.c_void_p(gs.cuda_stream)
def step(sp):
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp) == 0
    assert lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dzd), p(g0), p(gt), C.c_void_p(), sp) == 0
for K in (1, 5, 20, 100):
    with torch.cuda.stream(gs):
        step(gsp); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs, capture_error_mode="thread_local"):
            for _ in range(K): step(gsp)
        torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    rows = []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    r = np.median(np.array(rows), axis=0)
    # device-side span: events on the current stream around the replay
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sp_ = []
    for rep in range(30):
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); sp_.append(e0.elapsed_time(e1) * 1e3)
    print(f"K={K:4d}  hipGraphLaunch returns after {r[0]:6.1f} us; launch+sync {r[1]:7.1f} us = {r[1]/K:6.2f} us/step; device span between events {np.median(sp_):7.1f} us = {np.median(sp_)/K:6.2f} us/step")
