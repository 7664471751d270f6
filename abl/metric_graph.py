"""Metric config (GOKU pendulum, B = 256, T = 50): K (lde_forward + lde_adjoint) steps as stream launches against ONE hipGraph
replay of the same K steps.   python abl/metric_graph.py [K]"""
import sys, os, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_util import Native, make_desc
from oracle import oracle as O
from latentdiffeq_amd import _lib as LL
lib = LL.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T, B = 50, 256
ts = O.time_grid(T); tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
p = lambda t: C.c_void_p(t.data_ptr())
z0, L = O.pendulum_inputs(B)
nat = Native(make_desc())
z0d = torch.tensor(z0, device="cuda"); thd = torch.tensor(L, device="cuda")
zout = torch.empty((T, B, 2), device="cuda"); ret = torch.empty((B,), device="cuda", dtype=torch.int32)
dz = torch.randn((T, B, 2), device="cuda"); g0 = torch.empty((B, 2), device="cuda"); gt = torch.empty((B, 1), device="cuda")


def step(sp):
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp) == 0
    assert lib.lde_adjoint(nat.h, p(zout), p(thd), tsp, T, B, p(dz), p(g0), p(gt), None, sp) == 0


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = C.c_void_p(s.cuda_stream)
    for _ in range(20):
        step(sp)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            step(sp)
        torch.cuda.synchronize()
        print("stream launches: %.2f us/step" % ((time.perf_counter() - t0) / K * 1e6))
    ref0, refL = g0.clone(), gt.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K):
            step(sp)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        print("graph replay:    %.2f us/step" % ((time.perf_counter() - t0) / K * 1e6))
    assert torch.equal(ref0, g0) and torch.equal(refL, gt)
