"""GPU: behaviour of the C ABI around the kernels — workspace growth, the save-time cache, streams, weights updates,
determinism, argument errors. (Numerical parity lives in the other test_gpu_* files.)"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _native(W=None, **kw):
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    d = make_desc(**kw)
    nat = Native(d)
    if W is not None:
        nat.set_weights(W)
    return nat, copy_desc_to_oracle(d)


def test_workspace_grows_and_save_time_cache_tracks_changes(o32):
    """One handle, batches of different sizes and three different time grids in arbitrary order: every call must see
    ITS grid (the handle caches the device copy and re-uploads only on change) and a workspace that fits."""
    nat, od = _native(abstol=1e-6, reltol=1e-6)
    grids = [O.time_grid(50), O.time_grid(20, 0.1), O.time_grid(50) + 1.5, O.time_grid(50)]
    for B, ts in zip([64, 1000, 16, 300], grids):
        z0, L = O.pendulum_inputs(B, seed=B)
        z, ret, _ = nat.forward(z0, L, ts)
        zr, _, _ = o32.forward(od, z0, L, ts)
        assert (ret == 0).all() and np.abs(z - zr).max() <= 1e-5
        dz = O.cotangent(len(ts), B, 2, seed=B)
        g0, gL, _, _ = nat.adjoint(z, L, ts, dz)
        r0, rL, _, _ = o32.adjoint(od, z, L, ts, dz)
        assert np.abs(g0 - r0).max() <= 2e-4 * np.abs(r0).max()


def test_non_default_stream_and_two_handles_interleaved(o32):
    import torch
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    natA, od = _native(abstol=1e-6, reltol=1e-6)
    natB, _ = _native(abstol=1e-6, reltol=1e-6, rhs_kind=O.RHS_PENDULUM_FRICTION)
    B, T = 200, 50
    z0, Lp = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    dev = "cuda"
    z0d, thd = torch.from_numpy(z0).to(dev), torch.from_numpy(Lp).to(dev)
    outA = torch.empty((T, B, 2), device=dev)
    outB = torch.empty((T, B, 2), device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):   # interleaved enqueues on two streams, no host sync in between
        L.check(lib.lde_forward(natA.h, p(z0d), p(thd), tsp, T, B, p(outA), None, C.c_void_p(s1.cuda_stream)), natA.h, "A")
        L.check(lib.lde_forward(natB.h, p(z0d), p(thd), tsp, T, B, p(outB), None, C.c_void_p(s2.cuda_stream)), natB.h, "B")
    torch.cuda.synchronize()
    zA, _, _ = o32.forward(od, z0, Lp, ts)
    zB, _, _ = o32.forward(O.make_desc(abstol=1e-6, reltol=1e-6, rhs_kind=O.RHS_PENDULUM_FRICTION), z0, Lp, ts)
    assert np.abs(outA.cpu().numpy() - zA).max() <= 1e-5 and np.abs(outB.cpu().numpy() - zB).max() <= 1e-5


def test_weights_can_be_replaced_and_results_are_deterministic(o32):
    layers = (4, 24, 4)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=4, param_dim=0, layers=layers, activation=O.ACT_TANH, abstol=1e-6, reltol=1e-6)
    W1, W2 = O.mlp_weights(layers, seed=1), O.mlp_weights(layers, seed=2)
    nat, od = _native(W1, **kw)
    z0 = (0.5 * np.random.default_rng(0).standard_normal((33, 4))).astype(np.float32)
    ts = O.time_grid(12, 0.1)
    dz = O.cotangent(12, 33, 4)
    za, _, _ = nat.forward(z0, None, ts)
    ga = nat.adjoint(za, None, ts, dz)
    nat.set_weights(W2)
    zb, _, _ = nat.forward(z0, None, ts)
    nat.set_weights(W1)
    zc, _, _ = nat.forward(z0, None, ts)
    gc = nat.adjoint(zc, None, ts, dz)
    assert np.array_equal(za, zc) and not np.array_equal(za, zb), "bitwise reproducible; weights really replaced"
    assert np.array_equal(ga[0], gc[0]) and np.array_equal(ga[2], gc[2]), "dW reduction is order-deterministic"
    zr, _, _ = o32.forward(od, z0, None, ts, W=W1)
    assert np.abs(za - zr).max() <= 2e-5
    # dW is accumulated (+=) into the caller's buffer
    import torch
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    dev = "cuda"
    zo, dzo = torch.from_numpy(za).to(dev), torch.from_numpy(dz).to(dev)
    dz0 = torch.empty((33, 4), device=dev)
    dW = torch.full((nat.nW,), 1.0, device=dev)
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(lib.lde_adjoint(nat.h, p(zo), None, tsp, 12, 33, p(dzo), p(dz0), None, p(dW), None), nat.h, "adj")
    torch.cuda.synchronize()
    assert np.allclose(dW.cpu().numpy() - 1.0, ga[2], rtol=0, atol=2e-7 + 1e-6 * np.abs(ga[2]).max())


def test_argument_errors_are_codes_not_crashes():
    import torch
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    nat, _ = _native()
    z0, Lp = O.pendulum_inputs(8)
    ts = O.time_grid(5)
    dev = "cuda"
    z0d, thd = torch.from_numpy(z0).to(dev), torch.from_numpy(Lp).to(dev)
    out = torch.empty((5, 8, 2), device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.lde_forward(nat.h, None, p(thd), tsp, 5, 8, p(out), None, None) == -1           # NULL z0
    assert lib.lde_forward(nat.h, p(z0d), None, tsp, 5, 8, p(out), None, None) == -1           # pendulum needs θ̂
    assert lib.lde_forward(nat.h, p(z0d), p(thd), tsp, 0, 8, p(out), None, None) == -1         # T < 1
    bad = np.array([0.0, 0.1, 0.1, 0.2, 0.3])
    assert lib.lde_forward(nat.h, p(z0d), p(thd), bad.ctypes.data_as(C.POINTER(C.c_double)), 5, 8, p(out), None, None) == -1
    assert b"strictly increasing" in lib.lde_last_error(nat.h)
    # MLP handle without weights
    from tests.gpu_util import Native, make_desc
    m = Native(make_desc(rhs_kind=O.RHS_MLP, state_dim=2, param_dim=0, layers=(2, 8, 2)))
    assert lib.lde_forward(m.h, p(z0d), None, tsp, 5, 8, p(out), None, None) == -5             # LDE_ERR_NO_WEIGHTS
    assert lib.lde_set_weights(m.h, None, 3) == -1
    # coupled adaptive solves are limited to 4096 trajectories per GPU (one resident workgroup per CU)
    big = Native(make_desc(rhs_kind=O.RHS_MLP, state_dim=2, param_dim=0, layers=(2, 8, 2), batching=O.BATCH_COUPLED))
    big.set_weights(O.mlp_weights((2, 8, 2)))
    zb = torch.zeros((5000, 2), device=dev)
    ob = torch.empty((5, 5000, 2), device=dev)
    assert lib.lde_forward(big.h, p(zb), None, tsp, 5, 5000, p(ob), None, None) == -2          # LDE_ERR_UNSUPPORTED
    assert b"4096" in lib.lde_last_error(big.h)


def test_forward_and_adjoint_are_graph_capturable(o32):
    """After lde_reserve (or one eager call of the same shape) the hot calls allocate nothing and synchronise nothing:
    they can be captured into a hipGraph and replayed on new inputs (MI355X-first: graphs instead of a tracing compiler)."""
    import torch
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    nat, od = _native(abstol=1e-6, reltol=1e-6)
    B, T = 256, 50
    ts = O.time_grid(T)
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    dev = "cuda"
    z0a, La = O.pendulum_inputs(B, seed=1)
    z0b, Lb = O.pendulum_inputs(B, seed=2)
    dz = O.cotangent(T, B, 2)
    z0d, thd, dzd = torch.from_numpy(z0a).to(dev), torch.from_numpy(La).to(dev), torch.from_numpy(dz).to(dev)
    out = torch.empty((T, B, 2), device=dev)
    g0, gL = torch.empty((B, 2), device=dev), torch.empty((B, 1), device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())

    def step():
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.lde_forward(nat.h, p(z0d), p(thd), tsp, T, B, p(out), None, s), nat.h, "fwd")
        L.check(lib.lde_adjoint(nat.h, p(out), p(thd), tsp, T, B, p(dzd), p(g0), p(gL), None, s), nat.h, "adj")

    step()                       # eager warm-up: workspace sized, save-time grid uploaded
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    z0d.copy_(torch.from_numpy(z0b))     # new inputs in the same buffers
    thd.copy_(torch.from_numpy(Lb))
    graph.replay()
    torch.cuda.synchronize()
    zr, _, _ = o32.forward(od, z0b, Lb, ts)
    r0, rL, _, _ = o32.adjoint(O.make_desc(abstol=1e-6, reltol=1e-6, sensealg=O.SENSE_PARALLEL_CHECKPOINTED), zr, Lb, ts, dz)
    assert np.abs(out.cpu().numpy() - zr).max() <= 1e-5
    assert np.abs(g0.cpu().numpy() - r0).max() <= 2e-4 * np.abs(r0).max()
    assert np.abs(gL.cpu().numpy() - rL).max() <= 2e-4 * np.abs(rL).max()
