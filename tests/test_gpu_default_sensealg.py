"""GPU: the plug-in struct's DEFAULT gradient is the reference's.

`Pendulum()` carries `ForwardDiffSensitivity()` [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into `solve` at
[REF src/models/GOKU.jl:107, :121]: the exact derivative of the discrete solve on its accepted steps. Since round 6 that is what
`lde_problem_desc_default` (C ABI) and `Pendulum()` / `Pendulum_friction()` (Python mirror) select — LDE_SENSE_DISCRETE — without being
asked; `NODE` keeps DiffEqFlux's InterpolatingAdjoint [REF src/models/LatentODE.jl:67-70]; the continuous adjoints stay selectable.

Also here: what the host's pullback does when a solve accepts more steps than its step record holds: the reference
differentiates any solve up to maxiters, so the autograd bridge looks at the record's counts, grows the record and repeats the
(deterministic) forward solve instead of handing NaN gradients to the optimiser.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
NT = max(1, min(64, (os.cpu_count() or 2) // 2))


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_c_abi_default_desc_is_the_discrete_sensitivity_at_the_metric_shape(o32, o64):
    """lde_problem_desc_default, untouched, at BASELINE.json's metric shape (B = 256, T = 50, Tsit5 1e-6 / 1e-3): the pullback is the
    discrete sweep — equal to the oracle's derivative of the SAME recorded steps to 1e-4 (f32 and f64), ẑ to 2e-5."""
    from latentdiffeq_amd import _lib as L
    from tests.gpu_util import Native, copy_desc_to_oracle
    lib = L.load()
    d = L.ProblemDesc()
    assert lib.lde_problem_desc_default(C.byref(d)) == 0 and d.sensealg == L.SENSE_DISCRETE
    nat = Native(d)
    od = copy_desc_to_oracle(d)
    B, T = 256, 50
    z0, Lp = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    z, ret, st = nat.forward(z0, Lp, ts)
    assert (ret == 0).all()
    rec = nat.step_record(0, B)
    assert int(rec["n"].sum()) == st["naccept"]
    zr, _, _, _ = o32.forward_steps(od, z0, Lp, ts, rec=rec, nthreads=NT)
    assert np.abs(z - zr).max() <= 2e-5
    g0, gL, _, sb = nat.adjoint(z, Lp, ts, dz)
    assert sb["nfailed"] == 0 and sb["nreject"] == 0          # a sweep: no step control in the pullback
    r0, rL, _, _ = o32.adjoint_discrete(od, z, Lp, ts, dz, rec, nthreads=NT)
    z64, _, _, _ = o64.forward_steps(od, z0, Lp, ts, rec=rec, nthreads=NT)
    t0, tL, _, _ = o64.adjoint_discrete(od, z64, Lp, ts, dz, rec, nthreads=NT)
    errs = dict(dz0=_rel(g0, r0), dL=_rel(gL, rL), dz0_64=_rel(g0, t0), dL_64=_rel(gL, tL))
    assert all(v <= 1e-4 for v in errs.values()), errs
    # and it is NOT the continuous adjoint: at reltol 1e-3 the two definitions differ by more than arithmetic
    c0, cL, _, _ = o64.adjoint(O.make_desc(sensealg=O.SENSE_BACKSOLVE_CHECKPOINTED), z64, Lp, ts, dz, nthreads=NT)
    assert _rel(t0, c0) > 1e-4 or _rel(tL, cL) > 1e-4


def _grads(dq, z0, Lp, ts, dz, model_type=None):
    import torch
    import latentdiffeq_amd as la
    dec = la.Decoder(model_type or la.GOKU_basic(), (None, dq, None))
    a = torch.tensor(z0.T.copy(), device="cuda", requires_grad=True)
    b = torch.tensor(Lp.T.copy(), device="cuda", requires_grad=True)
    zh = la.diffeq_layer(dec, (a, b), ts)
    (zh * torch.tensor(dz, device="cuda").permute(2, 1, 0)).sum().backward()
    torch.cuda.synchronize()
    return zh.detach().permute(2, 1, 0).cpu().numpy(), a.grad.cpu().numpy().T, b.grad.cpu().numpy().T


def test_pendulum_struct_default_is_forwarddiff_exactly(o64):
    """`Pendulum()` ≡ `Pendulum(sensealg=DiscreteSensitivity())` bit for bit; `ParallelAdjoint()` (rounds 1–5's default) is still there and
    agrees with it to solver tolerance; `NODE` keeps the continuous adjoint."""
    import latentdiffeq_amd as la
    from latentdiffeq_amd import _lib as L
    assert la.Pendulum().sensealg.code == L.SENSE_DISCRETE and la.Pendulum_friction().sensealg.code == L.SENSE_DISCRETE
    assert isinstance(la.Pendulum().sensealg, la.ForwardDiffSensitivity)
    assert la.NODE(4, hidden_dim=8).sensealg.code == L.SENSE_BACKSOLVE_CHECKPOINTED
    B, T = 96, 50
    z0, Lp = O.pendulum_inputs(B, seed=5)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    za, a0, aL = _grads(la.Pendulum(), z0, Lp, ts, dz)
    zb, b0, bL = _grads(la.Pendulum(sensealg=la.DiscreteSensitivity()), z0, Lp, ts, dz)
    assert np.array_equal(za, zb) and np.array_equal(a0, b0) and np.array_equal(aL, bL)
    zc, c0, cL = _grads(la.Pendulum(sensealg=la.ParallelAdjoint()), z0, Lp, ts, dz)
    assert np.array_equal(za, zc)                                # the forward solve does not depend on the sensealg
    assert 0 < _rel(a0, c0) <= 2e-2 and 0 < _rel(aL, cL) <= 2e-2  # two definitions of the gradient at reltol 1e-3
    # tight tolerances: the two definitions meet
    kw = dict(abstol=1e-7, reltol=1e-7)
    _, d0, dL = _grads(la.Pendulum(**kw), z0, Lp, ts, dz)
    _, e0, eL = _grads(la.Pendulum(sensealg=la.ParallelAdjoint(), **kw), z0, Lp, ts, dz)
    assert _rel(d0, e0) <= 2e-4 and _rel(dL, eL) <= 2e-4


def test_pullback_regrows_a_step_record_that_overflowed():
    """A solve that accepts more steps than the record holds: the raw C ABI answers NaN gradients + LDE_RET_MAXITERS (never a truncated
    sweep); the autograd bridge (`_SolveFn.backward`) sees the overflow in the record's counts, raises "record_capacity", repeats the forward
    solve and returns the gradient a large record gives — bit for bit. `check_record = False` hands the NaNs through."""
    import latentdiffeq_amd as la
    from latentdiffeq_amd import _lib as L
    B, T = 40, 50
    z0, Lp = O.pendulum_inputs(B, seed=7)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    kw = dict(abstol=1e-8, reltol=1e-8)
    big = la.Pendulum(**kw)
    zr, r0, rL = _grads(big, z0, Lp, ts, dz)
    st = big._native().stats(0)
    assert st["max_steps"] > 24                                     # the solve needs more than the small record below
    small = la.Pendulum(**kw)
    h = small._native()
    L.check(h.lib.lde_set_option(h.ptr, b"record_capacity", 8.0), h.ptr, "lde_set_option")
    zs, s0, sL = _grads(small, z0, Lp, ts, dz)
    assert np.array_equal(zs, zr) and np.array_equal(s0, r0) and np.array_equal(sL, rL)
    cap = C.c_double(0)
    h.lib.lde_get_option(h.ptr, b"record_capacity", C.byref(cap))
    assert cap.value > 24                                            # the handle keeps the larger capacity for the next solve
    zs2, s02, _ = _grads(small, z0, Lp, ts, dz)                      # … which then fits at once
    assert np.array_equal(s02, r0)
    raw = la.Pendulum(**kw)
    raw.check_record = False
    hr = raw._native()
    L.check(hr.lib.lde_set_option(hr.ptr, b"record_capacity", 8.0), hr.ptr, "lde_set_option")
    _, n0, nL = _grads(raw, z0, Lp, ts, dz)
    assert np.isnan(n0).all() and np.isnan(nL).all() and hr.stats(1)["nfailed"] == B


def test_step_record_status_entry_point():
    """lde_step_record_capacity / lde_step_record_status on the handle's own record and on a caller-owned one."""
    import torch
    from latentdiffeq_amd import _lib as L
    from tests.gpu_util import Native, make_desc
    nat = Native(make_desc(sensealg=L.SENSE_DISCRETE))
    lib = nat.lib
    B, T = 32, 50
    z0, Lp = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    assert lib.lde_step_record_capacity(nat.h, T) == 200             # max(64, 4·T)
    _, _, st = nat.forward(z0, Lp, ts)
    nmax, cap = C.c_int32(-1), C.c_int32(-1)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.lde_step_record_status(nat.h, None, B, T, C.byref(nmax), C.byref(cap), s) == 0
    rec = nat.step_record(0, B)
    assert nmax.value == int(rec["n"].max()) and cap.value == 200
    assert lib.lde_step_record_status(nat.h, None, B + 1, T, C.byref(nmax), C.byref(cap), s) == -1   # no record of that shape
    nat.set_option("record_capacity", 4)
    assert lib.lde_step_record_capacity(nat.h, T) == 4
    buf = torch.empty(int(lib.lde_step_record_bytes(nat.h, B, T)), dtype=torch.uint8, device="cuda")
    assert lib.lde_set_step_record(nat.h, C.c_void_p(buf.data_ptr()), buf.numel()) == 0
    nat.forward(z0, Lp, ts)
    assert lib.lde_step_record_status(nat.h, C.c_void_p(buf.data_ptr()), B, T, C.byref(nmax), C.byref(cap), s) == 0
    assert cap.value == 4 and nmax.value == int(rec["n"].max()) > 4   # counts run on past the capacity
    lib.lde_set_step_record(nat.h, None, 0)


@pytest.mark.parametrize("B,kind,sense,kernel", [
    (384, "pendulum", "discrete", "k_pend_forward_lp"),       # four dense-output waves (B ≤ 512)
    (640, "pendulum", "discrete", "k_pend_forward_lp"),       # three (B > 512): the other instantiation, recording
    (1000, "pendulum", "continuous", "k_pend_forward_lp"),    # … and not recording
    (1100, "pendulum", "discrete", "k_pend_forward_ws"),      # beyond the threshold: the next mapping that writes step records
    (1500, "pendulum", "continuous", "k_pend_forward_tl"),    # … resp. lanes = save times (B ≤ 2 048 without a record)
    (640, "friction", "discrete", "k_pend_forward_sh"),       # every other solve of a trajectory per workgroup: B ≤ 768 when it records
    (640, "friction", "continuous", "k_pend_forward_tl"),     # … B ≤ 256 when it does not
])
def test_default_mapping_of_the_batches_between_the_metric_and_the_large_ones(o32, B, kind, sense, kernel):
    """The forward mapping the library picks BY ITSELF (option "pend_sh_max_b" = −1: the measured thresholds, abl/lp_midB.py) at batches between
    the metric's 256 and the large ones, named by lde_last_kernel — and held to the oracle like every other mapping: ẑ on the kernel's own recorded
    steps to 2e-5, the gradient of those steps to 1e-4 (discrete); free-running against the oracle's own solve to 3e-4 (continuous)."""
    from latentdiffeq_amd import _lib as L
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    rhs = L.RHS_PENDULUM if kind == "pendulum" else L.RHS_PENDULUM_FRICTION
    d = make_desc(rhs_kind=rhs, sensealg=L.SENSE_DISCRETE if sense == "discrete" else L.SENSE_PARALLEL_CHECKPOINTED)
    nat = Native(d)
    od = copy_desc_to_oracle(d)
    T = 50
    z0, Lp = O.pendulum_inputs(B, seed=11)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    z, ret, st = nat.forward(z0, Lp, ts)
    assert (ret == 0).all()
    assert nat.lib.lde_last_kernel(nat.h, 0).decode() == kernel
    if sense == "discrete":
        rec = nat.step_record(0, B)
        assert int(rec["n"].sum()) == st["naccept"]
        zr, _, _, _ = o32.forward_steps(od, z0, Lp, ts, rec=rec, nthreads=NT)
        assert np.abs(z - zr).max() <= 2e-5
        g0, gL, _, sb = nat.adjoint(z, Lp, ts, dz)
        assert sb["nfailed"] == 0
        r0, rL, _, _ = o32.adjoint_discrete(od, z, Lp, ts, dz, rec, nthreads=NT)
        assert _rel(g0, r0) <= 1e-4 and _rel(gL, rL) <= 1e-4
    else:
        zo, _, _ = o32.forward(od, z0, Lp, ts, nthreads=NT)
        assert np.abs(z - zo).max() <= 3e-4
