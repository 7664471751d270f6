"""GPU: LDE_SENSE_DISCRETE — the exact derivative of the discrete solve, what the reference's GOKU default
`ForwardDiffSensitivity()` delivers [REF examples/pendulum_friction-less/pendulum.jl:11], [REF src/models/GOKU.jl:107, :121] —
against the oracle's restatement (oracle/lde_oracle.c: discrete_block, pinned by torch autograd and finite differences in
tests/test_oracle_discrete.py).

Kernel and checker are put on the SAME discrete solve: the step record the kernel's forward solve wrote (t_n, dt_n per accepted
step) is handed to the oracle, which replays exactly those steps (`forward_steps(rec=…)`) and differentiates them
(`adjoint_discrete`). What is compared is therefore arithmetic, not two step-size controllers:
    |Δẑ| ≤ 2e-5 (f32 round-off through the stages),
    every gradient ≤ 1e-4 of its largest entry against the f32 oracle AND against the float64 oracle on the same steps —
also for relu networks at the reference's default tolerances (reltol 1e-3), where the continuous-adjoint gates are 1e-2.

The one thing arithmetic cannot pin: a relu unit whose pre-activation lies within f32 round-off of zero is switched on in one
implementation and off in the other (different summation orders), and that trajectory's gradient then differs by a finite amount.
At the BASELINE sizes this happens to about one unit evaluation in 10⁷ (measured, abl/disc_diag.py: c3 at B = 1024 — 21 M unit
evaluations — 2 trajectories; the reference's NODE shape at B = 64 — 1; c4 at B = 512 — 0; the tanh twins of all three: 0, every
gradient within 3e-6). The full-size relu tests therefore hold every trajectory to 1e-4 EXCEPT those the float64 oracle finds within
1e-5 (relative) of a kink on the same steps — these are named by the oracle, must be few, and a mismatch anywhere else fails.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
NT = max(1, min(64, (os.cpu_count() or 2) // 2))
Z_TOL, G_TOL = 2e-5, 1e-4


def _native(W, **kw):
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    kw.setdefault("sensealg", O.SENSE_DISCRETE)
    d = make_desc(**kw)
    nat = Native(d)
    if W is not None:
        nat.set_weights(W)
    return nat, copy_desc_to_oracle(d)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


KINK = 1e-5     # a relu pre-activation this close to zero (relative to Σ|w·x| + |b|) may fall on either side in f32


def _check(nat, od, o32, o64, z0, theta, ts, dz, W, z_tol=Z_TOL, g_tol=G_TOL, check64=True, kinks=False):
    B = z0.shape[0]
    z, ret, st = nat.forward(z0, theta, ts)
    assert (ret == 0).all() and st["nfailed"] == 0
    rec = nat.step_record(0, B)
    assert int(rec["n"].min()) >= 1 and int(rec["n"].sum() if od.batching == O.BATCH_PER_TRAJECTORY else rec["n"][0]) == st["naccept"]
    # the oracle on the kernel's steps
    zr, retr, _, _ = o32.forward_steps(od, z0, theta, ts, W=W, rec=rec, nthreads=NT)
    assert (retr == 0).all()
    assert np.abs(z - zr).max() <= z_tol, np.abs(z - zr).max()
    g0, gth, gW, sb = nat.adjoint(z, theta, ts, dz)
    assert sb["nfailed"] == 0 and sb["nreject"] == 0
    r0, rth, rW, ri = o32.adjoint_discrete(od, z, theta, ts, dz, rec, W=W, nthreads=NT, margins=kinks)
    if kinks:
        # per-trajectory errors; the trajectories the oracle finds at a relu kink are exempt from the 1e-4 gate (module docstring)
        per = np.abs(g0 - r0).max(axis=1) / np.abs(r0).max()
        if theta is not None:
            per = np.maximum(per, np.abs(gth - rth).max(axis=1) / np.abs(rth).max())
        near = ri["margins"] < KINK
        off = per > g_tol
        assert np.median(per) <= 2e-6, np.median(per)
        assert not (off & ~near).any(), ("a gradient differs away from any relu kink", np.nonzero(off & ~near)[0][:8], per[off & ~near][:8])
        assert off.sum() <= max(2, B // 50) and per.max() <= 0.1, (int(off.sum()), per.max())
        if W is not None:
            assert _rel(gW, rW) <= (g_tol if not off.any() else 1e-2), _rel(gW, rW)
        return z, rec, (g0, gth, gW), st, sb
    errs = {"dz0": _rel(g0, r0)}
    if theta is not None:
        errs["dtheta"] = _rel(gth, rth)
    if W is not None:
        errs["dW"] = _rel(gW, rW)
    if check64:
        W64 = None if W is None else W.astype(np.float64)
        z64, _, _, _ = o64.forward_steps(od, z0, theta, ts, W=W64, rec=rec, nthreads=NT)
        t0, tth, tW, _ = o64.adjoint_discrete(od, z64, theta, ts, dz, rec, W=W64, nthreads=NT)
        errs["dz0_64"] = _rel(g0, t0)
        if theta is not None:
            errs["dtheta_64"] = _rel(gth, tth)
        if W is not None:
            errs["dW_64"] = _rel(gW, tW)
    bad = {k: v for k, v in errs.items() if not v <= g_tol}
    assert not bad, (bad, errs)
    return z, rec, (g0, gth, gW), st, sb


# ---------------------------------------------------------------------------------------------------- analytic right-hand sides
@pytest.mark.parametrize("kind,solver,B", [
    (O.RHS_PENDULUM, O.SOLVER_TSIT5, 64),            # c1 / the metric's mapping (one trajectory per workgroup)
    (O.RHS_PENDULUM, O.SOLVER_TSIT5, 256),           # the metric config
    (O.RHS_PENDULUM, O.SOLVER_TSIT5, 1000),          # the lane-per-trajectory mapping, ragged last workgroup
    (O.RHS_PENDULUM_FRICTION, O.SOLVER_TSIT5, 96),
    (O.RHS_PENDULUM, O.SOLVER_RK4, 80),              # fixed steps that do not hit the save times: the Hermite interpolant's pullback
    (O.RHS_PENDULUM_FRICTION, O.SOLVER_RK4, 300),
])
@pytest.mark.parametrize("mapping", ["steps_side_by_side", "lane_per_trajectory"])
def test_goku_discrete_matches_oracle_on_the_same_steps(o32, o64, kind, solver, B, mapping):
    """Both mappings of the pullback: k_pend_adjoint_disc_tp (a wave per trajectory, lanes = steps × tangents: the default up to
    option "pend_disc_tp_max_b") and k_pend_adjoint_disc (a lane per trajectory, reverse accumulation)."""
    kw = dict(rhs_kind=kind, solver=solver)
    if solver == O.SOLVER_RK4:
        kw.update(adaptive=0, dt=0.13)
    nat, od = _native(None, **kw)
    if mapping == "lane_per_trajectory":
        nat.set_option("pend_disc_tp_max_b", 0)
    z0, L = O.pendulum_inputs(B, seed=3)
    ts = O.time_grid(50)
    dz = O.cotangent(50, B, 2)
    _check(nat, od, o32, o64, z0, L, ts, dz, None)


@pytest.mark.parametrize("B,options,tol", [
    (200, {}, 1e-6),                                       # k_pend_forward_lp (round 6): a trajectory per workgroup, the stepping wave — lane pairs — writes the round's records
    (200, {"pend_lp": 0}, 1e-6),                           # k_pend_forward_sh: the same mapping with every lane carrying the whole solve
    (200, {"pend_sh_max_b": 0}, 1e-6),                     # k_pend_forward_ws: 64 trajectories per workgroup, each stepper lane its own records
    (1000, {}, 1e-6),                                      # (the default at this batch: k_pend_forward_lp with three dense-output waves, B > 512)
    (600, {"record_capacity": 512}, 3e-9),                 # … and several rounds of ITS ring
    (1000, {"pend_ws": 0}, 1e-6),                          # k_pend_forward: a lane per trajectory, records at accept
    (200, {"record_capacity": 64}, 3e-9),                  # ≈ 100 steps: records from several rounds of the ring (48 per round), then overflow → see below
    (200, {"pend_lp": 0, "record_capacity": 64}, 3e-9),    # the same through k_pend_forward_sh (the default at this shape is k_pend_forward_lp)
    (200, {"pend_sh_max_b": 0, "record_capacity": 512}, 3e-9),   # several rounds of k_pend_forward_ws's ring (96 per round)
    (1000, {"pend_tl_max_b": 0, "pend_sh_max_b": 0, "pend_ws": 0, "pend_lb_min_b": 0}, 1e-6),   # the large-batch form (rows of ẑ through the LDS ring), forced at a test-sized batch
])
def test_goku_discrete_every_recording_forward_mapping(o32, o64, B, options, tol):
    """Every forward mapping that writes step records: the record reproduces the kernel's own solve in the oracle (|Δẑ| ≤ 2e-5 on the
    recorded steps — a wrong start time or start state of a single step would show) and the pullback on it matches."""
    nat, od = _native(None, abstol=tol, reltol=tol)
    for k, v in options.items():
        nat.set_option(k, v)
    z0, L = O.pendulum_inputs(B, seed=9)
    ts = O.time_grid(50)
    dz = O.cotangent(50, B, 2)
    if options.get("record_capacity") == 64:               # too small for this solve: NaN gradients, never a truncated sweep
        z, ret, st = nat.forward(z0, L, ts)
        g0, gth, _, sb = nat.adjoint(z, L, ts, dz)
        assert st["max_steps"] > 64 and np.isnan(g0).any()
        nat.set_option("record_capacity", 512)
    _, rec, _, st, _ = _check(nat, od, o32, o64, z0, L, ts, dz, None)
    if tol < 1e-7:
        assert int(rec["n"].max()) > 96


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("solver", [O.SOLVER_TSIT5, O.SOLVER_RK4])
def test_goku_discrete_smallest_grids_and_batches(o32, o64, T, solver):
    """One, two and three save times (T = 1: no step at all — the pullback is the cotangent of ẑ₀ itself), five trajectories (less than a
    workgroup row of the XCD-aware map), both pullback mappings."""
    B = 5
    kw = dict(solver=solver)
    if solver == O.SOLVER_RK4:
        kw.update(adaptive=0, dt=0.02)
    z0, L = O.pendulum_inputs(B, seed=12)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    res = []
    for tp in (1 << 20, 0):
        nat, od = _native(None, **kw)
        nat.set_option("pend_disc_tp_max_b", tp)
        if T == 1:
            z, ret, st = nat.forward(z0, L, ts)
            g0, gth, _, sb = nat.adjoint(z, L, ts, dz)
            assert np.array_equal(z[0], z0) and np.array_equal(g0, dz[0]) and (gth == 0).all() and sb["nfailed"] == 0
        else:
            _, _, (g0, gth, _), _, _ = _check(nat, od, o32, o64, z0, L, ts, dz, None)
        res.append((g0, gth))
    assert np.allclose(res[0][0], res[1][0], rtol=0, atol=1e-6 * max(1e-30, np.abs(res[1][0]).max()))
    assert np.allclose(res[0][1], res[1][1], rtol=0, atol=1e-6 * max(1e-30, np.abs(res[1][1]).max()) + 1e-12)


@pytest.mark.parametrize("T", [3000, 4000])
def test_goku_discrete_thousands_of_save_times(o32, o64, T):
    """T = 3000: the steps-side-by-side kernel with 51 KB of LDS (many save times per step, chunks of twelve per round trip);
    T = 4000: beyond what a launch gets of LDS without asking — the lane-per-trajectory kernel serves it whatever the option says."""
    B = 6
    z0, L = O.pendulum_inputs(B, seed=13)
    ts = np.arange(T) * (2.45 / (T - 1))
    dz = O.cotangent(T, B, 2)
    out = []
    for tp in (1 << 20, 0):
        nat, od = _native(None)
        nat.set_option("pend_disc_tp_max_b", tp)
        _, _, g, _, _ = _check(nat, od, o32, o64, z0, L, ts, dz, None)
        out.append(g)
    assert _rel(out[0][0], out[1][0]) <= 1e-5 and _rel(out[0][1], out[1][1]) <= 1e-5


@pytest.mark.parametrize("kind,solver", [(O.RHS_PENDULUM, O.SOLVER_TSIT5), (O.RHS_PENDULUM_FRICTION, O.SOLVER_TSIT5), (O.RHS_PENDULUM, O.SOLVER_RK4)])
def test_goku_discrete_long_records_and_ragged_save_grids(o32, o64, kind, solver):
    """Records longer than one round of the steps-side-by-side kernel (21 steps per round; here 60 … 200), 150 save times on a ragged
    grid (several per step, some steps with none, more save times than lanes), both mappings against the oracle and each other."""
    B, T = 70, 150
    rng = np.random.default_rng(11)
    ts = np.concatenate([[0.0], np.cumsum(rng.uniform(0.002, 0.05, T - 1) * rng.choice([1.0, 1.0, 4.0], T - 1))])
    kw = dict(rhs_kind=kind, solver=solver, abstol=1e-8, reltol=1e-8)
    if solver == O.SOLVER_RK4:
        kw.update(adaptive=0, dt=0.031)
    z0, L = O.pendulum_inputs(B, seed=4)
    dz = O.cotangent(T, B, 2)
    out = []
    for tp in (1 << 20, 0):
        nat, od = _native(None, **kw)
        nat.set_option("pend_disc_tp_max_b", tp)
        nat.set_option("record_capacity", 512)
        _, rec, g, st, sb = _check(nat, od, o32, o64, z0, L, ts, dz, None)
        assert int(rec["n"].max()) > 42
        out.append(g)
    assert _rel(out[0][0], out[1][0]) <= 1e-5 and _rel(out[0][1], out[1][1]) <= 1e-5


def test_goku_discrete_agrees_with_the_continuous_adjoint_to_solver_tolerance(o64):
    """The two definitions of the gradient (discretise-then-differentiate vs the continuous adjoint) meet as the tolerance tightens."""
    B, T = 128, 50
    z0, L = O.pendulum_inputs(B, seed=5)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    out = {}
    for sa in (O.SENSE_DISCRETE, O.SENSE_PARALLEL_CHECKPOINTED):
        nat, _ = _native(None, sensealg=sa, abstol=1e-7, reltol=1e-7)
        z, ret, _ = nat.forward(z0, L, ts)
        out[sa] = nat.adjoint(z, L, ts, dz)
    a, c = out[O.SENSE_DISCRETE], out[O.SENSE_PARALLEL_CHECKPOINTED]
    assert _rel(a[0], c[0]) < 2e-4 and _rel(a[1], c[1]) < 2e-4


def test_goku_discrete_failure_semantics_and_record_overflow(o32):
    B, T = 64, 50
    z0, L = O.pendulum_inputs(B, seed=2)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    nat, od = _native(None)
    z, ret, _ = nat.forward(z0, L, ts)
    zb = z.copy()
    zb[:, 3, :] = np.nan                      # a failed trajectory is a NaN block [REF src/models/GOKU.jl:114] ⇒ zero gradient
    g0, gth, _, sb = nat.adjoint(zb, L, ts, dz)
    assert sb["nfailed"] == 1 and (g0[3] == 0).all() and gth[3, 0] == 0 and np.isfinite(g0).all()
    # a record too small for the solve: NaN gradients and a retcode — never a truncated sweep
    nat2, _ = _native(None)
    nat2.set_option("record_capacity", 4)
    z2, ret2, _ = nat2.forward(z0, L, ts)
    assert (ret2 == 0).all() and np.array_equal(z2, z)
    g0, gth, _, sb = nat2.adjoint(z2, L, ts, dz)
    assert sb["nfailed"] == B and np.isnan(g0).all() and np.isnan(gth).all()
    # without a forward on this handle the pullback refuses
    nat3, _ = _native(None)
    with pytest.raises(Exception, match="no step record"):
        nat3.adjoint(z, L, ts, dz)


# ---------------------------------------------------------------------------------------------------- MLP right-hand sides
MLP_CASES = {
    # the BASELINE shapes at test-sized batches (full sizes: test_discrete_at_baseline_sizes below)
    "c4_relu_coupled": (dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=(32, 128, 128, 32), batching=O.BATCH_COUPLED), 48),
    "latentode_ref_relu_coupled": (dict(rhs_kind=O.RHS_MLP, state_dim=16, param_dim=0, layers=(16, 200, 200, 16), batching=O.BATCH_COUPLED), 24),
    "c3_pend_plus_mlp_relu": (dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 64, 64, 2)), 40),
    "c2_rk4_relu": (dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=(8, 200, 200, 8), solver=O.SOLVER_RK4, adaptive=0, dt=0.05,
                         batching=O.BATCH_COUPLED), 32),
    "aug_tanh_per_trajectory": (dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, augment_dim=2, layers=(8, 48, 48, 8), activation=O.ACT_TANH,
                                     batching=O.BATCH_PER_TRAJECTORY), 37),
    "five_layers_tanh_coupled": (dict(rhs_kind=O.RHS_MLP, state_dim=5, param_dim=0, layers=(5, 40, 72, 40, 24, 5), activation=O.ACT_TANH,
                                      batching=O.BATCH_COUPLED), 19),
    "rk4_off_grid_per_trajectory": (dict(rhs_kind=O.RHS_MLP, state_dim=4, param_dim=0, layers=(4, 32, 32, 4), activation=O.ACT_TANH,
                                         solver=O.SOLVER_RK4, adaptive=0, dt=0.07, batching=O.BATCH_PER_TRAJECTORY), 20),
}


def _mlp_inputs(kw, B, T=50, seed=7):
    layers = kw["layers"]
    W = O.mlp_weights(layers, seed=seed)
    D = kw.get("state_dim", 2)
    Dp = D + kw.get("augment_dim", 0)
    if kw["rhs_kind"] == O.RHS_PENDULUM_PLUS_MLP:
        z0, theta = O.pendulum_inputs(B, seed=seed)
    else:
        z0, theta = (0.5 * np.random.default_rng(seed).standard_normal((B, D))).astype(np.float32), None
    return W, z0, theta, O.time_grid(T), O.cotangent(T, B, Dp)


@pytest.mark.parametrize("name", list(MLP_CASES))
def test_mlp_discrete_matches_oracle_on_the_same_steps(o32, o64, name):
    kw, B = MLP_CASES[name]
    W, z0, theta, ts, dz = _mlp_inputs(kw, B)
    nat, od = _native(W, **kw)
    _check(nat, od, o32, o64, z0, theta, ts, dz, W)


def test_mlp_discrete_is_bitwise_deterministic_and_accumulates_dW():
    kw, B = MLP_CASES["c4_relu_coupled"]
    W, z0, theta, ts, dz = _mlp_inputs(kw, B)
    nat, _ = _native(W, **kw)
    z, _, _ = nat.forward(z0, theta, ts)
    a = nat.adjoint(z, theta, ts, dz)
    z2, _, _ = nat.forward(z0, theta, ts)
    b = nat.adjoint(z2, theta, ts, dz)
    assert np.array_equal(z, z2) and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])


def test_discrete_uses_far_fewer_evaluations_than_the_continuous_adjoint():
    """The point of the sensealg on the MLP right-hand sides: 2S evaluations per accepted FORWARD step instead of 6–7 per attempt of a
    reverse-time solve with a forced stop at every save time (c4 345 vs 62 forward evaluations)."""
    kw, B = MLP_CASES["c4_relu_coupled"]
    W, z0, theta, ts, dz = _mlp_inputs(kw, B)
    nfe = {}
    for sa in (O.SENSE_DISCRETE, O.SENSE_BACKSOLVE_CHECKPOINTED):
        nat, _ = _native(W, **{**kw, "sensealg": sa})
        z, _, st = nat.forward(z0, theta, ts)
        _, _, _, sb = nat.adjoint(z, theta, ts, dz)
        nfe[sa] = (st["nfe"], sb["nfe"])
    assert nfe[O.SENSE_DISCRETE][1] <= 2 * nfe[O.SENSE_DISCRETE][0] + 14
    assert nfe[O.SENSE_DISCRETE][1] * 2 < nfe[O.SENSE_BACKSOLVE_CHECKPOINTED][1]


@pytest.mark.parametrize("name,B", [("c3_pend_plus_mlp_relu", 1024), ("c4_relu_coupled", 512), ("latentode_ref_relu_coupled", 64),
                                    ("c2_rk4_relu", 256), ("c4_relu_coupled", 4096)])   # (4096: configs[3]'s whole batch on one GPU — the tile kernels)
def test_discrete_at_baseline_sizes(o32, o64, name, B):
    """BASELINE.json configs[2] (B = 1024), configs[3] at one GPU's share (B = 512), the reference's own LatentODE example (B = 64),
    configs[1] (B = 256): relu networks at the reference's DEFAULT tolerances, every trajectory's gradient held to 1e-4 unless the oracle
    finds it at a relu kink (module docstring) — and the tanh twin of each shape, where nothing is exempt: 1e-5."""
    kw, _ = MLP_CASES[name]
    W, z0, theta, ts, dz = _mlp_inputs(kw, B, seed=3)
    nat, od = _native(W, **kw)
    _check(nat, od, o32, o64, z0, theta, ts, dz, W, kinks=True)
    kwt = {**kw, "activation": O.ACT_TANH}
    nat, od = _native(W, **kwt)
    _check(nat, od, o32, o64, z0, theta, ts, dz, W, g_tol=1e-5, check64=False)


# ---------------------------------------------------------------------------------------------------- through the reference's interface
def test_diffeq_layer_with_exact_forwarddiff_sensitivity_two_graphs_in_flight(o32):
    """`Pendulum(sensalg=ForwardDiffSensitivity(exact=True))` through diffeq_layer + autograd; two forwards of ONE diffeq before their
    pullbacks: every graph node carries its own step record (lde_set_step_record)."""
    import torch
    import latentdiffeq_amd as lde
    from latentdiffeq_amd import api
    dq = api.Pendulum(sensalg=api.ForwardDiffSensitivity(exact=True))
    dec = api.Decoder(api.GOKU_basic(), (None, dq, None))
    B, T = 48, 50
    ts = O.time_grid(T)
    outs, leaves = [], []
    for seed in (1, 2):
        z0, L = O.pendulum_inputs(B, seed=seed)
        zt = torch.tensor(z0.T.copy(), device="cuda", requires_grad=True)      # [D, B]
        Lt = torch.tensor(L.T.copy(), device="cuda", requires_grad=True)
        outs.append(api.diffeq_layer(dec, (zt, Lt), ts))
        leaves.append((zt, Lt, z0, L))
    dzs = [O.cotangent(T, B, 2, seed=10 + i) for i in range(2)]
    loss = sum((o * torch.tensor(dz, device="cuda").permute(2, 1, 0)).sum() for o, dz in zip(outs, dzs))
    loss.backward()
    d = O.make_desc(sensealg=O.SENSE_DISCRETE)
    for (zt, Lt, z0, L), dz in zip(leaves, dzs):
        z, _, rec, _ = o32.forward_steps(d, z0, L, ts)
        r0, rth, _, _ = o32.adjoint_discrete(d, z, L, ts, dz, rec)
        # (the oracle on its OWN steps here: two adaptive f32 solves — the gate is the solver's tolerance, the point is which record was used)
        assert _rel(zt.grad.T.cpu().numpy(), r0) < 5e-3 and _rel(Lt.grad.T.cpu().numpy(), rth) < 5e-3


def test_latentode_diffeq_layer_with_discrete_sensitivity_reaches_the_network_parameters(o64):
    """`NODE(…, sensealg=DiscreteSensitivity())` through diffeq_layer + autograd [REF src/models/LatentODE.jl:61-78]: the gradient of the
    network parameters (every Linear's weight and bias) and of ẑ₀ against the float64 oracle's discrete sweep on its own steps, at a
    tolerance where the two adaptive solves' step sequences no longer matter (tanh: no kinks)."""
    import torch
    from latentdiffeq_amd import api
    B, T, D, H = 24, 30, 4, 48
    torch.manual_seed(3)
    dq = api.NODE(D, hidden_dim=H, device="cuda", sensealg=api.DiscreteSensitivity(), activation="tanh", abstol=1e-7, reltol=1e-7)
    dec = api.Decoder(api.LatentODE(), (None, dq, None))
    rng = np.random.default_rng(5)
    z0 = rng.standard_normal((B, D)).astype(np.float32)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, D)
    zt = torch.tensor(z0.T.copy(), device="cuda", requires_grad=True)
    zhat = api.diffeq_layer(dec, zt, ts)                                            # [D, B, T]
    (zhat * torch.tensor(dz, device="cuda").permute(2, 1, 0)).sum().backward()
    W = dq.flat_weights().detach().cpu().numpy().astype(np.float64)
    d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=(D, H, H, D), activation=O.ACT_TANH, batching=O.BATCH_COUPLED,
                    sensealg=O.SENSE_DISCRETE, abstol=1e-7, reltol=1e-7)
    z, ret, rec, _ = o64.forward_steps(d, z0, None, ts, W=W)
    r0, _, rW, _ = o64.adjoint_discrete(d, z, None, ts, dz, rec, W=W)
    assert np.abs(zhat.detach().permute(2, 1, 0).cpu().numpy() - z).max() <= 2e-5
    assert _rel(zt.grad.T.cpu().numpy(), r0) <= 1e-3
    gW = torch.cat([torch.cat([m.weight.grad.t().reshape(-1), m.bias.grad]) for m in dq.dudt if isinstance(m, torch.nn.Linear)]).cpu().numpy()
    assert gW.shape == rW.shape and _rel(gW, rW) <= 1e-3, _rel(gW, rW)
