"""GPU parity at the batch sizes BASELINE.json names (the other GPU tests use fixture-sized batches): c3 at B = 1024,
c2's adjoint at B = 256, c4 coupled at B = 512 and at B = 4096 (256 workgroups through the grid-wide sum of the coupled
adaptive controller, launched cooperatively). The oracle runs the whole batch with OpenMP (per-trajectory mode over
trajectories, coupled mode over the columns of every stage evaluation) — seconds on the GPU box's host cores.
Every case also checks bitwise run-to-run determinism of the kernels.

Tolerances are the ones of tests/test_gpu_mlp.py for the same workloads at small B (stated there); the float64 truth is
taken on a column subsample where the control mode allows it (per-trajectory)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
NT = max(1, min(64, (os.cpu_count() or 2) // 2))


def _native(W, **kw):
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    d = make_desc(**kw)
    nat = Native(d)
    if W is not None:
        nat.set_weights(W)
    return nat, copy_desc_to_oracle(d)


def _z0(B, D, seed=1):
    return (0.5 * np.random.default_rng(seed).standard_normal((B, D))).astype(np.float32)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_c3_full_batch_1024(o32, o64):
    """BASELINE.json configs[2]: GOKU pendulum + 2-64-64-2 MLP, Tsit5, per-trajectory control, B = 1024, with adjoint."""
    layers = (2, 64, 64, 2)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
    nat, od = _native(W, **kw)
    B, T = 1024, 50
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    z, ret, st = nat.forward(z0, L, ts)
    g0, gL, gW, sb = nat.adjoint(z, L, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0 and sb["nfailed"] == 0
    zr, retr, info = o32.forward(od, z0, L, ts, W=W, nthreads=NT)
    r0, rL, rW, infob = o32.adjoint(od, z, L, ts, dz, W=W, nthreads=NT)
    per_traj = np.abs(z - zr).max(axis=(0, 2))
    assert abs(st["naccept"] - info["naccept"]) <= 0.05 * info["naccept"] + 2
    assert abs(sb["naccept"] - infob["naccept"]) <= 0.1 * infob["naccept"] + 2
    # float64 truth on every 16th column (per-trajectory control: a column's solve does not depend on the others)
    sub = np.arange(0, B, 16)
    d64 = O.make_desc(**{**kw, "abstol": 1e-10, "reltol": 1e-10})
    W64 = W.astype(np.float64)
    z64, _, _ = o64.forward(d64, z0[sub], L[sub], ts, W=W64, nthreads=NT)
    t0, tL, _, _ = o64.adjoint(d64, z64, L[sub], ts, dz[:, sub], W=W64, nthreads=NT)
    # the gates of tests/test_gpu_mlp.py::_check_forward: at the default tolerance two correct f32 solves differ by at most about
    # the solver's own error e_o (relu kinks + reltol 1e-3: a few 1e-3 here), and the kernel is no farther from the truth than 1.5 e_o
    scale = max(1.0, np.abs(zr).max())
    e_o = np.abs(zr[:, sub] - z64).max()
    assert np.abs(z[:, sub] - zr[:, sub]).max() <= max(3e-4 * scale, 1.5 * e_o)
    assert np.abs(z[:, sub] - z64).max() <= 1.5 * e_o + 1e-5 * scale
    assert per_traj.max() <= max(1e-3 * scale, 3.0 * e_o) and np.quantile(per_traj, 0.9) <= max(3e-4 * scale, e_o)   # all 1024 columns
    # default tolerance + relu: gradients of two correct f32 solves agree to about 1 % (tests/test_gpu_mlp.py, B = 100)
    for g, r, t, what in ((g0[sub], r0[sub], t0, "dz0"), (gL[sub], rL[sub], tL, "dL")):
        s = np.abs(t).max()
        assert np.abs(g - r).max() <= 1e-2 * s, what
        assert np.abs(g - t).max() <= 1.5 * np.abs(r - t).max() + 5e-3 * s, what
    assert _rel(g0, r0) <= 1e-2 and _rel(gL, rL) <= 1e-2 and _rel(gW, rW) <= 1e-2
    z2, _, _ = nat.forward(z0, L, ts)
    h0, hL, hW, _ = nat.adjoint(z2, L, ts, dz)
    assert np.array_equal(z, z2) and np.array_equal(g0, h0) and np.array_equal(gL, hL) and np.array_equal(gW, hW)


def test_c2_adjoint_full_batch_256(o32, o64):
    """BASELINE.json configs[1]: LatentODE D = 8, 8-200-200-8, fixed-step RK4 dt = 0.05, B = 256 — the adjoint at full size."""
    layers = (8, 200, 200, 8)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=layers, solver=O.SOLVER_RK4, adaptive=0, dt=0.05,
              batching=O.BATCH_COUPLED)
    nat, od = _native(W, **kw)
    B, T = 256, 50
    z0, ts = _z0(B, 8), O.time_grid(T)
    dz = O.cotangent(T, B, 8)
    z, _, st = nat.forward(z0, None, ts)
    g0, _, gW, sb = nat.adjoint(z, None, ts, dz)
    zr, _, _ = o32.forward(od, z0, None, ts, W=W, nthreads=NT)
    r0, _, rW, info = o32.adjoint(od, z, None, ts, dz, W=W, nthreads=NT)
    assert sb["naccept"] == info["naccept"] == 49 and sb["nfailed"] == 0
    assert np.abs(z - zr).max() <= 2e-5 * max(1.0, np.abs(zr).max())               # no controller: f32 round-off only
    # float64 with the SAME step size: what is left is f32 round-off of the two f32 implementations
    d64 = O.make_desc(**kw)
    W64 = W.astype(np.float64)
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W64, nthreads=NT)
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W64, nthreads=NT)
    for g, r, t, what in ((g0, r0, t0, "dz0"), (gW, rW, tW, "dW")):
        s = np.abs(t).max()
        assert np.abs(g - r).max() <= 2e-4 * s, what
        assert np.abs(g - t).max() <= 1.5 * np.abs(r - t).max() + 2e-4 * s, what
    h0, _, hW, _ = nat.adjoint(z, None, ts, dz)
    assert np.array_equal(g0, h0) and np.array_equal(gW, hW)


@pytest.mark.parametrize("B", [512, 4096])
def test_c4_coupled_full_batches(o32, B):
    """BASELINE.json configs[3]: LatentODE D = 32, 32-128-128-32, Tsit5, ONE coupled solve on the [32 × B] state (RMS norm over all
    entries): B = 512 is one GPU's share of the 4096, B = 4096 puts 256 workgroups through the grid-wide sum of every step
    (cooperative launch). Against the oracle's coupled solve of the same batch; bitwise determinism across runs."""
    layers = (32, 128, 128, 32)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=O.BATCH_COUPLED)
    nat, od = _native(W, **kw)
    T = 50
    z0, ts = _z0(B, 32), O.time_grid(T)
    dz = O.cotangent(T, B, 32)
    z, ret, st = nat.forward(z0, None, ts)
    g0, _, gW, sb = nat.adjoint(z, None, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0 and sb["nfailed"] == 0
    zr, _, info = o32.forward(od, z0, None, ts, W=W, nthreads=NT)
    r0, _, rW, infob = o32.adjoint(od, z, None, ts, dz, W=W, nthreads=NT)
    assert abs(st["naccept"] - info["naccept"]) <= 0.1 * info["naccept"] + 2
    assert abs(sb["naccept"] - infob["naccept"]) <= 0.1 * infob["naccept"] + 2
    scale = max(1.0, np.abs(zr).max())
    assert np.abs(z - zr).max() <= 3e-4 * scale                                      # default tolerance (tests/test_gpu_mlp.py)
    assert _rel(g0, r0) <= 5e-3 and _rel(gW, rW) <= 5e-3
    z2, _, _ = nat.forward(z0, None, ts)
    h0, _, hW, _ = nat.adjoint(z2, None, ts, dz)
    assert np.array_equal(z, z2) and np.array_equal(g0, h0) and np.array_equal(gW, hW)


def test_coupled_solve_while_another_stream_keeps_the_gpu_busy(o32):
    """The grid-wide sum of the coupled adaptive solve needs every workgroup resident: with a second stream saturating the
    CUs the cooperative launch must still give the same result as on an idle device (no timeout, no NaN blocks)."""
    import torch
    layers = (32, 128, 128, 32)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=O.BATCH_COUPLED)
    nat, _ = _native(W, **kw)
    B, T = 2048, 50
    z0, ts = _z0(B, 32), O.time_grid(T)
    dz = O.cotangent(T, B, 32)
    z_idle, ret, _ = nat.forward(z0, None, ts)
    g_idle = nat.adjoint(z_idle, None, ts, dz)
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(40):
            a = torch.tanh(a @ a * 1e-4)      # ≈ 1.1 TFLOP each: the device stays full for the duration of the solves
    z_busy, ret2, st = nat.forward(z0, None, ts)
    g_busy = nat.adjoint(z_busy, None, ts, dz)
    torch.cuda.synchronize()
    assert (ret == 0).all() and (ret2 == 0).all() and st["nfailed"] == 0
    assert np.array_equal(z_idle, z_busy) and np.array_equal(g_idle[0], g_busy[0]) and np.array_equal(g_idle[2], g_busy[2])


def test_reference_default_latentode_b64(o32, o64):
    """The reference's OWN LatentODE example [REF examples/pendulum_friction-less/model_train_LatentODE.jl:37, :42], [REF nODE.jl:11-16]:
    NODE(16) = 16-200-200-16 relu, Tsit5 at the default tolerances, ONE coupled solve on the [16 × 64] state, T = 50 — forward and
    adjoint at the full shape against the f32 oracle and float64 (gates: c4's, the other relu network under the adaptive coupled
    controller), plus bitwise determinism."""
    layers = (16, 200, 200, 16)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=16, param_dim=0, layers=layers, batching=O.BATCH_COUPLED)
    nat, od = _native(W, **kw)
    B, T = 64, 50
    z0, ts = _z0(B, 16), O.time_grid(T)
    dz = O.cotangent(T, B, 16)
    z, ret, st = nat.forward(z0, None, ts)
    g0, _, gW, sb = nat.adjoint(z, None, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0 and sb["nfailed"] == 0
    zr, _, info = o32.forward(od, z0, None, ts, W=W, nthreads=NT)
    r0, _, rW, infob = o32.adjoint(od, z, None, ts, dz, W=W, nthreads=NT)
    assert abs(st["naccept"] - info["naccept"]) <= 0.1 * info["naccept"] + 2
    # (the reverse solve of THIS problem sits at a cliff of the controller: 139, 122 and 142 accepted steps from three summation orders of
    #  the same kernel during round 4, 127–131 in the f32 oracle — the gate is for a controller that is wrong, not for round-off)
    assert abs(sb["naccept"] - infob["naccept"]) <= 0.15 * infob["naccept"] + 2
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    W64 = W.astype(np.float64)
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W64, nthreads=NT)
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W64, nthreads=NT)
    scale = max(1.0, np.abs(zr).max())
    e_k, e_o = np.abs(z - z64).max(), np.abs(zr - z64).max()
    assert np.abs(z - zr).max() <= 3e-4 * scale
    assert e_k <= 1.5 * e_o + 1e-5 * scale, (e_k, e_o)
    assert _rel(g0, r0) <= 5e-3 and _rel(gW, rW) <= 5e-3, (_rel(g0, r0), _rel(gW, rW))
    for g, r, t, what in ((g0, r0, t0, "dz0"), (gW, rW, tW, "dW")):
        assert _rel(g, t) <= 1.5 * _rel(r, t) + 5e-3, (what, _rel(g, t), _rel(r, t))
    z2, _, _ = nat.forward(z0, None, ts)
    h0, _, hW, _ = nat.adjoint(z2, None, ts, dz)
    assert np.array_equal(z, z2) and np.array_equal(g0, h0) and np.array_equal(gW, hW)


def test_c3_tanh_forced_rejections(o32, o64):
    """k_mlp64_adj folds gW₂ over an ACCEPTED step's six (h₁, δ₂) pairs and drops the pending sums of a rejected attempt; c3's smooth tanh
    twin never rejects under the default controller (nreject = 0 in every bench line), and with 49 save-time stops the backward steps are
    clipped before the controller can overshoot. Here it is pushed into rejections in BOTH directions: eight save times (intervals of
    0.35), an integral controller without safety factor (gamma = 1, beta1 = 0.2, beta2 = 0, qmax = 50) — the oracle rejects ≈ 6.6
    attempts per trajectory backward (measured: 6768 of 26836 attempts at B = 1024). The f32 oracle itself sits 1.5e-5 (dẑ₀) / 9e-6 (dW)
    from float64 under this controller; the gates are 5e-5: a fold that kept one rejected attempt's pairs, or lost one accepted step's,
    is ≈ 1/20 of the gradient — three orders above."""
    layers = (2, 64, 64, 2)
    kw = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers, activation=O.ACT_TANH)
    ctl = dict(abstol=1e-6, reltol=1e-6, gamma=1.0, beta1=0.2, beta2=0.0, qmax=50.0)
    W = O.mlp_weights(layers, seed=3)
    nat, od = _native(W, **kw, **ctl)
    B, T = 1024, 8
    z0, L = O.pendulum_inputs(B)
    ts = np.linspace(0.0, 2.45, T)
    dz = O.cotangent(T, B, 2)
    z, ret, st = nat.forward(z0, L, ts)
    g0, gL, gW, sb = nat.adjoint(z, L, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0 and sb["nfailed"] == 0
    assert sb["nreject"] > 2 * B and st["nreject"] > 2 * B, (st["nreject"], sb["nreject"])   # the path under test ran, in bulk
    r0, rL, rW, infob = o32.adjoint(od, z, L, ts, dz, W=W, nthreads=NT)
    assert infob["nreject"] > 2 * B
    d64 = O.make_desc(**kw, abstol=1e-11, reltol=1e-11)
    W64 = W.astype(np.float64)
    z64, _, _ = o64.forward(d64, z0, L, ts, W=W64, nthreads=NT)
    t0, tL, tW, _ = o64.adjoint(d64, z64, L, ts, dz, W=W64, nthreads=NT)
    assert np.abs(z - z64).max() <= 5e-5 * max(1.0, np.abs(z64).max())
    for g, r, t, what in ((g0, r0, t0, "dz0"), (gL, rL, tL, "dL"), (gW, rW, tW, "dW")):
        assert _rel(g, r) <= 5e-5, (what, _rel(g, r))
        assert _rel(g, t) <= 5e-5, (what, _rel(g, t))


TIGHT = {
    # name: (desc kwargs, B, D, pendulum inputs?, gates (z vs oracle, z vs f64, grads vs oracle, grads vs f64) in units of the scale)
    # Measured (abl/c3_tight.py, MI355X): the tanh networks — smooth right-hand sides, the solver's order holds — agree to 3e-6 in ẑ and
    # 5e-7 in every gradient between kernel, f32 oracle and float64: the gates below are 1e-5, so a 1 % gradient bug cannot pass. The relu
    # networks of the configs do not converge with the tolerance: every kink crossing costs O(h) locally whatever reltol says, and BOTH f32
    # solves sit 1.5e-4 (c3) / 1.6e-5 (c4) from float64 with gradients 2–4e-3 / 1.2e-3 off; the gates are those levels, plus "no farther
    # from float64 than 1.5× the oracle".
    "c3_relu": (dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 64, 64, 2)), 1024, 2, True, (5e-4, 5e-4, 1e-2, 1e-2)),
    "c3_tanh": (dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 64, 64, 2), activation=O.ACT_TANH), 1024, 2, True, (2e-5, 2e-5, 1e-5, 1e-5)),
    "c4_relu": (dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=(32, 128, 128, 32), batching=O.BATCH_COUPLED), 512, 32, False,
                (1e-5, 1e-4, 5e-3, 5e-3)),
    "c4_tanh": (dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=(32, 128, 128, 32), batching=O.BATCH_COUPLED,
                     activation=O.ACT_TANH), 512, 32, False, (1e-5, 1e-5, 1e-5, 1e-5)),
    # the reference's default NODE shape [REF nODE.jl:11-16] at its example's batch [REF model_train_LatentODE.jl:42]: relu as in the
    # reference (the gates of c4_relu), and the tanh twin that holds the code path to 1e-5
    "ref_relu": (dict(rhs_kind=O.RHS_MLP, state_dim=16, param_dim=0, layers=(16, 200, 200, 16), batching=O.BATCH_COUPLED), 64, 16, False,
                 (5e-5, 1e-4, 5e-3, 5e-3)),   # (measured, round 4: kernel and f32 oracle 1.3e-5·scale apart on this relu network at 1e-6)
    "ref_tanh": (dict(rhs_kind=O.RHS_MLP, state_dim=16, param_dim=0, layers=(16, 200, 200, 16), batching=O.BATCH_COUPLED,
                      activation=O.ACT_TANH), 64, 16, False, (1e-5, 1e-5, 1e-5, 1e-5)),
}


@pytest.mark.parametrize("case", sorted(TIGHT))
def test_full_batch_tight_tolerance(case, o32, o64):
    """c3 (B = 1024, per-trajectory control) and c4 (B = 512, one coupled solve) at abstol = reltol = 1e-6, every column, forward and
    adjoint, against the f32 oracle and against float64 — with the configs' relu networks and with the same shapes on tanh."""
    kw, B, D, pend, (gz, gz64, gg, gg64) = TIGHT[case]
    kw = {**kw, "abstol": 1e-6, "reltol": 1e-6}
    W = O.mlp_weights(kw["layers"], seed=3)
    nat, od = _native(W, **kw)
    T = 50
    if pend:
        z0, L = O.pendulum_inputs(B)
    else:
        z0, L = _z0(B, D), None
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, D)
    z, ret, st = nat.forward(z0, L, ts)
    g0, gL, gW, sb = nat.adjoint(z, L, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0 and sb["nfailed"] == 0
    zr, _, _ = o32.forward(od, z0, L, ts, W=W, nthreads=NT)
    r0, rL, rW, _ = o32.adjoint(od, z, L, ts, dz, W=W, nthreads=NT)
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    W64 = W.astype(np.float64)
    z64, _, _ = o64.forward(d64, z0, L, ts, W=W64, nthreads=NT)
    t0, tL, tW, _ = o64.adjoint(d64, z64, L, ts, dz, W=W64, nthreads=NT)
    scale = max(1.0, np.abs(zr).max())
    e_k, e_o = np.abs(z - z64).max(), np.abs(zr - z64).max()
    assert np.abs(z - zr).max() <= gz * scale, np.abs(z - zr).max()
    assert e_k <= gz64 * scale and e_k <= 1.5 * e_o + 1e-5 * scale, (e_k, e_o)
    pairs = [(g0, r0, t0, "dz0"), (gW, rW, tW, "dW")] + ([(gL, rL, tL, "dL")] if pend else [])
    for g, r, t, what in pairs:
        assert _rel(g, r) <= gg, (what, _rel(g, r))
        assert _rel(g, t) <= gg64 and _rel(g, t) <= 1.5 * _rel(r, t) + 0.2 * gg64, (what, _rel(g, t), _rel(r, t))
