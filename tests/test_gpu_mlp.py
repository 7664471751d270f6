"""GPU parity of the MLP right-hand sides (LatentODE path, physics+MLP) against the CPU oracle.

Tolerances: fixed-step RK4 has no controller ⇒ fp32 round-off only: |Δẑ| ≤ 2e-5·max(1,|ẑ|). Adaptive Tsit5 at the
OrdinaryDiffEq default tolerance: see tests/test_gpu_pendulum.py (≤ 3e-4 from the oracle, no farther from the float64
truth than 1.5× the oracle + 1e-5); at 1e-6/1e-6 ≤ 2e-5."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _native(W, options=(), **kw):
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    d = make_desc(**kw)
    nat = Native(d)
    for k_, v_ in dict(options).items():        # kernel-choice knobs (lde_set_option): which family serves the call
        nat.set_option(k_, v_)
    if W is not None:
        nat.set_weights(W)
    return nat, copy_desc_to_oracle(d)


def _z0(B, D, seed=1):
    return (0.5 * np.random.default_rng(seed).standard_normal((B, D))).astype(np.float32)


def _check_forward(z, zr, zt, tight):
    scale = max(1.0, np.abs(zr).max())
    if tight and zt is None:
        assert np.abs(z - zr).max() <= 2e-5 * scale
    elif tight:  # adaptive at 1e-6 on a relu network: kinks make the step sequence round-off sensitive
        assert np.abs(z - zr).max() <= 1e-4 * scale
        assert np.abs(z - zt).max() <= 1.5 * np.abs(zr - zt).max() + 2e-5 * scale
    else:  # two correct fp32 solves differ by at most about the solver's own error at this tolerance
        e_o = np.abs(zr - zt).max()
        assert np.abs(z - zr).max() <= max(3e-4 * scale, 1.5 * e_o)
        assert np.abs(z - zt).max() <= 1.5 * e_o + 1e-5 * scale


@pytest.mark.parametrize("B", [16, 40, 256])
def test_c2_rk4_fixed_step_coupled(o32, o64, B):
    """BASELINE config 2: D=8, 8→200→200→8 relu, RK4 dt=0.05, coupled batch (several workgroups, ragged last tile)."""
    layers = (8, 200, 200, 8)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=layers, solver=O.SOLVER_RK4, adaptive=0, dt=0.05,
              batching=O.BATCH_COUPLED)
    nat, od = _native(W, **kw)
    z0, ts = _z0(B, 8), O.time_grid(50)
    z, ret, st = nat.forward(z0, None, ts)
    zr, _, info = o32.forward(od, z0, None, ts, W=W)
    assert (ret == 0).all() and np.array_equal(z[0], z0)
    assert st["naccept"] == info["naccept"] == 49 and st["nfe"] == info["nfe"] == 197
    _check_forward(z, zr, None, True)


@pytest.mark.parametrize("batching", [O.BATCH_PER_TRAJECTORY, O.BATCH_COUPLED])
@pytest.mark.parametrize("tol", [(1e-6, 1e-3), (1e-6, 1e-6)])
@pytest.mark.parametrize("B", [16, 72])
def test_tsit5_mlp_forward(o32, o64, batching, tol, B):
    """Adaptive Tsit5 on a 32→128→128→32 relu MLP (config-4 shape); B=72 ⇒ 5 workgroups, coupled mode then needs the
    grid-wide error norm."""
    layers = (32, 128, 128, 32)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=batching, abstol=tol[0], reltol=tol[1])
    nat, od = _native(W, **kw)
    z0, ts = _z0(B, 32), O.time_grid(50)
    z, ret, st = nat.forward(z0, None, ts)
    zr, _, info = o32.forward(od, z0, None, ts, W=W)
    assert (ret == 0).all()
    dt_ = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=batching, abstol=1e-10, reltol=1e-10)
    zt, _, _ = o64.forward(dt_, z0, None, ts, W=W.astype(np.float64))
    _check_forward(z, zr, zt, tol[1] < 1e-4)
    assert abs(st["naccept"] - info["naccept"]) <= 0.05 * info["naccept"] + 1
    if batching == O.BATCH_COUPLED:
        assert st["nfe"] == 6 * (st["naccept"] + st["nreject"]) + 2


def test_c3_pendulum_plus_mlp_tight_tanh(o32, o64):
    """Smooth variant at 1e-7: the physics+MLP right-hand side and its adjoint to 2e-5 / 1e-4 of the float64 answer."""
    layers = (2, 64, 64, 2)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7)
    nat, od = _native(W, **kw)
    B, T = 50, 50
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    z, ret, _ = nat.forward(z0, L, ts)
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    zt, _, _ = o64.forward(d64, z0, L, ts, W=W.astype(np.float64))
    assert (ret == 0).all() and np.abs(z - zt).max() <= 2e-5
    g0, gL, gW, _ = nat.adjoint(z, L, ts, dz)
    t0, tL, tW, _ = o64.adjoint(d64, zt, L, ts, dz, W=W.astype(np.float64))
    for g, t in ((g0, t0), (gL, tL), (gW, tW)):
        assert np.abs(g - t).max() <= 1e-4 * np.abs(t).max()


def test_c3_pendulum_plus_mlp_forward(o32, o64):
    layers = (2, 64, 64, 2)
    W = O.mlp_weights(layers, seed=3)
    nat, od = _native(W, rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
    B = 100
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(50)
    z, ret, st = nat.forward(z0, L, ts)
    zr, _, info = o32.forward(od, z0, L, ts, W=W)
    assert (ret == 0).all()
    zt, _, _ = o64.forward(O.make_desc(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers, abstol=1e-10, reltol=1e-10), z0, L, ts,
                           W=W.astype(np.float64))
    _check_forward(z, zr, zt, False)
    assert abs(st["naccept"] - info["naccept"]) <= 0.03 * info["naccept"] + 1


def test_augmented_tanh_and_odd_sizes(o32):
    """AugmentedNDELayer: extra zero rows [REF LatentODE.jl:71]; widths that are not multiples of 16/4; 4 Dense layers."""
    layers = (7, 33, 50, 21, 7)
    W = O.mlp_weights(layers, seed=4)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=5, param_dim=0, augment_dim=2, layers=layers, activation=O.ACT_TANH,
              abstol=1e-6, reltol=1e-6)
    nat, od = _native(W, **kw)
    z0, ts = _z0(21, 5), O.time_grid(20)
    z, ret, _ = nat.forward(z0, None, ts)
    zr, _, _ = o32.forward(od, z0, None, ts, W=W)
    assert z.shape == (20, 21, 7) and np.array_equal(z[0, :, :5], z0) and (z[0, :, 5:] == 0).all()
    _check_forward(z, zr, None, True)


def test_linear_rhs_vs_expm():
    """Independent of the oracle: one Dense layer ⇒ ż = Az + b, compared with scipy.linalg.expm."""
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    D, B, T = 4, 19, 20
    A = rng.standard_normal((D, D)) * 0.5
    b = rng.standard_normal(D) * 0.1
    W = np.concatenate([A.flatten(order="F"), b]).astype(np.float32)
    nat, _ = _native(W, rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=(D, D), abstol=1e-7, reltol=1e-7)
    z0 = rng.standard_normal((B, D)).astype(np.float32)
    ts = O.time_grid(T, 0.1)
    z, ret, _ = nat.forward(z0, None, ts)
    M = np.zeros((D + 1, D + 1))
    M[:D, :D], M[:D, D] = A.astype(np.float32), b.astype(np.float32)
    for j, t in enumerate(ts):
        E = expm(M * t)
        assert np.abs((E[:D, :D] @ z0.T.astype(np.float64)).T + E[:D, D] - z[j]).max() <= 2e-5


# ------------------------------------------------------------------------------------------------ adjoint
def _check_grads(g, r, t, lim_o, lim_t, what):
    s = np.abs(t).max()
    assert np.abs(g - r).max() <= lim_o * s, what
    assert np.abs(g - t).max() <= 1.5 * np.abs(r - t).max() + lim_t * s, what


@pytest.mark.parametrize("B", [16, 40])
def test_c2_rk4_adjoint(o32, o64, B):
    """Config 2 backward: fixed-step RK4 continuous adjoint, dW accumulated over all 49·4 stage evaluations."""
    layers = (8, 200, 200, 8)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=layers, solver=O.SOLVER_RK4, adaptive=0, dt=0.05,
              batching=O.BATCH_COUPLED)
    nat, od = _native(W, **kw)
    z0, ts = _z0(B, 8), O.time_grid(50)
    dz = O.cotangent(50, B, 8)
    z, _, _ = nat.forward(z0, None, ts)
    g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    r0, _, rW, info = o32.adjoint(od, z, None, ts, dz, W=W)
    assert st["naccept"] == info["naccept"] == 49
    d64 = O.make_desc(**{**kw, "adaptive": False, "dt": 0.05 / 8})
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W.astype(np.float64))
    _check_grads(g0, r0, t0, 2e-4, 1e-3, "dz0")
    _check_grads(gW, rW, tW, 2e-4, 1e-3, "dW")


@pytest.mark.parametrize("batching", [O.BATCH_PER_TRAJECTORY, O.BATCH_COUPLED])
@pytest.mark.parametrize("B", [16, 72])
def test_tsit5_mlp_adjoint(o32, o64, batching, B):
    layers = (32, 128, 128, 32)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, layers=layers, batching=batching)
    nat, od = _native(W, **kw)
    z0, ts = _z0(B, 32), O.time_grid(50)
    dz = O.cotangent(50, B, 32)
    z, _, _ = nat.forward(z0, None, ts)
    g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    r0, _, rW, info = o32.adjoint(od, z, None, ts, dz, W=W)
    assert st["nfailed"] == 0
    # relu network + reltol 1e-3: the step sequence of two correct f32 solves differs by round-off-triggered accept/reject flips
    assert abs(st["naccept"] - info["naccept"]) <= 0.15 * info["naccept"] + 2
    d64 = O.make_desc(**{**kw, "abstol": 1e-10, "reltol": 1e-10})
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W.astype(np.float64))
    _check_grads(g0, r0, t0, 5e-3, 5e-3, "dz0")
    _check_grads(gW, rW, tW, 5e-3, 5e-3, "dW")


def test_c3_pendulum_plus_mlp_adjoint(o32, o64):
    layers = (2, 64, 64, 2)
    W = O.mlp_weights(layers, seed=3)
    kw = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
    nat, od = _native(W, **kw)
    B = 100
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(50)
    dz = O.cotangent(50, B, 2)
    z, _, _ = nat.forward(z0, L, ts)
    g0, gL, gW, st = nat.adjoint(z, L, ts, dz)
    r0, rL, rW, info = o32.adjoint(od, z, L, ts, dz, W=W)
    assert st["nfailed"] == 0 and abs(st["naccept"] - info["naccept"]) <= 0.1 * info["naccept"] + 2
    d64 = O.make_desc(**{**kw, "abstol": 1e-10, "reltol": 1e-10})
    z64, _, _ = o64.forward(d64, z0, L, ts, W=W.astype(np.float64))
    t0, tL, tW, _ = o64.adjoint(d64, z64, L, ts, dz, W=W.astype(np.float64))
    # default tolerance + relu: gradients of two correct fp32 solves agree to about 1 %
    _check_grads(g0, r0, t0, 1e-2, 5e-3, "dz0")
    _check_grads(gL, rL, tL, 1e-2, 5e-3, "dL")
    _check_grads(gW, rW, tW, 1e-2, 5e-3, "dW")


def test_augmented_tanh_adjoint_tight(o32, o64):
    """Smooth activation at tight tolerance: gradients must agree with the float64 adjoint to 1e-4."""
    layers = (7, 33, 50, 21, 7)
    W = O.mlp_weights(layers, seed=4)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=5, param_dim=0, augment_dim=2, layers=layers, activation=O.ACT_TANH,
              abstol=1e-7, reltol=1e-7)
    nat, od = _native(W, **kw)
    B, T = 21, 20
    z0, ts = _z0(B, 5), O.time_grid(T)
    dz = O.cotangent(T, B, 7)
    z, _, _ = nat.forward(z0, None, ts)
    g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W.astype(np.float64))
    assert g0.shape == (B, 5)
    assert np.abs(g0 - t0).max() <= 1e-4 * np.abs(t0).max()
    assert np.abs(gW - tW).max() <= 1e-4 * np.abs(tW).max()


def test_per_trajectory_rejections_partial_acceptance(o32, o64):
    """A huge user-supplied initial dt forces rejected steps in some columns of a tile while others accept:
    the weight gradient must still equal the oracle's (accepted steps only)."""
    layers = (4, 32, 32, 4)
    W = O.mlp_weights(layers, seed=6, scale=2.0)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=4, param_dim=0, layers=layers, activation=O.ACT_TANH, abstol=1e-6, reltol=1e-6,
              dt=0.5)
    nat, od = _native(W, **kw)
    B, T = 37, 8
    z0 = (np.random.default_rng(3).standard_normal((B, 4)) * np.linspace(0.05, 2.0, B)[:, None]).astype(np.float32)
    ts = O.time_grid(T, 0.5)
    dz = O.cotangent(T, B, 4)
    z, _, fs = nat.forward(z0, None, ts)
    g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    assert st["nreject"] > 0 and fs["nreject"] > 0
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11, "dt": 0.0})
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W.astype(np.float64))
    assert np.abs(z - z64).max() <= 2e-5 * max(1, np.abs(z64).max())
    assert np.abs(g0 - t0).max() <= 2e-4 * np.abs(t0).max()
    assert np.abs(gW - tW).max() <= 2e-4 * np.abs(tW).max()


@pytest.mark.parametrize("case", ["rk4_coupled", "tsit5_per_traj", "c3"])
def test_staging_overflow_path_gives_the_same_gradient(case):
    """The adjoint stages (a_l, δ_l) panels in HBM and a second kernel forms dW. When a workgroup runs out of staging
    slots it folds them into a private slab inside the solve kernel. Forcing a tiny staging area (2 step attempts) must
    reproduce the gradient of the roomy run: dz0 bit-for-bit (the solve is untouched), dW up to summation order."""
    if case == "rk4_coupled":
        layers = (8, 200, 200, 8)
        kw = dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=layers, solver=O.SOLVER_RK4, adaptive=0, dt=0.05,
                  batching=O.BATCH_COUPLED)
        B, D, T, nst = 40, 8, 20, 4
    elif case == "tsit5_per_traj":
        layers = (6, 40, 70, 6)
        kw = dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, layers=layers, activation=O.ACT_TANH, abstol=1e-6,
                  reltol=1e-5, dt=0.5)   # dt=0.5: rejected first attempts, per-column accept/reject inside a tile
        B, D, T, nst = 53, 6, 12, 6
    else:
        layers = (2, 64, 64, 2)
        kw = dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
        B, D, T, nst = 48, 2, 15, 6
    W = O.mlp_weights(layers, seed=5)
    ts = O.time_grid(T)
    if case == "c3":
        z0, L = O.pendulum_inputs(B)
    else:
        z0, L = _z0(B, D, seed=2), None
    dz = O.cotangent(T, B, D)
    res = []
    for slots in (None, 2 * nst):
        nat, _ = _native(W, options={} if slots is None else {"mlp_stage_slots": slots}, **kw)
        z, _, _ = nat.forward(z0, L, ts)
        g0, gL, gW, st = nat.adjoint(z, L, ts, dz)
        assert st["nfailed"] == 0 and st["naccept"] >= T - 1
        res.append((g0, gL, gW, st))
    (a0, aL, aW, ast), (b0, bL, bW, bst) = res
    # The roomy run uses the small-batch kernel (one trajectory per workgroup) and the tiny one its 16-column fallback: the same
    # solve in another summation order. Fixed step: round-off; smooth right-hand side at 1e-6: agreement to round-off; relu at
    # the default tolerance: two correct fp32 solves agree to about 1 % (see test_c3_pendulum_plus_mlp_adjoint).
    lim = {"rk4_coupled": 5e-6, "tsit5_per_traj": 2e-5, "c3": 1e-2}[case]
    assert abs(ast["naccept"] - bst["naccept"]) <= (0.1 if case == "c3" else 0.02) * ast["naccept"] + 1
    assert np.abs(a0 - b0).max() <= lim * np.abs(a0).max()
    if aL is not None:
        assert np.abs(aL - bL).max() <= lim * np.abs(aL).max()
    assert np.isfinite(aW).all() and np.abs(aW).max() > 0
    assert np.abs(aW - bW).max() <= max(lim, 5e-5) * np.abs(aW).max()


@pytest.mark.parametrize("case", ["c3", "tanh_per_traj", "tanh_coupled", "rk4_fixed", "wide_128", "aug_odd"])
def test_four_column_kernel_matches_sixteen_column_kernel(case, o64):
    """Networks ≤ 64 wide run the adjoint with four trajectories per wave (lde_mlp4.h); option "mlp4" = 0 forces the 16-column
    workgroup kernel. Same algorithm, other summation order: at tight tolerance both must sit within 1e-4 of the float64
    adjoint and within 5e-5 of each other (option "mlp4_maxw" = 256 lets the 128-wide case through the four-column kernel)."""
    tight = dict(abstol=1e-7, reltol=1e-7)
    opts4 = {}
    if case == "c3":
        layers, kw, B, D, T = (2, 64, 64, 2), dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, activation=O.ACT_TANH, **tight), 70, 2, 12
    elif case == "tanh_per_traj":
        layers, kw, B, D, T = (6, 40, 33, 6), dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, activation=O.ACT_TANH, **tight), 37, 6, 10
    elif case == "tanh_coupled":
        layers, kw, B, D, T = (6, 40, 33, 6), dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, activation=O.ACT_TANH,
                                                   batching=O.BATCH_COUPLED, **tight), 37, 6, 10
    elif case == "rk4_fixed":
        layers, kw, B, D, T = (8, 64, 64, 8), dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, solver=O.SOLVER_RK4, adaptive=0,
                                                   dt=0.0125, batching=O.BATCH_COUPLED), 19, 8, 12
    elif case == "wide_128":
        layers, kw, B, D, T = (32, 128, 128, 32), dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, activation=O.ACT_TANH,
                                                       batching=O.BATCH_COUPLED, **tight), 24, 32, 8
        opts4 = {"mlp4_maxw": 256}
    else:
        layers, kw, B, D, T = (7, 33, 50, 21, 7), dict(rhs_kind=O.RHS_MLP, state_dim=5, param_dim=0, augment_dim=2,
                                                       activation=O.ACT_TANH, **tight), 21, 5, 9
    kw["layers"] = layers
    W = O.mlp_weights(layers, seed=4)
    ts = O.time_grid(T)
    if case == "c3":
        z0, L = O.pendulum_inputs(B)
    else:
        z0, L = _z0(B, D, seed=3), None
    Dp = D + kw.get("augment_dim", 0)
    dz = O.cotangent(T, B, Dp)
    out = {}
    for flag in ("1", "0"):
        nat, od = _native(W, options={**opts4, "mlp4": int(flag)}, **kw)
        z, _, _ = nat.forward(z0, L, ts)
        out[flag] = nat.adjoint(z, L, ts, dz)
    d64 = O.make_desc(**{**kw, **({"abstol": 1e-11, "reltol": 1e-11} if kw.get("adaptive", 1) else {"dt": kw["dt"] / 8, "adaptive": False})})
    z64, _, _ = o64.forward(d64, z0, L, ts, W=W.astype(np.float64))
    t0, tL, tW, _ = o64.adjoint(d64, z64, L, ts, dz, W=W.astype(np.float64))
    (a0, aL, aW, ast), (b0, bL, bW, bst) = out["1"], out["0"]
    assert ast["nfailed"] == 0 and bst["nfailed"] == 0
    assert abs(ast["naccept"] - bst["naccept"]) <= 0.15 * bst["naccept"] + 1   # fp32 noise in the error estimates of the first tiny steps
    for g4, g16, t, what in ((a0, b0, t0, "dz0"), (aL, bL, tL, "dL"), (aW, bW, tW, "dW")):
        if g4 is None:
            continue
        s = np.abs(t).max()
        assert np.abs(g4 - g16).max() <= 5e-5 * s, what
        assert np.abs(g4 - t).max() <= (1e-3 if case == "rk4_fixed" else 1e-4) * s, what   # rk4: the h⁴ error of dt = 0.0125 itself


def test_torch_api_latentode_trains_the_node_weights(o32, o64):
    """diffeq_layer(::Decoder{LatentODE}, ẑ₀, t) through the reference-shaped host API: ẑ [D', B, T], and — unlike the
    reference, where `dudt` is invisible to Flux.params — the adjoint's dW reaches the torch parameters
    of `dudt` (through the differentiable destructure-order flattening)."""
    import torch
    import latentdiffeq_amd as la
    torch.manual_seed(0)
    D, aug, H, B, T = 6, 2, 32, 24, 20
    node = la.NODE(D, hidden_dim=H, augment_dim=aug, device="cuda", abstol=1e-7, reltol=1e-7, activation="tanh")
    dec = la.Decoder(la.LatentODE(), (None, node, None))
    z0 = _z0(B, D)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, D + aug)
    z0t = torch.tensor(z0.T.copy(), device="cuda", requires_grad=True)           # [D, B]
    zhat = la.diffeq_layer(dec, z0t, ts)
    assert tuple(zhat.shape) == (D + aug, B, T) and zhat.is_cuda
    (zhat * torch.tensor(dz, device="cuda").permute(2, 1, 0)).sum().backward()
    W = node.flat_weights().detach().cpu().numpy()
    od = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, augment_dim=aug, layers=tuple(node.layer_sizes),
                     batching=O.BATCH_COUPLED, abstol=1e-10, reltol=1e-10, activation=O.ACT_TANH)
    zt, _, _ = o64.forward(od, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(od, zt, None, ts, dz, W=W.astype(np.float64))
    assert np.abs(zhat.detach().permute(2, 1, 0).cpu().numpy() - zt).max() <= 2e-5
    assert np.abs(z0t.grad.cpu().numpy().T - t0).max() <= 2e-4 * np.abs(t0).max()
    # gradients arrive on the torch parameters in THEIR layout (weight [out, in] row-major, bias)
    off = 0
    for m in node.dudt:
        if isinstance(m, torch.nn.Linear):
            o_, i_ = m.weight.shape
            gw = tW[off:off + o_ * i_].reshape(i_, o_).T      # vec(W) column-major [out×in]
            off += o_ * i_
            gb = tW[off:off + o_]
            off += o_
            assert np.abs(m.weight.grad.cpu().numpy() - gw).max() <= 3e-4 * np.abs(tW).max()
            assert np.abs(m.bias.grad.cpu().numpy() - gb).max() <= 3e-4 * np.abs(tW).max()
    # single process: the gradient all-reduce is a no-op and must not need an initialised process group
    la.FlatGradAllReduce(node.dudt.parameters())()


def test_hook_and_sharded_entry_point():
    """transform_after_diffeq is applied where the reference applies it; diffeq_layer_sharded(world=1) == diffeq_layer."""
    import torch
    import latentdiffeq_amd as la

    class Doubling(la.Pendulum):
        def transform_after_diffeq(self, x):     # sees [D, T, B] for GOKU  [REF GOKU.jl:124]
            assert tuple(x.shape) == (2, 10, 8)
            return 2 * x
    z0, L = O.pendulum_inputs(8)
    ts = O.time_grid(10)
    a = torch.tensor(z0.T.copy(), device="cuda"), torch.tensor(L.T.copy(), device="cuda")
    plain = la.diffeq_layer(la.Decoder(la.GOKU_basic(), (None, la.Pendulum(), None)), a, ts)
    hooked = la.diffeq_layer(la.Decoder(la.GOKU_basic(), (None, Doubling(), None)), a, ts)
    assert tuple(hooked.shape) == (2, 8, 10) and torch.equal(hooked, 2 * plain)
    shard = la.diffeq_layer_sharded(la.Decoder(la.GOKU_basic(), (None, la.Pendulum(), None)), a, ts, rank=1, world=2)
    assert torch.equal(shard, plain[:, 4:, :])


@pytest.mark.parametrize("seed", range(16))
def test_random_mlp_shapes_forward_and_adjoint(o64, seed):
    """Shape fuzzing of the padding / tiling logic: random depth (1–6 Dense layers), widths that are not multiples of
    4/16/32 (including 1 and >128), random state/augment dims, both batching modes and both solvers, ragged batches.
    Smooth activation + tight tolerance so that the float64 oracle is a sharp reference (1e-4 relative)."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 41))
    aug = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
    Dp = D + aug
    nl = int(rng.integers(1, 7))
    widths = [int(rng.choice([1, 3, 5, 16, 17, 31, 32, 33, 50, 64, 65, 100, 129, 200])) for _ in range(nl - 1)]
    layers = (Dp, *widths, Dp)
    batching = int(rng.integers(0, 2))
    rk4 = rng.random() < 0.3
    B = int(rng.choice([1, 5, 16, 17, 40]))
    T = int(rng.integers(2, 12))
    W = O.mlp_weights(layers, seed=seed, scale=0.7)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, augment_dim=aug, layers=layers, activation=O.ACT_TANH,
              batching=batching)
    kw.update(dict(solver=O.SOLVER_RK4, adaptive=0, dt=0.025) if rk4 else dict(abstol=1e-7, reltol=1e-7))
    nat, od = _native(W, **kw)
    z0 = _z0(B, D, seed=seed)
    ts = O.time_grid(T, 0.1)
    dz = O.cotangent(T, B, Dp, seed=seed)
    from latentdiffeq_amd._lib import LdeError
    z, ret, _ = nat.forward(z0, None, ts)
    try:
        g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    except LdeError as e:   # the only acceptable refusal: a deep AND wide net whose tile state exceeds the 160 KiB LDS
        assert "does not fit the 160 KiB LDS" in str(e) and sum(widths) >= 500, (layers, str(e))
        return
    kw64 = dict(kw)
    kw64.update(dict(dt=0.025 / 4) if rk4 else dict(abstol=1e-11, reltol=1e-11))
    d64 = O.make_desc(**{k: (bool(v) if k == "adaptive" else v) for k, v in kw64.items()})
    zt, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, zt, None, ts, dz, W=W.astype(np.float64))
    assert (ret == 0).all() and st["nfailed"] == 0
    assert z.shape == (T, B, Dp)
    assert np.abs(z - zt).max() <= 1e-4 * max(1.0, np.abs(zt).max()), (layers, batching, rk4, B, T)
    assert np.abs(g0 - t0).max() <= 2e-4 * np.abs(t0).max() + 1e-9, (layers, batching, rk4, B, T)
    assert np.abs(gW - tW).max() <= 2e-4 * np.abs(tW).max() + 1e-9, (layers, batching, rk4, B, T)


@pytest.mark.parametrize("batching", [O.BATCH_PER_TRAJECTORY, O.BATCH_COUPLED])
def test_mlp_tsit5_with_fixed_step(o32, batching):
    layers = (6, 40, 40, 6)
    W = O.mlp_weights(layers, seed=9)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, layers=layers, batching=batching, adaptive=0, dt=0.025)
    nat, od = _native(W, **kw)
    B, T = 35, 20
    z0, ts = _z0(B, 6), O.time_grid(T)
    dz = O.cotangent(T, B, 6)
    z, ret, st = nat.forward(z0, None, ts)
    zr, _, info = o32.forward(od, z0, None, ts, W=W)
    assert (ret == 0).all() and st["naccept"] == info["naccept"]
    assert np.abs(z - zr).max() <= 2e-5 * max(1.0, np.abs(zr).max())
    g0, _, gW, _ = nat.adjoint(z, None, ts, dz)
    r0, _, rW, _ = o32.adjoint(od, z, None, ts, dz, W=W)
    assert np.abs(g0 - r0).max() <= 2e-4 * np.abs(r0).max() and np.abs(gW - rW).max() <= 2e-4 * np.abs(rW).max()


@pytest.mark.parametrize("layers,B", [((4, 48, 48, 4), 3001), ((8, 96, 96, 8), 1003)])
def test_mlp_adjoint_large_batches(o32, o64, layers, B):
    """Many tiles per CU, ragged last tile, both adjoint kernels (≤ 64 wide: four columns per wave; wider: 16-column
    workgroups), per-trajectory control at 1e-6: gradients against the float64 adjoint."""
    D = layers[0]
    W = O.mlp_weights(layers, seed=9)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=layers, activation=O.ACT_TANH, abstol=1e-6, reltol=1e-6)
    nat, od = _native(W, **kw)
    T = 8
    z0, ts = _z0(B, D, seed=6), O.time_grid(T)
    dz = O.cotangent(T, B, D)
    z, ret, _ = nat.forward(z0, None, ts)
    g0, _, gW, st = nat.adjoint(z, None, ts, dz)
    assert (ret == 0).all() and st["nfailed"] == 0
    d64 = O.make_desc(**{**kw, "abstol": 1e-11, "reltol": 1e-11})
    z64, _, _ = o64.forward(d64, z0, None, ts, W=W.astype(np.float64))
    t0, _, tW, _ = o64.adjoint(d64, z64, None, ts, dz, W=W.astype(np.float64))
    assert np.abs(z - z64).max() <= 2e-5 * max(1, np.abs(z64).max())
    assert np.abs(g0 - t0).max() <= 2e-4 * np.abs(t0).max()
    assert np.abs(gW - tW).max() <= 2e-4 * np.abs(tW).max()


@pytest.mark.parametrize("layers,batching,solver", [((8, 200, 200, 8), O.BATCH_COUPLED, O.SOLVER_RK4), ((2, 64, 64, 2), O.BATCH_PER_TRAJECTORY, O.SOLVER_TSIT5)],
                         ids=["c2", "c3-like"])
def test_one_mlp_handle_changing_shapes(o32, layers, batching, solver):
    """One handle with save grids and batches growing and shrinking: every workspace (panels, staging area, slabs) regrows
    cleanly, forward and adjoint match the oracle at each shape."""
    W = O.mlp_weights(layers, seed=3)
    D = layers[0]
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=layers, solver=solver, batching=batching, abstol=1e-6, reltol=1e-6,
              activation=O.ACT_TANH)        # smooth: the adaptive step sequence is then not round-off sensitive (no relu kinks)
    if solver == O.SOLVER_RK4:
        kw.update(adaptive=0, dt=0.05)
    nat, od = _native(W, **kw)
    for T, B in ((5, 16), (50, 64), (10, 200), (51, 33), (3, 257)):
        z0, ts = _z0(B, D, seed=T), O.time_grid(T)
        z, ret, _ = nat.forward(z0, None, ts)
        zr, _, _ = o32.forward(od, z0, None, ts, W=W)
        assert (ret == 0).all() and np.abs(z - zr).max() <= 1e-4 * max(1.0, np.abs(zr).max()), (T, B)
        dz = O.cotangent(T, B, D)
        g0, _, gW, _ = nat.adjoint(z, None, ts, dz)
        r0, _, rW, _ = o32.adjoint(od, z, None, ts, dz, W=W)
        e0, eW = np.abs(g0 - r0).max() / np.abs(r0).max(), np.abs(gW - rW).max() / np.abs(rW).max()
        assert e0 <= 2e-3 and eW <= 2e-3, (T, B, e0, eW)


@pytest.mark.parametrize("case", ["c2_rk4_coupled", "c3_per_traj", "c4_coupled", "tanh_per_traj_4d", "rk4_per_traj_small", "deep_4_layers",
                                  "tsit5_d12_h150_coupled", "rk4_d20_h96_per_traj", "tanh_d8_h70_aug", "tanh_d8_h70_backsolve",
                                  "relu_d32_h128_long_grid"])
def test_kernel_families_agree(case, o64):
    """Four kernel families serve the MLP right-hand sides: 16-column MFMA tiles (large batches; options "mlpv" = "mlp64" = "mlpw" = 0
    force them), one trajectory per workgroup with lanes = hidden units (k_mlpv; "mlp64" = "mlpw" = 0), one wave per trajectory
    with everything in registers (k_mlp64: three layers ≤ 64 wide, D' ≤ 4, per-trajectory control) and W waves per trajectory with
    weights and state in registers (k_mlpw: three layers ≤ 200 wide, D' ≤ 32). Same algorithm, same control arithmetic: they
    agree like two correct f32 solves — round-off for fixed steps and smooth networks at tight tolerance, the solver's own error
    where a relu network meets the adaptive controller — and every family is no farther from the float64 adjoint than that."""
    cfg = {
        "c2_rk4_coupled": dict(layers=(8, 200, 200, 8), B=48, kw=dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, solver=O.SOLVER_RK4, adaptive=0, dt=0.05, batching=O.BATCH_COUPLED), lim=5e-6),
        "c3_per_traj": dict(layers=(2, 64, 64, 2), B=80, kw=dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP), lim=1e-2),
        "c4_coupled": dict(layers=(32, 128, 128, 32), B=40, kw=dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, batching=O.BATCH_COUPLED), lim=5e-3),
        "tanh_per_traj_4d": dict(layers=(4, 48, 33, 4), B=37, kw=dict(rhs_kind=O.RHS_MLP, state_dim=4, param_dim=0, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7), lim=1e-4),
        "rk4_per_traj_small": dict(layers=(3, 20, 64, 3), B=19, kw=dict(rhs_kind=O.RHS_MLP, state_dim=3, param_dim=0, solver=O.SOLVER_RK4, adaptive=0, dt=0.025, activation=O.ACT_TANH), lim=5e-6),
        "deep_4_layers": dict(layers=(6, 40, 24, 40, 6), B=21, kw=dict(rhs_kind=O.RHS_MLP, state_dim=6, param_dim=0, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7), lim=1e-4),
        # the other instantiations of k_mlpw: (D' > 8, H > 128), (D' > 8, H ≤ 128) fixed step, (D' ≤ 8, H ≤ 128) with augmented rows
        "tsit5_d12_h150_coupled": dict(layers=(12, 150, 137, 12), B=24, kw=dict(rhs_kind=O.RHS_MLP, state_dim=12, param_dim=0, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7, batching=O.BATCH_COUPLED), lim=1e-4),
        "rk4_d20_h96_per_traj": dict(layers=(20, 96, 128, 20), B=33, kw=dict(rhs_kind=O.RHS_MLP, state_dim=20, param_dim=0, solver=O.SOLVER_RK4, adaptive=0, dt=0.025, activation=O.ACT_TANH), lim=5e-6),
        "tanh_d8_h70_aug": dict(layers=(8, 70, 65, 8), B=26, kw=dict(rhs_kind=O.RHS_MLP, state_dim=6, augment_dim=2, param_dim=0, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7), lim=1e-4),
        # the adjoint without restarts from the saved states (BacksolveAdjoint(checkpointing = false))
        "tanh_d8_h70_backsolve": dict(layers=(8, 70, 65, 8), B=26, kw=dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, activation=O.ACT_TANH, abstol=1e-7, reltol=1e-7, sensealg=O.SENSE_BACKSOLVE), lim=1e-4),
        # 220 save times × 32 states: the cotangents no longer fit k_mlpw's LDS copy (the jump reads them from HBM)
        "relu_d32_h128_long_grid": dict(layers=(32, 128, 128, 32), B=12, T=220, kw=dict(rhs_kind=O.RHS_MLP, state_dim=32, param_dim=0, abstol=1e-6, reltol=1e-6), lim=2e-3),
    }[case]
    layers, B, kw, lim = cfg["layers"], cfg["B"], dict(cfg["kw"], layers=cfg["layers"]), cfg["lim"]
    W = O.mlp_weights(layers, seed=8)
    D = kw.get("state_dim", 2)
    T = cfg.get("T", 20)
    ts = O.time_grid(T)
    if kw["rhs_kind"] == O.RHS_PENDULUM_PLUS_MLP:
        z0, L = O.pendulum_inputs(B)
    else:
        z0, L = _z0(B, D, seed=4), None
    dz = O.cotangent(T, B, D + kw.get("augment_dim", 0))
    res = {}
    # ("mlpw": k_mlpw where k_mlpb — W₂ as register blocks, the weight gradient folded on the CU; round 4 — is the default)
    for fam, opts in (("new", {}), ("mlpw", {"mlpb": 0}), ("tiles", {"mlpv": 0, "mlp64": 0, "mlpw": 0}), ("mlpv", {"mlp64": 0, "mlpw": 0})):
        nat, _ = _native(W, options=opts, **kw)
        z, ret, st = nat.forward(z0, L, ts)
        g0, gL, gW, sb = nat.adjoint(z, L, ts, dz)
        assert (ret == 0).all() and sb["nfailed"] == 0
        res[fam] = (z, g0, gL, gW)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    for fam in ("new", "mlpw", "mlpv"):
        z, g0, gL, gW = res[fam]
        zt, t0, tL, tW = res["tiles"]
        assert np.abs(z - zt).max() <= max(lim, 2e-5) * max(1.0, np.abs(zt).max()), (fam, "z")
        assert rel(g0, t0) <= max(lim, 2e-5) and rel(gW, tW) <= max(lim, 5e-5), (fam, rel(g0, t0), rel(gW, tW))
        if gL is not None:
            assert rel(gL, tL) <= max(lim, 2e-5), fam
