"""Known-answer tests that pin the CPU oracle (oracle/lde_oracle.c).

The reference ships no tests for this path ([REF test/runtests.jl:4-6] is an empty testset) and Julia is not
available, so the oracle is pinned by independent answers instead:
  1. order conditions of the Tsit5 constants compiled into the oracle,
  2. the exact frictionless pendulum (Jacobi elliptic functions) and scipy DOP853 at 1e-13,
  3. linear RHS (one Dense layer) against scipy.linalg.expm,
  4. gradients: continuous adjoint vs float64 central finite differences and vs torch autograd through an
     unrolled RK4,
  5. the reference's failure semantics: NaN block + retcode, never an exception  [REF src/models/GOKU.jl:114],
  6. layout: [D×B×T] column-major, ẑ[:,:,1] == ẑ₀, zero augmentation  [REF GOKU.jl:125], [REF LatentODE.jl:71].
"""
import itertools

import numpy as np
import pytest
from scipy.integrate import solve_ivp
from scipy.linalg import expm
from scipy.special import ellipj, ellipk

from oracle import oracle as O


# ---------------------------------------------------------------------------------------------- 1
def test_tsit5_order_conditions(o64):
    c, a, bt, r1, r = o64.tableau()
    b = a[6].copy()                      # FSAL: b = last row of A, b7 = 0
    b7 = np.append(b, 0.0)
    A = np.zeros((7, 7))
    A[:, :6] = a
    assert np.allclose(A.sum(axis=1), c, atol=1e-15), "row sums = c"
    assert abs(bt.sum()) < 1e-15, "Σ b̃ = 0"
    e = np.ones(7)
    Ac, c2, c3 = A @ c, c * c, c ** 3
    # all 17 rooted-tree conditions up to order 5: (Φ(t), 1/γ(t))
    conds = [
        (e, 1), (c, 1 / 2), (c2, 1 / 3), (Ac, 1 / 6), (c3, 1 / 4), (c * Ac, 1 / 8), (A @ c2, 1 / 12), (A @ Ac, 1 / 24),
        (c ** 4, 1 / 5), (c2 * Ac, 1 / 10), (c * (A @ c2), 1 / 15), (c * (A @ Ac), 1 / 30), (Ac * Ac, 1 / 20),
        (A @ c3, 1 / 20), (A @ (c * Ac), 1 / 40), (A @ (A @ c2), 1 / 60), (A @ (A @ Ac), 1 / 120),
    ]
    for phi, g in conds:
        assert abs(b7 @ phi - g) < 2e-15
    # the embedded solution b − b̃ is 4th order (first 8 conditions)
    bh = b7 - bt
    for phi, g in conds[:8]:
        assert abs(bh @ phi - g) < 5e-15

    # continuous extension: b_i(1) = b_i and Σ b_i(Θ) c_i^k = Θ^{k+1}/(k+1), k = 0..3
    def bw(th):
        w = np.zeros(7)
        w[0] = th * (1 + th * (r1[0] + th * (r1[1] + th * r1[2])))
        w[1:] = th * th * (r[:, 0] + th * (r[:, 1] + th * r[:, 2]))
        return w
    assert np.allclose(bw(1.0), b7, atol=1e-14)
    for th in (0.1, 0.37, 0.5, 0.9):
        for k in range(4):
            assert abs(bw(th) @ c ** k - th ** (k + 1) / (k + 1)) < 1e-14


# ---------------------------------------------------------------------------------------------- 2
def _pendulum_exact_from_rest(theta0, L, t, G=10.0):
    k = np.sin(theta0 / 2)
    m = k * k
    K = ellipk(m)
    sn, cn, dn, _ = ellipj(K - np.sqrt(G / L) * t, m)
    return 2 * np.arcsin(k * sn)


def test_pendulum_exact_solution_from_rest(o64):
    B = 16
    rng = np.random.default_rng(3)
    th0 = rng.uniform(0.1, 2.5, B)
    L = rng.uniform(1, 2, (B, 1))
    z0 = np.stack([th0, np.zeros(B)], axis=1)
    ts = O.time_grid(50)
    d = O.make_desc(abstol=1e-11, reltol=1e-11)
    z, ret, _ = o64.forward(d, z0, L, ts)
    assert (ret == 0).all()
    exact = np.stack([_pendulum_exact_from_rest(th0[b], L[b, 0], ts) for b in range(B)], axis=1)
    assert np.abs(z[..., 0] - exact).max() < 5e-10


def _dop853(kind, z0, L, ts):
    fr = 0.7 if kind == O.RHS_PENDULUM_FRICTION else 0.0
    out = np.zeros((len(ts), len(z0), 2))
    for b in range(len(z0)):
        s = solve_ivp(lambda t, y: [y[1], -10.0 / L[b, 0] * np.sin(y[0]) - fr * y[1]], (ts[0], ts[-1]), z0[b],
                      method="DOP853", rtol=1e-13, atol=1e-13, t_eval=ts)
        out[:, b] = s.y.T
    return out


@pytest.mark.parametrize("kind", [O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION])
def test_pendulum_vs_dop853_and_tolerance_ladder(o64, o32, kind):
    """Reproduces the ladder of SURVEY.md §6: ≈13 steps/80 evals at the OrdinaryDiffEq defaults, ≈29 at 1e-6/1e-6,
    ≈64 at 1e-8/1e-8, with errors 3e-4 / 7e-7 / 7e-9 against the truth."""
    B = 64
    z0, L = O.pendulum_inputs(B, dtype=np.float64)
    ts = O.time_grid(50)
    truth = _dop853(kind, z0, L, ts)
    exp = {(1e-6, 1e-3): 5e-4, (1e-6, 1e-6): 2e-6, (1e-8, 1e-8): 2e-8}
    prev = None
    for (at, rt), lim in exp.items():
        d = O.make_desc(rhs_kind=kind, abstol=at, reltol=rt)
        z, ret, info = o64.forward(d, z0, L, ts)
        assert (ret == 0).all()
        err = np.abs(z - truth).max()
        assert err < lim, (at, rt, err)
        assert info["nfe"] == 6 * (info["naccept"] + info["nreject"]) + 2 * B
        if prev is not None:
            assert info["naccept"] > prev
        prev = info["naccept"]
    if kind == O.RHS_PENDULUM:
        d = O.make_desc()
        _, _, info = o64.forward(d, z0, L, ts)
        assert 11 <= info["naccept"] / B <= 15 and info["nreject"] == 0
    # fp32 oracle: same algorithm, fp32 state — within solver accuracy of the truth
    z32, _, _ = o32.forward(O.make_desc(rhs_kind=kind), z0, L, ts)
    assert np.abs(z32 - truth).max() < 5e-4
    z32, _, _ = o32.forward(O.make_desc(rhs_kind=kind, abstol=1e-6, reltol=1e-6), z0, L, ts)
    assert np.abs(z32 - truth).max() < 5e-6


def test_energy_drift(o64):
    z0, L = O.pendulum_inputs(32, dtype=np.float64)
    ts = O.time_grid(100)
    z, _, _ = o64.forward(O.make_desc(abstol=1e-9, reltol=1e-9), z0, L, ts)
    E = 0.5 * z[..., 1] ** 2 - 10.0 / L[None, :, 0] * np.cos(z[..., 0])
    assert np.abs(E - E[0]).max() < 1e-7


def test_rk4_fixed_step_convergence_order(o64):
    z0, L = O.pendulum_inputs(8, dtype=np.float64)
    ts = O.time_grid(50)
    truth = _dop853(O.RHS_PENDULUM, z0, L, ts)
    errs = []
    for h in (0.05, 0.025, 0.0125):
        d = O.make_desc(solver=O.SOLVER_RK4, adaptive=False, dt=h)
        z, _, info = o64.forward(d, z0, L, ts)
        assert info["naccept"] == 8 * round(2.45 / h)
        errs.append(np.abs(z - truth).max())
    assert 12 < errs[0] / errs[1] < 20 and 12 < errs[1] / errs[2] < 20  # 4th order: ×16 per halving


def test_off_grid_saveat_uses_dense_output(o64):
    """saveat does not force steps: save times strictly inside steps come from the 4th-order interpolant."""
    z0, L = O.pendulum_inputs(8, dtype=np.float64)
    rng = np.random.default_rng(1)
    ts = np.sort(rng.uniform(0.0, 3.0, 31))
    truth = _dop853(O.RHS_PENDULUM, z0, L, ts)
    d = O.make_desc(abstol=1e-9, reltol=1e-9)
    z, _, info = o64.forward(d, z0, L, ts)
    assert np.abs(z - truth).max() < 1e-7
    # same number of steps as a 2-point solve over the same span (save times do not alter the step sequence)
    _, _, info2 = o64.forward(d, z0, L, ts[[0, -1]])
    assert info["naccept"] == info2["naccept"]
    # RK4 + off-grid save times: cubic Hermite
    d = O.make_desc(solver=O.SOLVER_RK4, adaptive=False, dt=0.01)
    z, _, _ = o64.forward(d, z0, L, ts)
    assert np.abs(z - truth).max() < 1e-6


# ---------------------------------------------------------------------------------------------- 3
@pytest.mark.parametrize("batching", [O.BATCH_PER_TRAJECTORY, O.BATCH_COUPLED])
def test_linear_rhs_vs_expm(o64, batching):
    rng = np.random.default_rng(0)
    D, B, T = 4, 5, 20
    A = rng.standard_normal((D, D)) * 0.5
    bvec = rng.standard_normal(D) * 0.1
    W = np.concatenate([A.flatten(order="F"), bvec])   # destructure order: vec(W) column-major, then b
    z0 = rng.standard_normal((B, D))
    ts = O.time_grid(T, 0.1)
    d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=(D, D), batching=batching, abstol=1e-11,
                    reltol=1e-11)
    z, _, _ = o64.forward(d, z0, None, ts, W=W)
    M = np.zeros((D + 1, D + 1))
    M[:D, :D], M[:D, D] = A, bvec
    for j, t in enumerate(ts):
        E = expm(M * t)
        assert np.abs((E[:D, :D] @ z0.T).T + E[:D, D] - z[j]).max() < 1e-10


def test_rotation_field_generates_sinusoids(o64):
    """ż = [[0,−ω],[ω,0]] z embedded in D=8 (the accuracy KAT for BASELINE config 2: RK4, fixed dt)."""
    D, B, T, w = 8, 6, 50, 2.0
    A = np.zeros((D, D))
    for i in range(0, D, 2):
        A[i, i + 1], A[i + 1, i] = -w, w
    W = np.concatenate([A.flatten(order="F"), np.zeros(D)])
    rng = np.random.default_rng(2)
    z0 = rng.standard_normal((B, D))
    ts = O.time_grid(T)
    d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=(D, D), batching=O.BATCH_COUPLED,
                    solver=O.SOLVER_RK4, adaptive=False, dt=0.05)
    z, _, info = o64.forward(d, z0, None, ts, W=W)
    assert info["naccept"] == 49 and info["nfe"] == 1 + 4 * 49
    c, s = np.cos(w * ts)[:, None], np.sin(w * ts)[:, None]
    for i in range(0, D, 2):
        ex0 = c * z0[None, :, i] - s * z0[None, :, i + 1]
        assert np.abs(z[..., i] - ex0).max() < 1e-5  # RK4 at h = 0.05, ωh = 0.1


# ---------------------------------------------------------------------------------------------- 4
def _fd(loss, x, eps=1e-6, idx=None):
    g = np.zeros(x.size if idx is None else len(idx))
    flat = x.reshape(-1)
    for n, i in enumerate(range(x.size) if idx is None else idx):
        e = np.zeros_like(flat)
        e[i] = eps
        g[n] = (loss((flat + e).reshape(x.shape)) - loss((flat - e).reshape(x.shape))) / (2 * eps)
    return g


@pytest.mark.parametrize("kind,sense", itertools.product([O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION],
                                                         [O.SENSE_BACKSOLVE_CHECKPOINTED, O.SENSE_BACKSOLVE,
                                                          O.SENSE_PARALLEL_CHECKPOINTED]))
def test_pendulum_adjoint_vs_finite_differences(o64, kind, sense):
    B, T = 6, 50
    z0, L = O.pendulum_inputs(B, dtype=np.float64)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2, dtype=np.float64) * B * T
    d = O.make_desc(rhs_kind=kind, abstol=1e-11, reltol=1e-11, sensealg=sense)
    z, _, _ = o64.forward(d, z0, L, ts)
    g0, gL, _, _ = o64.adjoint(d, z, L, ts, dz)
    f0 = _fd(lambda x: (o64.forward(d, x, L, ts)[0] * dz).sum(), z0).reshape(B, 2)
    fL = _fd(lambda x: (o64.forward(d, z0, x, ts)[0] * dz).sum(), L).reshape(B, 1)
    assert np.abs(g0 - f0).max() < 2e-7 * np.abs(f0).max()
    assert np.abs(gL - fL).max() < 2e-7 * np.abs(fL).max()
    # at the OrdinaryDiffEq default tolerance the adjoint is still accurate to ~1e-5 (49 forced stops keep dt ≤ 0.05)
    dd = O.make_desc(rhs_kind=kind, sensealg=sense)
    zd, _, _ = o64.forward(dd, z0, L, ts)
    h0, hL, _, info = o64.adjoint(dd, zd, L, ts, dz)
    assert np.abs(h0 - f0).max() < 5e-4 * np.abs(f0).max() and np.abs(hL - fL).max() < 5e-4 * np.abs(fL).max()
    assert info["naccept"] >= B * (T - 1)


@pytest.mark.parametrize("batching", [O.BATCH_PER_TRAJECTORY, O.BATCH_COUPLED])
@pytest.mark.parametrize("solver", [O.SOLVER_TSIT5, O.SOLVER_RK4])
def test_mlp_adjoint_vs_finite_differences(o64, batching, solver):
    """tanh MLP (smooth, so finite differences are clean), with one augmented state row."""
    rng = np.random.default_rng(0)
    B, D, aug, H, T = 4, 3, 1, 12, 10
    Dp = D + aug
    layers = (Dp, H, H, Dp)
    W = O.mlp_weights(layers, seed=3, dtype=np.float64)
    z0 = rng.standard_normal((B, D)) * 0.5
    ts = O.time_grid(T, 0.1)
    dz = rng.standard_normal((T, B, Dp))
    kw = dict(adaptive=False, dt=0.0125) if solver == O.SOLVER_RK4 else dict(abstol=1e-11, reltol=1e-11)
    d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, augment_dim=aug, layers=layers, batching=batching,
                    activation=O.ACT_TANH, solver=solver, **kw)
    # the loss is always evaluated with a fine fixed-step solve (smooth in the inputs)
    dl = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, augment_dim=aug, layers=layers, batching=batching,
                     activation=O.ACT_TANH, solver=O.SOLVER_RK4, adaptive=False, dt=0.0125 / 2)
    z, _, _ = o64.forward(d, z0, None, ts, W=W)
    assert z.shape == (T, B, Dp) and np.array_equal(z[0, :, :D], z0) and (z[0, :, D:] == 0).all()
    g0, _, gW, _ = o64.adjoint(d, z, None, ts, dz, W=W)
    f0 = _fd(lambda x: (o64.forward(dl, x, None, ts, W=W)[0] * dz).sum(), z0).reshape(B, D)
    idx = rng.choice(W.size, 40, replace=False)
    fW = _fd(lambda w: (o64.forward(dl, z0, None, ts, W=w)[0] * dz).sum(), W, idx=idx)
    assert np.abs(g0 - f0).max() < 1e-6 * np.abs(f0).max()
    assert np.abs(gW[idx] - fW).max() < 1e-6 * np.abs(fW).max()


def test_pendulum_plus_mlp_adjoint_vs_finite_differences(o64):
    rng = np.random.default_rng(1)
    B, T = 4, 10
    layers = (2, 16, 16, 2)
    W = O.mlp_weights(layers, seed=3, scale=0.5, dtype=np.float64)
    z0, L = O.pendulum_inputs(B, dtype=np.float64)
    ts = O.time_grid(T, 0.1)
    dz = rng.standard_normal((T, B, 2))
    mk = lambda **kw: O.make_desc(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers, activation=O.ACT_TANH, **kw)
    d = mk(abstol=1e-11, reltol=1e-11)
    dl = mk(solver=O.SOLVER_RK4, adaptive=False, dt=0.005)
    z, _, _ = o64.forward(d, z0, L, ts, W=W)
    g0, gL, gW, _ = o64.adjoint(d, z, L, ts, dz, W=W)
    f0 = _fd(lambda x: (o64.forward(dl, x, L, ts, W=W)[0] * dz).sum(), z0).reshape(B, 2)
    fL = _fd(lambda x: (o64.forward(dl, z0, x, ts, W=W)[0] * dz).sum(), L).reshape(B, 1)
    idx = rng.choice(W.size, 40, replace=False)
    fW = _fd(lambda w: (o64.forward(dl, z0, L, ts, W=w)[0] * dz).sum(), W, idx=idx)
    assert np.abs(g0 - f0).max() < 1e-6 * np.abs(f0).max()
    assert np.abs(gL - fL).max() < 1e-6 * np.abs(fL).max()
    assert np.abs(gW[idx] - fW).max() < 1e-6 * np.abs(fW).max()


def test_relu_mlp_rk4_adjoint_vs_torch_autograd(o64):
    """Discretise-then-differentiate (torch autograd through the unrolled RK4) and the continuous adjoint integrated
    with the same RK4 agree to O(h⁴) away from relu kinks; the NODE architecture of [REF nODE.jl:12-14]."""
    import torch
    torch.manual_seed(0)
    B, D, H, T = 5, 4, 10, 8
    layers = (D, H, H, D)
    W = O.mlp_weights(layers, seed=5, dtype=np.float64)
    rng = np.random.default_rng(4)
    z0 = rng.standard_normal((B, D)) * 0.5
    ts = O.time_grid(T, 0.1)
    dz = rng.standard_normal((T, B, D))
    h = 0.0125
    d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=layers, batching=O.BATCH_COUPLED,
                    solver=O.SOLVER_RK4, adaptive=False, dt=h)
    z, _, _ = o64.forward(d, z0, None, ts, W=W)
    g0, _, gW, _ = o64.adjoint(d, z, None, ts, dz, W=W)

    Wt = torch.tensor(W, requires_grad=True)
    zt = torch.tensor(z0, requires_grad=True)

    def f(u):
        off, a = 0, u
        for l in range(3):
            i, o = layers[l], layers[l + 1]
            Wl = Wt[off:off + o * i].reshape(i, o).T   # column-major [out×in]
            off += o * i
            bl = Wt[off:off + o]
            off += o
            a = a @ Wl.T + bl
            if l < 2:
                a = torch.relu(a)
        return a
    u, outs = zt, [zt]
    nsub = round(0.1 / h)
    for j in range(1, T):
        for _ in range(nsub):
            k1 = f(u); k2 = f(u + 0.5 * h * k1); k3 = f(u + 0.5 * h * k2); k4 = f(u + h * k3)
            u = u + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        outs.append(u)
    zt_all = torch.stack(outs)
    assert np.abs(zt_all.detach().numpy() - z).max() < 1e-12   # identical forward arithmetic
    (zt_all * torch.tensor(dz)).sum().backward()
    assert np.abs(g0 - zt.grad.numpy()).max() < 2e-4 * np.abs(g0).max()
    assert np.abs(gW - Wt.grad.numpy()).max() < 2e-4 * np.abs(gW).max()


def test_rhs_menu_and_vjp(o64):
    """f and its VJPs for every RHS kind against finite differences of f itself."""
    rng = np.random.default_rng(7)
    for kind in (O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION, O.RHS_MLP, O.RHS_PENDULUM_PLUS_MLP):
        mlp = kind in (O.RHS_MLP, O.RHS_PENDULUM_PLUS_MLP)
        D = 2 if kind != O.RHS_MLP else 5
        P = 1 if kind != O.RHS_MLP else 0
        layers = (D, 7, D) if mlp else ()
        W = O.mlp_weights(layers, seed=1, dtype=np.float64) if mlp else None
        d = O.make_desc(rhs_kind=kind, state_dim=D, param_dim=P, layers=layers, activation=O.ACT_TANH)
        z, th, lam = rng.standard_normal(D), (np.array([1.3]) if P else None), rng.standard_normal(D)
        f, vz, vth, dW = o64.rhs_vjp(d, z, th, lam, W=W)
        assert np.allclose(f, o64.rhs(d, z, th, W=W), atol=1e-15)
        if kind == O.RHS_PENDULUM:
            assert np.allclose(f, [z[1], -10 / 1.3 * np.sin(z[0])], atol=1e-14)   # [REF pendulum.jl:24-25]
        if kind == O.RHS_PENDULUM_FRICTION:
            assert np.allclose(f, [z[1], -10 / 1.3 * np.sin(z[0]) - 0.7 * z[1]], atol=1e-14)  # [REF pendulum.jl:72-73]
        fz = _fd(lambda x: o64.rhs(d, x, th, W=W) @ lam, z)
        assert np.abs(vz - fz).max() < 1e-8
        if P:
            ft = _fd(lambda x: o64.rhs(d, z, x, W=W) @ lam, th)
            assert np.abs(vth - ft).max() < 1e-8
        if mlp:
            fw = _fd(lambda w: o64.rhs(d, z, th, W=w) @ lam, W)
            assert np.abs(dW - fw).max() < 1e-8


# ---------------------------------------------------------------------------------------------- 5
def test_failure_gives_nan_block_not_exception(o32):
    z0, L = O.pendulum_inputs(64)
    ts = O.time_grid(50)
    z, ret, info = o32.forward(O.make_desc(maxiters=3), z0, L, ts)
    assert (ret == 1).all() and np.isnan(z).all() and info["nfailed"] == 64      # LDE_RET_MAXITERS
    # partial failure: only the trajectories that need more than 12 steps fail
    z, ret, info = o32.forward(O.make_desc(maxiters=12), z0, L, ts)
    bad = ret != 0
    assert 0 < bad.sum() < 64 and np.isnan(z[:, bad]).all() and np.isfinite(z[:, ~bad]).all()
    # non-finite input ⇒ that trajectory fails, the others do not
    z0b = z0.copy()
    z0b[5, 0] = np.inf
    z, ret, _ = o32.forward(O.make_desc(), z0b, L, ts)
    assert ret[5] != 0 and (np.delete(ret, 5) == 0).all() and np.isnan(z[:, 5]).all()
    # pullback through a NaN block is zero for that trajectory
    g0, gL, _, info = o32.adjoint(O.make_desc(), z, L, ts, O.cotangent(50, 64, 2))
    assert (g0[5] == 0).all() and gL[5] == 0 and np.isfinite(g0).all() and info["nfailed"] == 1


def test_invalid_descriptions_are_rejected(o32):
    z0, L = O.pendulum_inputs(4)
    ts = O.time_grid(5)
    with pytest.raises(RuntimeError):
        o32.forward(O.make_desc(solver=O.SOLVER_RK4, adaptive=True), z0, L, ts)      # adaptive RK4 unsupported
    with pytest.raises(RuntimeError):
        o32.forward(O.make_desc(adaptive=False, dt=0.0), z0, L, ts)                  # fixed step needs dt
    with pytest.raises(RuntimeError):
        o32.forward(O.make_desc(rhs_kind=O.RHS_MLP, state_dim=2, param_dim=0, layers=(3, 4, 3)), z0, None, ts)


# ---------------------------------------------------------------------------------------------- 6
def test_layout_and_edge_shapes(o32):
    z0, L = O.pendulum_inputs(7)          # ragged batch (not a multiple of anything)
    ts = O.time_grid(50)
    z, ret, _ = o32.forward(O.make_desc(), z0, L, ts)
    assert z.shape == (50, 7, 2) and np.array_equal(z[0], z0)   # ẑ[:,:,1] == ẑ₀ exactly
    # trajectories are independent: solving one column alone gives the same bits  [REF GOKU.jl:111]
    z3, _, _ = o32.forward(O.make_desc(), z0[3:4], L[3:4], ts)
    assert np.array_equal(z3[:, 0], z[:, 3])
    # T = 1 and B = 1
    z1, _, info = o32.forward(O.make_desc(), z0[:1], L[:1], ts[:1])
    assert z1.shape == (1, 1, 2) and np.array_equal(z1[0], z0[:1]) and info["nfe"] == 0
    g0, gL, _, _ = o32.adjoint(O.make_desc(), z1, L[:1], ts[:1], np.ones((1, 1, 2), np.float32))
    assert (g0 == 1).all() and (gL == 0).all()
    # a time grid that does not start at 0
    zs, _, _ = o32.forward(O.make_desc(), z0, L, ts + 3.0)
    assert np.abs(zs - z).max() < 1e-5   # autonomous system


def test_coupled_vs_per_trajectory_semantics(o64):
    """NeuralODE solves the whole [D'×B] matrix as one ODE (shared dt, norm over D'·B) [REF LatentODE.jl:70-72];
    the ensemble solves each column alone [REF GOKU.jl:111]. Same ODE, so results agree to tolerance, but the
    coupled solve takes ONE step sequence and a column's result depends (weakly) on its batch-mates."""
    rng = np.random.default_rng(0)
    D, B, T = 4, 6, 20
    layers = (D, 16, 16, D)
    W = O.mlp_weights(layers, seed=3, dtype=np.float64)
    z0 = rng.standard_normal((B, D)) * 0.5
    ts = O.time_grid(T)
    mk = lambda b, **kw: O.make_desc(rhs_kind=O.RHS_MLP, state_dim=D, param_dim=0, layers=layers, batching=b, **kw)
    zc, _, ic = o64.forward(mk(O.BATCH_COUPLED), z0, None, ts, W=W)
    zp, _, ip = o64.forward(mk(O.BATCH_PER_TRAJECTORY), z0, None, ts, W=W)
    assert np.abs(zc - zp).max() < 2e-3 and not np.array_equal(zc, zp)
    assert ic["naccept"] < ip["naccept"]                    # one sequence vs B sequences
    zc2, _, _ = o64.forward(mk(O.BATCH_COUPLED), z0[:3], None, ts, W=W)
    assert not np.array_equal(zc2, zc[:, :3])               # batch-mates matter in coupled mode
    assert np.abs(zc2 - zc[:, :3]).max() < 2e-3


def test_coupled_mode_threads_over_columns_agree_with_serial(o32):
    """bench.py's cpu_baseline runs the coupled configs with OpenMP over the columns of every stage evaluation; the
    checker itself (nthreads = 0) stays serial. Columns are independent inside a stage, so ẑ is bit-identical and the
    shared-weight gradient differs only by summation order."""
    layers = (8, 24, 24, 8)
    W = O.mlp_weights(layers, seed=3, scale=0.5)
    rng = np.random.default_rng(1)
    B, T = 12, 10
    z0 = (0.5 * rng.standard_normal((B, 8))).astype(np.float32)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 8)
    for kw in (dict(solver=O.SOLVER_TSIT5), dict(solver=O.SOLVER_RK4, adaptive=0, dt=0.05)):
        d = O.make_desc(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=layers, batching=O.BATCH_COUPLED, **kw)
        z1, r1, i1 = o32.forward(d, z0, None, ts, W=W)
        z4, r4, i4 = o32.forward(d, z0, None, ts, W=W, nthreads=4)
        assert np.array_equal(z1, z4) and i1["naccept"] == i4["naccept"]
        g1 = o32.adjoint(d, z1, None, ts, dz, W=W)
        g4 = o32.adjoint(d, z1, None, ts, dz, W=W, nthreads=4)
        assert np.array_equal(g1[0], g4[0])
        assert np.allclose(g1[2], g4[2], rtol=1e-4, atol=1e-7 * np.abs(g1[2]).max())
