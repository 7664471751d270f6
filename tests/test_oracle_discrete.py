"""Known-answer tests of the oracle's discrete (exact) sensitivity — LDE_SENSE_DISCRETE — and of its prescribed-step mode.

`ForwardDiffSensitivity()` is the GOKU default [REF examples/pendulum_friction-less/pendulum.jl:11], splatted into `solve`
at [REF src/models/GOKU.jl:107, :121]: the derivative of the DISCRETE solve on its accepted step sequence. Julia is not in the
image, so the oracle's restatement (oracle/lde_oracle.c: discrete_block) is pinned by two independent answers:
  1. torch autograd (float64) through the same steps unrolled by an independent implementation in this file — stage sums, the
     FSAL slope, Tsit5's free interpolant / RK4's cubic Hermite at the save times — on the step sequence the oracle recorded,
  2. central finite differences of the oracle's own PRESCRIBED-step forward solve (no torch, no re-implementation).
Also: a prescribed-step replay reproduces the adaptive solve bit for bit (forward and continuous adjoint), which is what lets
the GPU parity tests put kernel and checker on the same discrete solve (tests/test_gpu_discrete.py).
"""
import numpy as np
import pytest

from oracle import oracle as O

torch = pytest.importorskip("torch")


# ------------------------------------------------------------------------------------------------ an independent unrolled solve
def _interp_w(th, r1, r):
    w = [th * (1 + th * (r1[0] + th * (r1[1] + th * r1[2])))]
    for i in range(6):
        w.append(th * th * (r[i, 0] + th * (r[i, 1] + th * r[i, 2])))
    return w


def _unrolled(f, y0, ts, rt, rdt, ns, solver, tab):
    """y0: tensor [...]; returns the list of ẑ(t_j). f(y) -> dy. Steps (rt[n], rdt[n]), n < ns; the last one ends at ts[-1]."""
    c, a, bt, r1, r = tab
    outs = [y0]
    y = y0
    j = 1
    T = len(ts)
    k1 = f(y)
    for n in range(ns):
        t, dt = float(rt[n]), float(rdt[n])
        last = n == ns - 1
        tnew = ts[-1] if last else float(rt[n + 1])
        if solver == O.SOLVER_TSIT5:
            k = [k1]
            for s in range(1, 6):
                acc = sum(a[s, q] * k[q] for q in range(s))
                k.append(f(y + dt * acc))
            yn = y + dt * sum(a[6, q] * k[q] for q in range(6))
            k.append(f(yn))
        else:
            k2 = f(y + 0.5 * dt * k1)
            k3 = f(y + 0.5 * dt * k2)
            k4 = f(y + dt * k3)
            yn = y + dt / 6.0 * (k1 + 2.0 * (k2 + k3) + k4)
            k = [k1, k2, k3, k4, f(yn)]
        while j < T and ts[j] <= tnew:
            th = (ts[j] - t) / dt
            if th >= 1.0 or (j == T - 1 and last):
                outs.append(yn)
            elif solver == O.SOLVER_TSIT5:
                w = _interp_w(th, r1, r)
                outs.append(y + dt * sum(w[q] * k[q] for q in range(7)))
            else:
                h00, h10 = (1 + 2 * th) * (1 - th) ** 2, th * (1 - th) ** 2
                h01, h11 = th * th * (3 - 2 * th), th * th * (th - 1)
                outs.append(h00 * y + h10 * dt * k[0] + h01 * yn + h11 * dt * k[4])
            j += 1
        y, k1 = yn, k[-1]
    assert j == T
    return outs


def _mlp_fn(Wt, layers, act):
    offs, o = [], 0
    for l in range(len(layers) - 1):
        n_in, n_out = layers[l], layers[l + 1]
        offs.append((o, o + n_in * n_out, n_in, n_out))
        o += n_in * n_out + n_out

    def f(z):   # z [..., Dp]
        a = z
        for l, (w0, w1, n_in, n_out) in enumerate(offs):
            W = Wt[w0:w1].reshape(n_in, n_out)           # column-major [out×in] ⇒ W(o,i) at o + out·i ⇒ as [in, out] row-major
            a = a @ W + Wt[w1:w1 + n_out]
            if l < len(offs) - 1:
                a = torch.relu(a) if act == O.ACT_RELU else torch.tanh(a)
        return a
    return f


def _pend_fn(L, kind):
    def f(z):
        acc = (-10.0 / L) * torch.sin(z[..., 0])
        if kind == O.RHS_PENDULUM_FRICTION:
            acc = acc - 0.7 * z[..., 1]
        return torch.stack([z[..., 1], acc], dim=-1)
    return f


def _torch_grads(d, z0, theta, ts, dz, rec, W, tab):
    B, D = z0.shape
    Dp = D + d.augment_dim
    layers = [d.layer_sizes[i] for i in range(d.n_layers + 1)]
    Wt = None if W is None else torch.tensor(W, dtype=torch.float64, requires_grad=True)
    zt = torch.tensor(z0, dtype=torch.float64, requires_grad=True)
    tht = None if theta is None else torch.tensor(theta, dtype=torch.float64, requires_grad=True)
    pad = torch.zeros(B, Dp - D, dtype=torch.float64)
    y0 = torch.cat([zt, pad], dim=1)
    mlp = _mlp_fn(Wt, layers, d.activation) if d.rhs_kind in (O.RHS_MLP, O.RHS_PENDULUM_PLUS_MLP) else None
    coupled = d.batching != O.BATCH_PER_TRAJECTORY
    loss = 0.0
    zs = np.zeros((len(ts), B, Dp))
    blocks = [slice(0, B)] if coupled else [slice(b, b + 1) for b in range(B)]
    for bi, sl in enumerate(blocks):
        if d.rhs_kind == O.RHS_MLP:
            f = mlp
        else:
            pf = _pend_fn(tht[sl, 0], d.rhs_kind)
            f = (lambda z, pf=pf: pf(z) + mlp(z)) if d.rhs_kind == O.RHS_PENDULUM_PLUS_MLP else pf
        outs = _unrolled(f, y0[sl], ts, rec["t"][bi], rec["dt"][bi], int(rec["n"][bi]), d.solver, tab)
        za = torch.stack(outs)                      # [T, nb, Dp]
        zs[:, sl] = za.detach().numpy()
        loss = loss + (za * torch.tensor(dz[:, sl], dtype=torch.float64)).sum()
    loss.backward()
    return (zs, zt.grad.numpy(), None if tht is None or tht.grad is None else tht.grad.numpy(),
            None if Wt is None else Wt.grad.numpy())


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CASES = {
    # name: (desc kwargs, B, T, needs theta, layers)
    "pendulum": (dict(rhs_kind=O.RHS_PENDULUM), 6, 50),
    "friction": (dict(rhs_kind=O.RHS_PENDULUM_FRICTION), 5, 50),
    "mlp_relu_coupled": (dict(rhs_kind=O.RHS_MLP, state_dim=4, param_dim=0, layers=(4, 24, 24, 4), batching=O.BATCH_COUPLED), 5, 20),
    "mlp_tanh_per_traj": (dict(rhs_kind=O.RHS_MLP, state_dim=3, param_dim=0, augment_dim=1, layers=(4, 16, 16, 4),
                               activation=O.ACT_TANH, batching=O.BATCH_PER_TRAJECTORY), 4, 20),
    "pend_plus_mlp": (dict(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 16, 16, 2), activation=O.ACT_TANH), 4, 30),
    "mlp_rk4_on_grid": (dict(rhs_kind=O.RHS_MLP, state_dim=8, param_dim=0, layers=(8, 20, 20, 8), solver=O.SOLVER_RK4, adaptive=False,
                             dt=0.05, batching=O.BATCH_COUPLED), 4, 12),
    "pendulum_rk4_off_grid": (dict(rhs_kind=O.RHS_PENDULUM, solver=O.SOLVER_RK4, adaptive=False, dt=0.13), 4, 20),
}


def _inputs(name, seed=11):
    kw, B, T = CASES[name]
    d = O.make_desc(**kw)
    rng = np.random.default_rng(seed)
    D, P = d.state_dim, d.param_dim
    Dp = D + d.augment_dim
    if d.rhs_kind == O.RHS_MLP:
        z0 = rng.uniform(-1, 1, (B, D))
        theta = None
    else:
        z0, theta = O.pendulum_inputs(B, seed=seed, dtype=np.float64)
    W = None
    if d.n_layers:
        W = O.mlp_weights([d.layer_sizes[i] for i in range(d.n_layers + 1)], seed=seed + 1, dtype=np.float64)
        if d.rhs_kind == O.RHS_PENDULUM_PLUS_MLP:
            W = 0.3 * W
    ts = O.time_grid(T)
    dz = rng.normal(size=(T, B, Dp))
    return d, z0, theta, ts, dz, W


@pytest.mark.parametrize("name", list(CASES))
def test_discrete_adjoint_vs_torch_autograd_on_the_recorded_steps(o64, name):
    d, z0, theta, ts, dz, W = _inputs(name)
    z, ret, rec, info = o64.forward_steps(d, z0, theta, ts, W=W)
    assert (ret == 0).all()
    zs, gz, gth, gW = _torch_grads(d, z0, theta, ts, dz, rec, W, o64.tableau())
    assert np.abs(zs - z).max() < 1e-12 * max(1.0, np.abs(z).max()), "the unrolled solve is the oracle's solve"
    dz0, dth, dW, inf2 = o64.adjoint_discrete(d, z, theta, ts, dz, rec, W=W)
    assert inf2["nfailed"] == 0
    assert _rel(dz0, gz) < 1e-11
    if theta is not None:
        assert _rel(dth, gth) < 1e-11
    if W is not None:
        assert _rel(dW, gW) < 1e-11


@pytest.mark.parametrize("name", ["pendulum", "friction", "pendulum_rk4_off_grid"])
def test_discrete_adjoint_vs_finite_differences_of_the_prescribed_solve(o64, name):
    d, z0, theta, ts, dz, W = _inputs(name, seed=5)
    z, ret, rec, _ = o64.forward_steps(d, z0, theta, ts, W=W)
    dz0, dth, _, _ = o64.adjoint_discrete(d, z, theta, ts, dz, rec, W=W)

    def loss(z0_, th_):
        zz, _, _, _ = o64.forward_steps(d, z0_, th_, ts, W=W, rec=rec)
        return float((zz * dz).sum())
    eps = 1e-6
    for b in range(z0.shape[0]):
        for i in range(2):
            zp, zm = z0.copy(), z0.copy()
            zp[b, i] += eps
            zm[b, i] -= eps
            fd = (loss(zp, theta) - loss(zm, theta)) / (2 * eps)
            assert abs(fd - dz0[b, i]) < 2e-6 * max(1.0, abs(fd))
        tp, tm = theta.copy(), theta.copy()
        tp[b, 0] += eps
        tm[b, 0] -= eps
        fd = (loss(z0, tp) - loss(z0, tm)) / (2 * eps)
        assert abs(fd - dth[b, 0]) < 2e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize("name", ["pendulum", "mlp_relu_coupled", "mlp_tanh_per_traj", "pend_plus_mlp"])
def test_prescribed_steps_replay_the_adaptive_solve_bit_for_bit(o32, name):
    d, z0, theta, ts, dz, W = _inputs(name)
    z, ret, rec, info = o32.forward_steps(d, z0, theta, ts, W=W)
    z2, ret2, _, info2 = o32.forward_steps(d, z0, theta, ts, W=W, rec=rec)
    assert np.array_equal(z, z2) and np.array_equal(ret, ret2)
    assert info2["nreject"] == 0 and info2["naccept"] == info["naccept"]
    # the plain entry point takes the same steps
    z3, _, _ = o32.forward(d, z0, theta, ts, W=W)
    assert np.array_equal(z, z3)
    # continuous adjoint: recorded reverse-time steps, replayed
    g = o32.adjoint_steps(d, z, theta, ts, dz, W=W)
    g2 = o32.adjoint_steps(d, z, theta, ts, dz, W=W, rec=g[3])
    g0 = o32.adjoint(d, z, theta, ts, dz, W=W)
    for a, b, c in zip(g[:3], g2[:3], g0[:3]):
        if a is not None:
            assert np.array_equal(a, b) and np.array_equal(a, c)
    assert g2[4]["nreject"] == 0 and g2[4]["naccept"] == g[4]["naccept"]


@pytest.mark.parametrize("name", ["pendulum", "mlp_tanh_per_traj"])
def test_discrete_and_continuous_gradients_agree_to_solver_tolerance(o64, name):
    d, z0, theta, ts, dz, W = _inputs(name)
    d.abstol, d.reltol = 1e-9, 1e-9
    z, _, rec, _ = o64.forward_steps(d, z0, theta, ts, W=W)
    a = o64.adjoint_discrete(d, z, theta, ts, dz, rec, W=W)
    c = o64.adjoint(d, z, theta, ts, dz, W=W)
    assert _rel(a[0], c[0]) < 1e-5
    if theta is not None:
        assert _rel(a[1], c[1]) < 1e-5
    if W is not None:
        assert _rel(a[2], c[2]) < 1e-5


def test_discrete_adjoint_f32_close_to_f64_on_the_same_steps(o32, o64):
    d, z0, theta, ts, dz, W = _inputs("mlp_relu_coupled")
    z, _, rec, _ = o32.forward_steps(d, z0, theta, ts, W=W)
    a = o32.adjoint_discrete(d, z, theta, ts, dz, rec, W=W)
    z64, _, _, _ = o64.forward_steps(d, z0, theta, ts, W=W, rec=rec)
    b = o64.adjoint_discrete(d, z64, theta, ts, dz, rec, W=W)
    assert np.abs(z - z64).max() < 2e-5
    assert _rel(a[0], b[0]) < 1e-4 and _rel(a[2], b[2]) < 1e-4


def test_discrete_adjoint_failure_semantics(o32):
    """A NaN block in the saved solution (a failed trajectory [REF src/models/GOKU.jl:114]) gives zero gradients; an unusable record
    (no steps, or more steps than the record holds) likewise — never an exception."""
    d, z0, theta, ts, dz, W = _inputs("pendulum")
    z, _, rec, _ = o32.forward_steps(d, z0, theta, ts)
    zb = z.copy()
    zb[:, 1, :] = np.nan
    dz0, dth, _, info = o32.adjoint_discrete(d, zb, theta, ts, dz, rec)
    assert info["nfailed"] == 1 and (dz0[1] == 0).all() and dth[1, 0] == 0 and np.isfinite(dz0).all()
    short = dict(t=rec["t"].copy(), dt=rec["dt"].copy(), n=rec["n"].copy())
    short["n"][2] = 0
    short["n"][4] = short["t"].shape[1] + 1
    dz0, dth, _, info = o32.adjoint_discrete(d, z, theta, ts, dz, short)
    assert info["nfailed"] == 2 and (dz0[2] == 0).all() and (dz0[4] == 0).all() and (dz0[3] != 0).any()
