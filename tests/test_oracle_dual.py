"""CPU: the oracle's restatement of ForwardDiffSensitivity AS THE REFERENCE EXECUTES IT — the solve on dual numbers
[REF examples/pendulum_friction-less/pendulum.jl:8-11], [REF src/models/GOKU.jl:107, :121]; SciMLSensitivity 7.10.0 / DiffEqBase 6.104.3
[REF Manifest.toml:1200, :292] (oracle/lde_oracle.c: oracle_forward_dual).

What is pinned here:
  * with the step control on the VALUES alone (dual_norm = False) the dual solve takes the primal solve's steps and returns its ẑ bit
    for bit, and Σ_j J_jᵀΔ_j — the pullback the reference forms from the partials — equals the reverse sweep of LDE_SENSE_DISCRETE
    (oracle_adjoint_discrete) on those steps to round-off: forward and reverse mode of ONE derivative;
  * the Jacobians are the derivative of the discrete map: central finite differences of the prescribed-step solve (float64);
  * with the reference's dual-aware norm (ODE_DEFAULT_NORM on duals) the TRAINING solve takes another step sequence than
    the same solve without AD: the size of that deviation — which the kernels do not reproduce (the primal sequence is
    differentiated) — is measured and bounded here, so that it is a number, not a sentence: |Δẑ| 1.5e-4, gradients 3–4e-5 at the
    metric's configuration, inside the solve's own error (3e-4 / 1e-4 against the converged solve).
"""
import numpy as np
import pytest

from oracle import oracle as O


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("kind", [O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION])
@pytest.mark.parametrize("solver", [O.SOLVER_TSIT5, O.SOLVER_RK4])
def test_dual_solve_on_the_primal_steps_is_the_discrete_sensitivity(o32, o64, kind, solver):
    B, T = 24, 50
    z0, L = O.pendulum_inputs(B, seed=4)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    kw = dict(rhs_kind=kind, solver=solver, sensealg=O.SENSE_DISCRETE)
    if solver == O.SOLVER_RK4:
        kw.update(adaptive=False, dt=0.013)
    d = O.make_desc(**kw)
    for orc, tol in ((o64, 1e-11), (o32, 2e-4)):
        z, ret, rec, info = orc.forward_steps(d, z0, L, ts)
        zd, J, (g0, gL), retd, recd, infod = orc.forward_dual(d, z0, L, ts, dz_out=dz, dual_norm=False)
        assert (ret == 0).all() and (retd == 0).all()
        assert np.array_equal(recd["n"], rec["n"]) and infod["naccept"] == info["naccept"] and infod["nreject"] == info["nreject"]
        assert np.array_equal(zd, z), "the partials do not touch the values"
        r0, rL, _, _ = orc.adjoint_discrete(d, z, L, ts, dz, rec)
        assert _rel(g0, r0) <= tol and _rel(gL, rL) <= tol, (_rel(g0, r0), _rel(gL, rL))
        # the pullback is the contraction of the stored Jacobians
        assert np.allclose(np.einsum("tbiq,tbi->bq", J.astype(np.float64), dz)[:, :2], g0, rtol=0, atol=1e-6 * np.abs(g0).max())


def test_jacobians_are_the_derivative_of_the_discrete_map(o64):
    B, T = 6, 30
    z0, L = O.pendulum_inputs(B, seed=8)
    ts = O.time_grid(T)
    d = O.make_desc(sensealg=O.SENSE_DISCRETE)
    z, J, _, ret, rec, _ = o64.forward_dual(d, z0, L, ts, dual_norm=False)
    eps = 1e-6
    for q in range(3):
        zp, Lp, zm, Lm = z0.astype(np.float64).copy(), L.astype(np.float64).copy(), z0.astype(np.float64).copy(), L.astype(np.float64).copy()
        if q < 2:
            zp[:, q] += eps; zm[:, q] -= eps
        else:
            Lp[:, 0] += eps; Lm[:, 0] -= eps
        a, _, _, _ = o64.forward_steps(d, zp, Lp, ts, rec=rec)       # the SAME steps: their sizes are constants of the differentiation
        b, _, _, _ = o64.forward_steps(d, zm, Lm, ts, rec=rec)
        fd = (a - b) / (2 * eps)
        assert np.abs(fd - J[..., q]).max() <= 2e-8 * max(1.0, np.abs(J[..., q]).max()), q


def test_size_of_the_dual_norm_deviation(o64):
    """The reference's training-time solve (duals in the norm) against its inference-time solve (values only), metric configuration:
    how far apart are ẑ and the gradient? Both are solves of the same problem at the same tolerance; they differ like two step sequences."""
    B, T = 256, 50
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    d = O.make_desc(sensealg=O.SENSE_DISCRETE)                      # Tsit5, 1e-6 / 1e-3: what Pendulum() carries
    zp, _, (p0, pL), _, recp, ip = o64.forward_dual(d, z0, L, ts, dz_out=dz, dual_norm=False)
    zd, _, (d0, dL), _, recd, idd = o64.forward_dual(d, z0, L, ts, dz_out=dz, dual_norm=True)
    truth = O.make_desc(abstol=1e-12, reltol=1e-12, sensealg=O.SENSE_DISCRETE)
    zt, _, (t0, tL), _, _, _ = o64.forward_dual(truth, z0, L, ts, dz_out=dz, dual_norm=False)
    steps_p, steps_d = ip["naccept"] / B, idd["naccept"] / B
    dev_z, dev_g0, dev_gL = np.abs(zd - zp).max(), _rel(d0, p0), _rel(dL, pL)
    err_p = (np.abs(zp - zt).max(), _rel(p0, t0), _rel(pL, tL))
    err_d = (np.abs(zd - zt).max(), _rel(d0, t0), _rel(dL, tL))
    print(f"steps/trajectory: primal norm {steps_p:.1f}, dual norm {steps_d:.1f}; |ẑ_dual − ẑ_primal| {dev_z:.2e}; gradient deviation "
          f"{dev_g0:.2e} / {dev_gL:.2e}; against the converged solve: primal {err_p}, dual {err_d}")
    # measured (float64, B = 256): 12.9 against 12.8 accepted steps per trajectory (the dual norm enters the scale AND the error: the ratio
    # barely moves), |ẑ_dual − ẑ_primal| 1.5e-4, gradient deviation 2.6e-5 / 4.1e-5 of the largest entry — both solves within 3e-4 / 1.2e-4 of
    # the converged one: the deviation is well inside what reltol = 1e-3 means
    assert abs(steps_d - steps_p) <= 0.1 * steps_p
    assert dev_z <= 5e-4 and dev_g0 <= 2e-4 and dev_gL <= 2e-4
    assert dev_z <= 2 * max(err_p[0], err_d[0]) and max(err_d) <= 2 * max(err_p) and max(err_p) <= 2 * max(err_d)


def test_failed_trajectory_gives_nan_block_and_zero_gradient(o64):
    B, T = 8, 20
    z0, L = O.pendulum_inputs(B, seed=2)
    ts = O.time_grid(T)
    d = O.make_desc(maxiters=3)
    z, J, (g0, gL), ret, _, info = o64.forward_dual(d, z0, L, ts, dz_out=O.cotangent(T, B, 2))
    assert (ret != 0).all() and np.isnan(z).all() and (g0 == 0).all() and (gL == 0).all() and info["nfailed"] == B
