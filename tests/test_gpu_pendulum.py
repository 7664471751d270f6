"""GPU parity: HIP per-trajectory pendulum kernels (through the C ABI) vs the CPU oracle on the same inputs.

Tolerances (fp32 state; north_star: final latent-state error within 1e-4 of the reference solve):
  * parity gate, tight tolerance 1e-6/1e-6: |ẑ_kernel − ẑ_oracle32| ≤ 1e-5 and ≤ 1e-5 from float64 truth.
  * default tolerance 1e-6/1e-3: the first, tiny steps have a round-off-dominated error estimate in fp32, so two
    correct fp32 implementations (different FMA contraction / sin) pick step sizes that differ by 10–25 % there and
    land ~1e-5 apart typically and up to ~1.5e-4 in the worst trajectory of a thousand — both ~3e-4 from the truth,
    which is the accuracy reltol=1e-3 buys. Gate: 99 % of trajectories ≤ 1e-4 from the oracle (max ≤ 3e-4), and the
    error against float64 truth no worse than 1.5× the oracle's own + 1e-5 (and ≤ 5e-4).
  * gradients: relative ≤ 5e-4 vs oracle-f32 at default tol, ≤ 1e-4 at tight tol; ≤ 5e-3 / 1e-3 vs the float64 adjoint.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _native(**kw):
    from tests.gpu_util import Native, make_desc, copy_desc_to_oracle
    d = make_desc(**kw)
    return Native(d), copy_desc_to_oracle(d)


@pytest.mark.parametrize("kind", [O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION])
@pytest.mark.parametrize("tol", [(1e-6, 1e-3), (1e-6, 1e-6)])
@pytest.mark.parametrize("B", [64, 256, 1000])
def test_forward_matches_oracle(o32, o64, kind, tol, B):
    nat, od = _native(rhs_kind=kind, abstol=tol[0], reltol=tol[1])
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(50)
    z, ret, st = nat.forward(z0, L, ts)
    zr, retr, info = o32.forward(od, z0, L, ts)
    assert (ret == 0).all() and (retr == 0).all()
    assert np.array_equal(z[0], z0), "ẑ[:,:,1] must equal ẑ₀ exactly"
    default = tol[1] > 1e-4
    per_traj = np.abs(z - zr).max(axis=(0, 2))
    if default:
        assert per_traj.max() <= 3e-4 and np.quantile(per_traj, 0.99) <= 1e-4
    else:
        assert per_traj.max() <= 1e-5
    # step counts agree up to round-off-level controller differences; NFE bookkeeping is exact
    assert abs(st["naccept"] - info["naccept"]) <= 0.02 * info["naccept"] + 1
    assert st["nfe"] == 6 * (st["naccept"] + st["nreject"]) + 2 * B
    # against float64 truth
    dtruth = O.make_desc(rhs_kind=kind, abstol=1e-10, reltol=1e-10)
    zt, _, _ = o64.forward(dtruth, z0, L, ts)
    e_k, e_o = np.abs(z - zt).max(), np.abs(zr - zt).max()
    assert e_k <= (min(5e-4, 1.5 * e_o + 1e-5) if default else 1e-5)


@pytest.mark.parametrize("T,dt", [(2, 0.05), (33, 0.05), (65, 0.04), (66, 0.04), (130, 0.02), (200, 0.0125)])
@pytest.mark.parametrize("B", [5, 256])
def test_small_batch_forward_over_save_grid_lengths(o32, B, T, dt):
    """The lanes-as-save-times forward kernel has two instantiations: T − 1 ≤ 64 (a lane serves exactly one save time; the stepping
    loop has no load) and longer grids (a lane serves several and fetches the next one inside the loop). Both sides of the boundary
    (T − 1 = 64, 65) and grids of one, two and four save times per lane against the oracle, at the tight tolerance (≤ 1e-5) and the
    default one (worst trajectory ≤ 3e-4, the gate of test_forward_matches_oracle)."""
    z0, L = O.pendulum_inputs(B, seed=11)
    ts = O.time_grid(T, dt)
    for tol, gate in (((1e-6, 1e-6), 1e-5), ((1e-6, 1e-3), 3e-4)):
        nat, od = _native(abstol=tol[0], reltol=tol[1])
        z, ret, st = nat.forward(z0, L, ts)
        zr, retr, info = o32.forward(od, z0, L, ts)
        assert (ret == 0).all() and (retr == 0).all()
        assert np.array_equal(z[0], z0)
        assert np.abs(z - zr).max() <= gate, (T, tol)
        assert st["nfe"] == 6 * (st["naccept"] + st["nreject"]) + 2 * B


@pytest.mark.parametrize("kind", [O.RHS_PENDULUM, O.RHS_PENDULUM_FRICTION])
@pytest.mark.parametrize("sense", [O.SENSE_BACKSOLVE_CHECKPOINTED, O.SENSE_BACKSOLVE, O.SENSE_PARALLEL_CHECKPOINTED])
@pytest.mark.parametrize("tol", [(1e-6, 1e-3), (1e-6, 1e-6)])
def test_adjoint_matches_oracle(o32, o64, kind, sense, tol):
    B, T = 256, 50
    nat, od = _native(rhs_kind=kind, abstol=tol[0], reltol=tol[1], sensealg=sense)
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    z, _, _ = nat.forward(z0, L, ts)
    g0, gL, _, st = nat.adjoint(z, L, ts, dz)
    r0, rL, _, info = o32.adjoint(od, z, L, ts, dz)
    s0, sL = np.abs(r0).max(), np.abs(rL).max()
    lim = 5e-4 if tol[1] > 1e-4 else 1e-4
    assert np.abs(g0 - r0).max() <= lim * s0
    assert np.abs(gL - rL).max() <= lim * sL
    # step counts agree to a few per cent, not exactly: at 1e-6 the f32 error estimate is round-off dominated, so the accepted
    # steps of two correct implementations differ (a controller bug would show as tens of per cent)
    assert abs(st["naccept"] - info["naccept"]) <= 0.06 * info["naccept"] + 1
    # float64 truth of the same continuous adjoint
    dtruth = O.make_desc(rhs_kind=kind, abstol=1e-11, reltol=1e-11, sensealg=min(sense, 1))
    zt, _, _ = o64.forward(dtruth, z0, L, ts)
    t0, tL, _, _ = o64.adjoint(dtruth, zt, L, ts, dz)
    lim = 5e-3 if tol[1] > 1e-4 else 1e-3
    assert np.abs(g0 - t0).max() <= lim * np.abs(t0).max()
    assert np.abs(gL - tL).max() <= lim * np.abs(tL).max()


@pytest.mark.parametrize("sense", [O.SENSE_BACKSOLVE_CHECKPOINTED, O.SENSE_PARALLEL_CHECKPOINTED])
def test_rk4_fixed_step(o32, sense):
    B, T = 128, 50
    nat, od = _native(solver=O.SOLVER_RK4, adaptive=0, dt=0.0125, sensealg=sense)
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    z, ret, st = nat.forward(z0, L, ts)
    zr, _, info = o32.forward(od, z0, L, ts)
    assert np.abs(z - zr).max() <= 1e-5
    assert st["naccept"] == info["naccept"] == B * 196
    dz = O.cotangent(T, B, 2)
    g0, gL, _, _ = nat.adjoint(z, L, ts, dz)
    r0, rL, _, _ = o32.adjoint(od, z, L, ts, dz)
    assert np.abs(g0 - r0).max() <= 1e-4 * np.abs(r0).max()
    assert np.abs(gL - rL).max() <= 1e-4 * np.abs(rL).max()


def test_off_grid_save_times_and_single_point(o32):
    nat, od = _native()
    z0, L = O.pendulum_inputs(70, seed=5)
    rng = np.random.default_rng(0)
    ts = np.sort(rng.uniform(0.3, 4.0, 23))
    z, ret, _ = nat.forward(z0, L, ts)
    zr, _, _ = o32.forward(od, z0, L, ts)
    assert np.abs(z - zr).max() <= 3e-4
    # T = 1: nothing to integrate
    z1, ret1, _ = nat.forward(z0, L, ts[:1])
    assert np.array_equal(z1[0], z0) and (ret1 == 0).all()
    dz = O.cotangent(1, 70, 2)
    g0, gL, _, _ = nat.adjoint(z1, L, ts[:1], dz)
    assert np.array_equal(g0, dz[0]) and (gL == 0).all()


def test_one_handle_growing_and_shrinking_shapes(o32):
    """One handle used with save grids and batches of changing size (progressive sequence length, ragged last minibatch):
    every workspace regrows cleanly — forward and adjoint match the oracle at every shape."""
    nat, od = _native()
    for T, B in [(2, 64), (3, 64), (50, 64), (7, 256), (50, 256), (51, 300), (4, 5)]:
        z0, L = O.pendulum_inputs(B, seed=T)
        ts = O.time_grid(T)
        z, ret, _ = nat.forward(z0, L, ts)
        zr, _, _ = o32.forward(od, z0, L, ts)
        assert (ret == 0).all() and np.abs(z - zr).max() <= 3e-4, (T, B)
        dz = O.cotangent(T, B, 2)
        g0, gL, _, _ = nat.adjoint(z, L, ts, dz)
        r0, rL, _, _ = o32.adjoint(od, z, L, ts, dz)
        assert np.abs(g0 - r0).max() <= 5e-4 * np.abs(r0).max() and np.abs(gL - rL).max() <= 5e-4 * np.abs(rL).max(), (T, B)


def test_failed_trajectories_give_nan_blocks(o32):
    """maxiters exhausted ⇒ retcode != 0 and a NaN [D×T] block for that trajectory only [REF GOKU.jl:114]."""
    nat, od = _native(maxiters=12)
    z0, L = O.pendulum_inputs(256)
    ts = O.time_grid(50)
    z, ret, st = nat.forward(z0, L, ts)
    zr, retr, info = o32.forward(od, z0, L, ts)
    # a trajectory sitting exactly at maxiters may flip with round-off: this batch has ≈ 25 trajectories that need 12 or 13 attempts, and two
    # correct f32 solves disagree on a few of them (k_pend_forward_sh: 3, k_pend_forward_lp — another order of operations — 6; gate 3 %)
    assert (ret != retr).sum() <= 8
    assert 0 < (ret != 0).sum() < 256
    assert st["nfailed"] == (ret != 0).sum()
    bad = ret != 0
    assert np.isnan(z[:, bad, :]).all() and np.isfinite(z[:, ~bad, :]).all()
    both = ~bad & (retr == 0)
    assert np.abs(z[:, both] - zr[:, both]).max() <= 3e-4
    # pullback through a NaN block: zero gradient for that trajectory, finite for the others
    dz = O.cotangent(50, 256, 2)
    nat2, od2 = _native()  # default maxiters (and the default, time-parallel adjoint): the adjoint itself must not run out of iterations
    g0, gL, _, sb = nat2.adjoint(z, L, ts, dz)
    assert (g0[bad] == 0).all() and (gL[bad] == 0).all() and np.isfinite(g0).all() and np.isfinite(gL).all()
    assert sb["nfailed"] == bad.sum()
    r0, rL, _, _ = o32.adjoint(od2, z, L, ts, dz)
    assert (r0[bad] == 0).all() and np.abs(g0 - r0).max() <= 5e-4 * np.abs(r0).max()


def test_large_batch_properties(o32):
    """B = 2^17 (no oracle at this size): energy conservation, determinism, and agreement of a subsample."""
    B, T = 1 << 17, 50
    nat, od = _native(abstol=1e-6, reltol=1e-6)
    z0, L = O.pendulum_inputs(B, seed=11)
    ts = O.time_grid(T)
    z, ret, _ = nat.forward(z0, L, ts)
    assert (ret == 0).all()
    E = 0.5 * z[..., 1] ** 2 - (10.0 / L[None, :, 0]) * np.cos(z[..., 0])
    assert np.abs(E - E[0]).max() <= 2e-4  # frictionless: ½ω² − (G/L)cos θ is conserved
    z2, _, _ = nat.forward(z0, L, ts)
    assert np.array_equal(z, z2), "bitwise deterministic"
    idx = np.arange(0, B, 997)
    zr, _, _ = o32.forward(od, z0[idx], L[idx], ts)
    assert np.abs(z[:, idx] - zr).max() <= 1e-5


@pytest.mark.parametrize("kind,tol", [(O.RHS_PENDULUM, (1e-6, 1e-3)), (O.RHS_PENDULUM_FRICTION, (1e-6, 1e-6))])
def test_large_batch_adjoint_streams_per_trajectory(o32, o64, kind, tol):
    """B = 2^16 + 5 > 32768: the time-parallel adjoint runs in its streaming form (k_pend_adjoint_stream: a lane per trajectory,
    interval-by-interval control, algorithmic traffic only). Against the oracle's time-parallel adjoint and the float64 adjoint
    on a subsample (trajectories are independent), a trajectory with a NaN block (zero pullback), and bitwise determinism."""
    B, T = (1 << 16) + 5, 50
    nat, od = _native(rhs_kind=kind, abstol=tol[0], reltol=tol[1])
    z0, L = O.pendulum_inputs(B, seed=5)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2, seed=6)
    z, ret, _ = nat.forward(z0, L, ts)
    assert (ret == 0).all()
    z[:, 77] = np.nan                                                   # a failed forward trajectory: the block is a constant
    g0, gL, _, st = nat.adjoint(z, L, ts, dz)
    assert st["nfailed"] == 1 and (g0[77] == 0).all() and gL[77] == 0
    assert st["naccept"] >= (B - 1) * (T - 1) and st["nfe"] == (B - 1) * (T - 1) + 6 * (st["naccept"] + st["nreject"]) + (T - 1)
    idx = np.concatenate([np.arange(0, B, 613), [B - 1]])
    idx = idx[idx != 77]
    r0, rL, _, _ = o32.adjoint(od, z[:, idx], L[idx], ts, dz[:, idx])
    tight = tol[1] < 1e-4
    lim = 1e-4 if tight else 5e-4
    assert np.abs(g0[idx] - r0).max() <= lim * np.abs(r0).max() and np.abs(gL[idx] - rL).max() <= lim * np.abs(rL).max()
    d64 = O.make_desc(rhs_kind=kind, abstol=1e-10, reltol=1e-10, sensealg=O.SENSE_BACKSOLVE_CHECKPOINTED)
    t0, tL, _, _ = o64.adjoint(d64, z[:, idx].astype(np.float64), L[idx].astype(np.float64), ts, dz[:, idx].astype(np.float64))
    lim64 = 1e-3 if tight else 5e-3
    assert np.abs(g0[idx] - t0).max() <= lim64 * np.abs(t0).max() and np.abs(gL[idx] - tL).max() <= lim64 * np.abs(tL).max()
    h0, hL, _, _ = nat.adjoint(z, L, ts, dz)
    assert np.array_equal(g0, h0) and np.array_equal(gL, hL)


def test_torch_api_diffeq_layer(o32):
    """The reference-shaped host API: ẑ = diffeq_layer(decoder, (ẑ₀, θ̂), t), differentiable — here with the continuous (time-parallel)
    adjoint against the oracle's reverse-time solve; the struct's default, ForwardDiffSensitivity = LDE_SENSE_DISCRETE, is held to the
    oracle's discrete sweep in tests/test_gpu_default_sensealg.py and tests/test_gpu_discrete.py."""
    import torch
    import latentdiffeq_amd as la
    B, T = 64, 50
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dec = la.Decoder(la.GOKU_basic(), (None, la.Pendulum(sensealg=la.ParallelAdjoint()), None))
    z0t = torch.tensor(z0.T.copy(), device="cuda", requires_grad=True)      # [D, B]
    tht = torch.tensor(L.T.copy(), device="cuda", requires_grad=True)        # [P, B]
    zhat = la.diffeq_layer(dec, (z0t, tht), ts)
    assert tuple(zhat.shape) == (2, B, T)
    od = O.make_desc()
    zr, _, _ = o32.forward(od, z0, L, ts)
    assert np.abs(zhat.detach().permute(2, 1, 0).cpu().numpy() - zr).max() <= 1e-4
    dz = O.cotangent(T, B, 2)
    (zhat * torch.tensor(dz, device="cuda").permute(2, 1, 0)).sum().backward()
    r0, rL, _, _ = o32.adjoint(od, zr, L, ts, dz)
    assert np.abs(z0t.grad.cpu().numpy().T - r0).max() <= 5e-4 * np.abs(r0).max()
    assert np.abs(tht.grad.cpu().numpy().T - rL).max() <= 5e-4 * np.abs(rL).max()


@pytest.mark.parametrize("B,T", [(300, 100), (40000, 50), (64, 1300), (1, 2), (3, 65), (3, 66)])
def test_parallel_adjoint_variants_agree_with_sequential_kernel(o32, B, T):
    """The time-parallel adjoint has three code paths (fused one-wave, fused multi-wave via LDS, two-kernel coalesced
    form for big batches / very long grids); each must reproduce the sequential checkpointed adjoint to solver tolerance
    (both are the same continuous adjoint; abstol=reltol=1e-6 here ⇒ 2e-4 relative)."""
    par, _ = _native(abstol=1e-6, reltol=1e-6, sensealg=O.SENSE_PARALLEL_CHECKPOINTED)
    seq, od = _native(abstol=1e-6, reltol=1e-6, sensealg=O.SENSE_BACKSOLVE_CHECKPOINTED)
    z0, L = O.pendulum_inputs(B, seed=7)
    ts = O.time_grid(T, 0.05 if T <= 100 else 0.004)
    dz = O.cotangent(T, B, 2, seed=8)
    z, ret, _ = par.forward(z0, L, ts)
    assert (ret == 0).all()
    g0, gL, _, sp = par.adjoint(z, L, ts, dz)
    s0, sL, _, ss = seq.adjoint(z, L, ts, dz)
    assert sp["nfailed"] == 0 and sp["naccept"] >= B * (T - 1)
    assert np.abs(g0 - s0).max() <= 2e-4 * np.abs(s0).max()
    assert np.abs(gL - sL).max() <= 2e-4 * max(np.abs(sL).max(), 1e-12)
    if B <= 300 and T <= 100:   # and the oracle's time-parallel twin
        odp = O.make_desc(abstol=1e-6, reltol=1e-6, sensealg=O.SENSE_PARALLEL_CHECKPOINTED)
        r0, rL, _, info = o32.adjoint(odp, z, L, ts, dz)
        assert np.abs(g0 - r0).max() <= 1e-4 * np.abs(r0).max() and np.abs(gL - rL).max() <= 1e-4 * np.abs(rL).max()
        assert sp["naccept"] == info["naccept"]


@pytest.mark.parametrize("sense", [O.SENSE_BACKSOLVE_CHECKPOINTED, O.SENSE_PARALLEL_CHECKPOINTED])
def test_tsit5_with_fixed_step(o32, sense):
    """`adaptive=false, dt=h` with Tsit5 (a legal kwargs combination of the reference's solve call): no controller, so the
    kernel and the oracle take identical steps and agree to fp32 round-off."""
    nat, od = _native(adaptive=0, dt=0.02, sensealg=sense)
    B, T = 96, 50
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    z, ret, st = nat.forward(z0, L, ts)
    zr, _, info = o32.forward(od, z0, L, ts)
    assert (ret == 0).all() and st["naccept"] == info["naccept"] and st["nreject"] == 0
    assert np.abs(z - zr).max() <= 1e-5
    dz = O.cotangent(T, B, 2)
    g0, gL, _, sb = nat.adjoint(z, L, ts, dz)
    r0, rL, _, ib = o32.adjoint(od, z, L, ts, dz)
    assert sb["naccept"] == ib["naccept"]
    assert np.abs(g0 - r0).max() <= 1e-4 * np.abs(r0).max() and np.abs(gL - rL).max() <= 1e-4 * np.abs(rL).max()


def test_very_long_save_grid(o32):
    """T = 7000 save times: the grid no longer fits the LDS staging buffer and is read from L2 instead."""
    nat, od = _native(abstol=1e-6, reltol=1e-6)
    B, T = 40, 7000
    z0, L = O.pendulum_inputs(B, seed=4)
    ts = O.time_grid(T, 0.0005)
    z, ret, _ = nat.forward(z0, L, ts)
    zr, _, _ = o32.forward(od, z0, L, ts)
    assert (ret == 0).all() and np.abs(z - zr).max() <= 1e-5
    dz = O.cotangent(T, B, 2)
    g0, gL, _, _ = nat.adjoint(z, L, ts, dz)           # T−1 > 1024 ⇒ two-kernel time-parallel form
    seq, ods = _native(abstol=1e-6, reltol=1e-6, sensealg=O.SENSE_BACKSOLVE_CHECKPOINTED)
    s0, sL, _, _ = seq.adjoint(z, L, ts, dz)           # sequential kernel, grid from L2 as well
    assert np.abs(g0 - s0).max() <= 2e-4 * np.abs(s0).max() and np.abs(gL - sL).max() <= 2e-4 * np.abs(sL).max()


def test_save_grid_at_the_lds_limit_of_the_small_batch_kernel(o32):
    """T = 6000: the largest save grid the small-batch forward kernel keeps in LDS beside its step records (≈ 155 KB of the
    160 KB); many saves per step, several record rounds at this tolerance, a ragged last workgroup."""
    nat, od = _native(abstol=1e-6, reltol=1e-6)
    B, T = 70, 6000
    z0, L = O.pendulum_inputs(B, seed=6)
    ts = O.time_grid(T, 0.0005)
    z, ret, _ = nat.forward(z0, L, ts)
    zr, _, _ = o32.forward(od, z0, L, ts)
    assert (ret == 0).all() and np.abs(z - zr).max() <= 1e-5


def _variant_run(options):
    """The forward solve of every case below with the given kernel-choice options (lde_set_option) on the handle."""
    from tests.gpu_util import Native, make_desc
    out = {}
    cases = [("default", dict(), 256, 50), ("tight", dict(abstol=1e-6, reltol=1e-6), 70, 50), ("friction", dict(rhs_kind=O.RHS_PENDULUM_FRICTION), 64, 23),
             ("rk4", dict(solver=O.SOLVER_RK4, adaptive=0, dt=0.013), 65, 50), ("one", dict(), 1, 3), ("long", dict(abstol=1e-8, reltol=1e-8), 130, 200),
             ("fail", dict(maxiters=9), 256, 50), ("mid", dict(), 3000, 50), ("dense", dict(), 130, 600)]
    for name, kw, B, T in cases:
        z0, L = O.pendulum_inputs(B, seed=3)
        ts = np.sort(np.random.default_rng(1).uniform(0.0, 3.0, T)) if name == "friction" else O.time_grid(T, 0.004) if name == "dense" else O.time_grid(T)
        nat = Native(make_desc(**kw))
        for k_, v_ in options.items():
            nat.set_option(k_, v_)
        z, ret, st = nat.forward(z0, L, ts)
        out[name + "_z"], out[name + "_ret"] = z, ret
        out[name + "_st"] = np.array([st["nfe"], st["naccept"], st["nreject"], st["nfailed"]])
    return out


_SINGLE = {}


@pytest.mark.parametrize("variant", ["ws", "tl", "lb", "lb8", "lb32", "sh", "lp"])
def test_small_batch_forward_kernels_match_the_single_wave_kernel(variant):
    """Five forward kernels share the step code and the dense-output formulas: k_pend_forward_sh (B ≤ 256: one trajectory per
    workgroup, a stepping wave + three dense-output waves; option "pend_sh_max_b" forces it for every batch here — the 200-point tight case
    needs several rounds of its 48-step record ring, the maxiters case takes its failure barrier), k_pend_forward_tl (B ≤ 1024: lanes = save
    times), k_pend_forward_ws (B ≤ 16384: a stepping wave + helper waves pipelined through LDS), k_pend_forward (one lane
    per trajectory; "pend_tl_max_b" = "pend_sh_max_b" = "pend_ws" = 0 forces it) and the large-batch form of the first (B ≥ 2¹⁷: a lane per
    trajectory, ẑ rows leave through an LDS ring as whole 512-byte stores; "pend_lb_min_b" = 0 forces it, and the 200-point
    grid with its 16-row ring exercises both the hold and the direct-store overflow). They agree like two correct f32 solves — default and tight
    tolerance, friction with off-grid save times, fixed-step RK4, a single trajectory, 200 save points (several save times
    per lane in the tl kernel), and trajectories that fail (NaN blocks). The kernel choice is a per-handle option (lde_set_option):
    the library reads no environment variable."""
    if "ref" not in _SINGLE:
        _SINGLE["ref"] = _variant_run(dict(pend_ws=0, pend_tl_max_b=0, pend_sh_max_b=0))
    b = _SINGLE["ref"]
    a = _variant_run({"ws": dict(pend_tl_max_b=0, pend_sh_max_b=0),
                      "tl": dict(pend_tl_max_b=1024, pend_sh_max_b=0),
                      "lb": dict(pend_tl_max_b=0, pend_sh_max_b=0, pend_ws=0, pend_lb_min_b=0),
                      # ring of 8 rows with a requested hold of 8 (≥ the ring: the host clamps it to 7 — unclamped, every
                      # lane would sit out every iteration and the solve would never end), and 32 rows with the default hold
                      "lb8": dict(pend_tl_max_b=0, pend_sh_max_b=0, pend_ws=0, pend_lb_min_b=0, pend_lb=8, pend_lb_hold=8),
                      "lb32": dict(pend_tl_max_b=0, pend_sh_max_b=0, pend_ws=0, pend_lb_min_b=0, pend_lb=32),
                      "sh": dict(pend_sh_max_b=1000000, pend_lp=0),
                      # k_pend_forward_lp (round 6; frictionless Tsit5 adaptive solves — the other cases of this variant run k_pend_forward_sh):
                      # the stepping wave in lane pairs on the Nyström form of the step, the helpers interpolating from the stage sines
                      "lp": dict(pend_sh_max_b=1000000)}[variant])
    # different compilations of the same step code: multiply-adds contract differently, so the adaptive step sequences part at
    # round-off level — the kernels agree like two correct f32 solves do (tests above: ≤ 3e-4 at the default tolerance,
    # ≤ 2e-5 at 1e-6), exactly where there is no controller (fixed-step RK4 ≤ 2e-6)
    tol = dict(default=3e-4, friction=3e-4, one=3e-4, fail=3e-4, tight=2e-5, long=2e-5, rk4=2e-6, mid=3e-4, dense=3e-4)   # (dense: ≈ 35 save times inside one step)
    for name, lim in tol.items():
        za, zb, ra, rb = a[name + "_z"], b[name + "_z"], a[name + "_ret"], b[name + "_ret"]
        flips = ra != rb
        # a trajectory sitting exactly at maxiters may flip (another order of operations — k_pend_forward_lp — moves a few more: ≤ 3 %)
        assert flips.sum() <= ((8 if variant == "lp" else 3) if name == "fail" else 0), (name, int(flips.sum()))
        ok = ~flips
        assert np.array_equal(np.isnan(za[:, ok]), np.isnan(zb[:, ok])), name
        assert np.nanmax(np.abs(za[:, ok] - zb[:, ok]), initial=0.0) <= lim, name
        sa, sb = a[name + "_st"], b[name + "_st"]
        # nfe, naccept, nreject, nfailed: within 2 % of the steps (k_pend_forward_lp's controller — 1/q from one exp2, no reciprocal — takes ≈ 3 %
        # fewer steps at 1e-8 for the same accuracy: abl/lp_accuracy.py; 5 % there)
        assert np.all(np.abs(sa - sb) <= (0.05 if variant == "lp" else 0.02) * np.maximum(sb, sb[1]) + 3), (name, sa, sb)


@pytest.mark.parametrize("sense", [O.SENSE_PARALLEL_CHECKPOINTED, O.SENSE_BACKSOLVE_CHECKPOINTED])
def test_large_angles_keep_their_restoring_force(o32, o64, sense):
    """v_sin_f32 / v_cos_f32 are defined for |x/2π| ≤ 256 only (≈ 1608 rad): beyond that the raw instruction returns 0 / 1 and
    a pendulum that has wound up many turns would silently drift in a straight line. The kernels reduce the argument first;
    known answer: the solution from θ₀ + 2πk equals the solution from θ₀ shifted by 2πk (the right-hand side is 2π-periodic)."""
    nat, od = _native(abstol=1e-6, reltol=1e-6, sensealg=sense)
    B, T, k = 64, 50, 320                                        # 2π·320 ≈ 2011 rad > 1608
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    shift = np.float32(2 * np.pi * k)
    z0s = z0.copy()
    z0s[:, 0] += shift
    z, ret, _ = nat.forward(z0s, L, ts)
    zb, _, _ = nat.forward(z0, L, ts)
    assert (ret == 0).all()
    # At |θ| ≈ 2000 the relative tolerance admits an error of reltol·|θ| ≈ 2e-3 per step in the angle and f32 resolves 1.2e-4
    # there, so "equal" means a few 1e-2 here — against a 0.2+ separation from the force-free drift θ₀ + ω₀·t.
    assert np.abs((z[..., 0].astype(np.float64) - float(shift)) - zb[..., 0]).max() <= 5e-2
    assert np.abs(z[..., 1] - zb[..., 1]).max() <= 5e-2
    drift = z0[None, :, 0] + z0[None, :, 1] * ts[:, None]        # what sin ≡ 0 would give
    assert np.abs(zb[..., 0] - drift).max() > 0.2                # (the two are far apart, so the checks above mean something)
    zr, _, _ = o32.forward(od, z0s, L, ts)                       # the oracle uses libm's sinf on the same inputs
    assert np.abs(z[..., 1] - zr[..., 1]).max() <= 5e-2
    dz = O.cotangent(T, B, 2)
    g0, gL, _, _ = nat.adjoint(z, L, ts, dz)
    r0, rL, _, _ = o32.adjoint(od, z, L, ts, dz)
    assert np.abs(g0 - r0).max() <= 2e-2 * np.abs(r0).max() and np.abs(gL - rL).max() <= 2e-2 * np.abs(rL).max()
