#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/*.npz from the CPU oracle.

    python tests/golden/make_golden.py

There is no reference implementation to import here (the reference is Julia; no Julia in the image), so these
vectors are produced by the oracle AFTER it passed the known-answer tests of tests/test_oracle_kat.py. They pin
(1) the oracle against silent drift and (2) the HIP kernels on the GPU box, where only these files — not the
generating machinery's environment — are guaranteed identical. Inputs are regenerated from seeds by
latentdiffeq_amd.synthetic; input checksums are stored to catch RNG drift.
Each fixture: inputs, fp32 oracle outputs (ẑ, retcode, stats, dt trace, gradients) and float64 outputs at 1e-11.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[0]: GOKU pendulum, batch 64, Tsit5 defaults
    "c1_goku_pendulum_b64": dict(kind=O.RHS_PENDULUM, B=64, keep=16, T=50),
    # metric config: batch 256
    "metric_goku_pendulum_b256": dict(kind=O.RHS_PENDULUM, B=256, keep=16, T=50),
    "metric_goku_pendulum_b256_tight": dict(kind=O.RHS_PENDULUM, B=256, keep=16, T=50, abstol=1e-6, reltol=1e-6),
    "goku_pendulum_friction_b32": dict(kind=O.RHS_PENDULUM_FRICTION, B=32, keep=16, T=50),
    # configs[1]: LatentODE D=8, 8-200-200-8, RK4 fixed step, coupled (golden at B=16: a coupled solve cannot be subsampled)
    "c2_latentode_rk4_d8_h200_b16": dict(kind=O.RHS_MLP, D=8, P=0, layers=(8, 200, 200, 8), B=16, keep=16, T=50,
                                         solver=O.SOLVER_RK4, adaptive=False, dt=0.05, batching=O.BATCH_COUPLED),
    # configs[2]: pendulum + 2-64-64-2 MLP, Tsit5, per-trajectory, adjoint
    "c3_pendulum_plus_mlp_b32": dict(kind=O.RHS_PENDULUM_PLUS_MLP, layers=(2, 64, 64, 2), B=32, keep=16, T=50),
    # configs[3]: LatentODE D=32, 32-128-128-32, Tsit5 coupled
    "c4_latentode_tsit5_d32_h128_b16": dict(kind=O.RHS_MLP, D=32, P=0, layers=(32, 128, 128, 32), B=16, keep=6, T=50,
                                            batching=O.BATCH_COUPLED),
    # the reference's own LatentODE example: NODE(16) = 16-200-200-16 relu, Tsit5, one coupled solve
    # [REF examples/pendulum_friction-less/model_train_LatentODE.jl:37, :42], [REF nODE.jl:11-16] (the example's batch is 64; the
    # fixture keeps 16 columns of a B = 16 coupled solve — the full batch is tests/test_gpu_baseline_sizes.py)
    "latentode_ref_tsit5_d16_h200_b16": dict(kind=O.RHS_MLP, D=16, P=0, layers=(16, 200, 200, 16), B=16, keep=8, T=50,
                                             batching=O.BATCH_COUPLED),
    # augmented NODE (AugmentedNDELayer), tanh, per-trajectory control
    "latentode_aug_tanh_d6a2_b16": dict(kind=O.RHS_MLP, D=6, P=0, aug=2, layers=(8, 32, 32, 8), B=16, keep=16, T=20,
                                        activation=O.ACT_TANH, batching=O.BATCH_PER_TRAJECTORY),
}


def inputs(cfg):
    B, T = cfg["B"], cfg["T"]
    D, aug = cfg.get("D", 2), cfg.get("aug", 0)
    ts = O.time_grid(T)
    if cfg["kind"] == O.RHS_MLP:
        rng = np.random.default_rng(1)
        z0, theta = (0.5 * rng.standard_normal((B, D))).astype(np.float32), None
    else:
        z0, theta = O.pendulum_inputs(B, seed=1)
    layers = cfg.get("layers", ())
    W = O.mlp_weights(layers, seed=3, scale=1.0) if layers else None
    dz = O.cotangent(T, B, D + aug, seed=2)
    return ts, z0, theta, W, dz


def desc(cfg, **over):
    kw = dict(rhs_kind=cfg["kind"], state_dim=cfg.get("D", 2), param_dim=cfg.get("P", 1), augment_dim=cfg.get("aug", 0),
              layers=cfg.get("layers", ()), activation=cfg.get("activation", O.ACT_RELU),
              solver=cfg.get("solver", O.SOLVER_TSIT5), batching=cfg.get("batching", O.BATCH_PER_TRAJECTORY),
              adaptive=cfg.get("adaptive", True), dt=cfg.get("dt", 0.0), abstol=cfg.get("abstol", 1e-6),
              reltol=cfg.get("reltol", 1e-3))
    kw.update(over)
    return O.make_desc(**kw)


def main(only=None):
    o32, o64 = O.Oracle("f32"), O.Oracle("f64")
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        ts, z0, theta, W, dz = inputs(cfg)
        d = desc(cfg)
        z, ret, info = o32.forward(d, z0, theta, ts, W=W)
        g0, gth, gW, binfo = o32.adjoint(d, z, theta, ts, dz, W=W)
        # float64 "truth" of the same problem (tight tolerance / small fixed step)
        d64 = desc(cfg, abstol=1e-11, reltol=1e-11) if cfg.get("adaptive", True) else desc(cfg, dt=cfg["dt"] / 8)
        z64, _, _ = o64.forward(d64, z0, theta, ts, W=W)
        t0, tth, tW, _ = o64.adjoint(d64, z64, theta, ts, dz, W=W)
        k = cfg["keep"]
        out = dict(
            ts=ts, z0=z0[:k], dz=dz[:, :k], z=z[:, :k], retcode=ret[:k], dz0=g0[:k], z64=z64[:, :k], dz0_64=t0[:k],
            fwd_stats=np.array([info["nfe"], info["naccept"], info["nreject"], info["nfailed"], info["max_steps"]]),
            bwd_stats=np.array([binfo["nfe"], binfo["naccept"], binfo["nreject"], binfo["nfailed"], binfo["max_steps"]]),
            dt_trace=info["dt_trace"][:64],
            checksums=np.array([np.float64(z0.astype(np.float64).sum()), np.float64(dz.astype(np.float64).sum()),
                                np.float64(z.astype(np.float64).sum()), np.float64(g0.astype(np.float64).sum()),
                                0.0 if W is None else np.float64(W.astype(np.float64).sum())]),
        )
        if theta is not None:
            out.update(theta=theta[:k], dtheta=gth[:k], dtheta_64=tth[:k])
        if W is not None:
            # weights are regenerated from the seed; keep a strided sample of dW (and its float64 twin) + its norm
            idx = np.arange(0, W.size, max(1, W.size // 512))
            out.update(dW_idx=idx, dW=gW[idx], dW_64=tW[idx], dW_norm=np.array([np.linalg.norm(gW.astype(np.float64))]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB  fwd {info['naccept']} acc / {info['nreject']} rej, "
              f"bwd {binfo['naccept']} acc / {binfo['nreject']} rej")


def chain_main():
    """Dense-chain fixture (scope row f-1): a small reconstructor-shaped chain, f32 oracle outputs + f64 truth."""
    o32, o64 = O.Oracle("f32"), O.Oracle("f64")
    sizes, acts, skips = (6, 48, 48, 48, 40), (O.CACT_RELU, O.CACT_RELU, O.CACT_RELU, O.CACT_SIGMOID), (0, 1, 1, 0)
    d = O.make_chain_desc(sizes, acts, skips)
    W = O.mlp_weights(sizes, seed=11)
    rng = np.random.default_rng(12)
    N = 24
    x = rng.standard_normal((N, sizes[0])).astype(np.float32)
    dy = (rng.standard_normal((N, sizes[-1])) / N).astype(np.float32)
    y32 = o32.chain_forward(d, W, x)
    dx32, dW32 = o32.chain_backward(d, W, x, dy)
    y64 = o64.chain_forward(d, W.astype(np.float64), x.astype(np.float64))
    dx64, dW64 = o64.chain_backward(d, W.astype(np.float64), x.astype(np.float64), dy.astype(np.float64))
    path = os.path.join(HERE, "chain_decoder.npz")
    np.savez_compressed(path, sizes=np.array(sizes), acts=np.array(acts), skips=np.array(skips), W=W, x=x, dy=dy,
                        y_f32=y32, dx_f32=dx32, dW_f32=dW32, y_f64=y64, dx_f64=dx64, dW_f64=dW64)
    print(f"chain_decoder: {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    import sys
    only = [a for a in sys.argv[1:] if not a.startswith("--")]   # fixture names: regenerate just those
    if "--chain-only" not in sys.argv:
        main(only)
    if not only:
        chain_main()
