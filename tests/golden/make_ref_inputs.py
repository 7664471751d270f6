#!/usr/bin/env python3
"""Write tests/golden/ref_inputs.bson — the inputs `julia/make_reference_vectors.jl` feeds to the REFERENCE (gabrevaya/LatentDiffEq.jl at its
pinned Manifest) so that a maintainer with Julia can produce tests/golden/ref_outputs.bson, the reference-made vectors this repository
cannot make itself (no Julia in the build image; SURVEY.md §8c: parity unpinned).

    python tests/golden/make_ref_inputs.py            # deterministic: seeds below; the file is committed

Container: BSON as BSON.jl reads it (latentdiffeq.jl_amd/bson.py), one document per case, arrays in the reference's layouts:
    z0 [D × B] Float32, theta [P × B] Float32, ts [T] Float64, dz [D' × B × T] Float32 (the cotangent ∂L/∂ẑ), abstol / reltol Float64,
    NODE cases: sizes [n_layers + 1] Int64 and W (Flux.destructure order: per Dense vec(W) column-major [out × in], then b) Float32.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from latentdiffeq_amd import bson  # noqa: E402
from latentdiffeq_amd import synthetic as S  # noqa: E402

CASES = {
    # BASELINE.json configs[0] / the metric's per-trajectory problem at the example's defaults (Tsit5, 1e-6 / 1e-3, ForwardDiffSensitivity)
    "c1_goku_pendulum": dict(kind="pendulum", B=64, T=50, abstol=1e-6, reltol=1e-3, seed=1),
    # the same at abstol = reltol = 1e-6: the parity gate's tolerance (step-size controllers agree far better)
    "c1_goku_pendulum_tight": dict(kind="pendulum", B=64, T=50, abstol=1e-6, reltol=1e-6, seed=1),
    "goku_pendulum_friction": dict(kind="pendulum_friction", B=32, T=50, abstol=1e-6, reltol=1e-3, seed=5),
    # the reference's own LatentODE example: NODE(16) = 16-200-200-16 relu, Tsit5, ONE coupled solve [REF nODE.jl:11-16]
    "latentode_ref_node16": dict(kind="node", B=16, T=50, abstol=1e-6, reltol=1e-3, seed=1, sizes=(16, 200, 200, 16)),
}


def case_inputs(cfg):
    B, T = cfg["B"], cfg["T"]
    ts = S.time_grid(T)
    out = dict(kind=cfg["kind"], abstol=float(cfg["abstol"]), reltol=float(cfg["reltol"]), ts=ts)
    if cfg["kind"] == "node":
        D = cfg["sizes"][0]
        z0 = (0.5 * np.random.default_rng(cfg["seed"]).standard_normal((B, D))).astype(np.float32)
        out.update(z0=z0.T.copy(), sizes=np.asarray(cfg["sizes"], np.int64), W=S.mlp_weights(cfg["sizes"], seed=3, scale=1.0),
                   dz=S.cotangent(T, B, D, seed=2).transpose(2, 1, 0).copy())
    else:
        z0, th = S.pendulum_inputs(B, seed=cfg["seed"])
        out.update(z0=z0.T.copy(), theta=th.T.copy(), dz=S.cotangent(T, B, 2, seed=2).transpose(2, 1, 0).copy())
    return out


def main():
    doc = {name: case_inputs(cfg) for name, cfg in CASES.items()}
    path = os.path.join(HERE, "ref_inputs.bson")
    bson.save(path, **doc)
    print(path, os.path.getsize(path), "bytes;", ", ".join(doc))


if __name__ == "__main__":
    main()
