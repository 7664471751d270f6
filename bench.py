#!/usr/bin/env python3
"""bench.py — trajectories/sec (forward solve + adjoint) of the latent-ODE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload goku_pendulum|c2|c3|c4] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one lde_forward + one lde_adjoint over one batch of synthetic trajectories whose inputs are
already resident in HBM. Default workload = BASELINE.json's metric config: GOKU pendulum (D=2, P=1, T=50,
t=0:0.05:2.45, Tsit5 abstol=1e-6 reltol=1e-3), batch 256 PER GPU (weak scaling: the batch shards by
trajectory with no data-path collective; a collective exists only for workloads with shared RHS-MLP
weights, whose dW is all-reduced once per step).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     dominant kernel's algorithmic bytes / its average launch duration (HIP events on the launch stream)
  "cpu_baseline": the CPU oracle (a port of the same algorithm, OpenMP over trajectories like the reference's
                  EnsembleThreads [REF src/models/GOKU.jl:121]) timed on this box's host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector == matrix (f32-in MFMA)

WORKLOADS = {
    # name: dict(describing BASELINE.json configs; SURVEY.md §8d inputs)
    "goku_pendulum": dict(desc="GOKU pendulum D=2 P=1 T=50 Tsit5 abstol=1e-6 reltol=1e-3, per-trajectory", B=256,
                          rhs="pendulum", D=2, P=1, T=50, layers=(), solver="tsit5", batching="per_trajectory"),
    "c2": dict(desc="LatentODE D=8 8-200-200-8 relu, RK4 fixed dt=0.05, coupled", B=256, rhs="mlp", D=8, P=0, T=50,
               layers=(8, 200, 200, 8), solver="rk4", batching="coupled", dt=0.05),
    "c3": dict(desc="GOKU pendulum + 2-64-64-2 MLP, Tsit5, per-trajectory", B=1024, rhs="pendulum_plus_mlp", D=2, P=1,
               T=50, layers=(2, 64, 64, 2), solver="tsit5", batching="per_trajectory"),
    "c4": dict(desc="LatentODE D=32 32-128-128-32 relu, Tsit5, coupled (512 per GPU)", B=512, rhs="mlp", D=32, P=0,
               T=50, layers=(32, 128, 128, 32), solver="tsit5", batching="coupled"),
}


def build_problem(w, B, seed_shift=0):
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd import synthetic as O
    lib = L.load()
    d = L.ProblemDesc()
    lib.lde_problem_desc_default(C.byref(d))
    d.rhs_kind = {"pendulum": L.RHS_PENDULUM, "mlp": L.RHS_MLP, "pendulum_plus_mlp": L.RHS_PENDULUM_PLUS_MLP}[w["rhs"]]
    d.state_dim, d.param_dim = w["D"], w["P"]
    d.n_layers = max(len(w["layers"]) - 1, 0)
    for i, s in enumerate(w["layers"]):
        d.layer_sizes[i] = s
    d.solver = L.SOLVER_TSIT5 if w["solver"] == "tsit5" else L.SOLVER_RK4
    d.batching = L.BATCH_COUPLED if w["batching"] == "coupled" else L.BATCH_PER_TRAJECTORY
    if w["solver"] == "rk4":
        d.adaptive, d.dt = 0, w["dt"]
    T, D = w["T"], w["D"]
    ts = O.time_grid(T)
    if w["rhs"] == "mlp":
        rng = np.random.default_rng(1 + seed_shift)
        z0 = (0.5 * rng.standard_normal((B, D))).astype(np.float32)
        theta = None
    else:
        z0, theta = O.pendulum_inputs(B, seed=1 + seed_shift)
    W = O.mlp_weights(w["layers"], seed=3, scale=1.0) if w["layers"] else None
    dz = O.cotangent(T, B, D, seed=2 + seed_shift)
    return d, ts, z0, theta, W, dz


def alg_bytes_per_traj(w):
    """fwd reads 4(D+P), writes 4·D'·T; adjoint reads 4·D'·T, writes 4(D+P)."""
    D, P, T = w["D"], w["P"], w["T"]
    fwd = 4 * (D + P) + 4 * D * T
    bwd = 4 * D * T + 4 * (D + P)
    return fwd, bwd


def flops_per_eval(w):
    return 2 * sum(a * b for a, b in zip(w["layers"][:-1], w["layers"][1:])) if w["layers"] else 0


def cpu_baseline(w, d_native, ts, z0, theta, W, dz, budget_s=12.0):
    """Time the CPU oracle (same algorithm, -O3 -march=native build, OpenMP over trajectories) on a bounded sample."""
    from oracle import oracle as O
    try:
        orc = O.Oracle("f32", native=True)
    except Exception:
        orc = O.Oracle("f32")
    od = O.Desc()
    C.memmove(C.byref(od), C.byref(d_native), C.sizeof(od))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    B = z0.shape[0]
    # bounded sample: the same workload, at most `cap` trajectories per pass
    cap = B if w["batching"] == "per_trajectory" else min(B, 64)
    z0s, ths, dzs = z0[:cap], (None if theta is None else theta[:cap]), dz[:, :cap]

    def one(nt):
        z, _, _ = orc.forward(od, z0s, ths, ts, W=W, nthreads=nt)
        orc.adjoint(od, z, ths, ts, dzs, W=W, nthreads=nt)

    def rate(nt, budget):
        one(nt)  # warm-up (thread pool, page faults)
        n, t0 = 0, time.perf_counter()
        while True:
            one(nt)
            n += 1
            el = time.perf_counter() - t0
            if el > budget:
                return n, el

    # EnsembleThreads analogue: pick the thread count that is fastest on this box (more threads than
    # trajectories-worth-of-work only adds fork/join cost), then time that one for the budget.
    cands = [1] if w["batching"] != "per_trajectory" else sorted({1, 8, 16, 32, 64, min(avail, 128), avail} & set(range(1, avail + 1)))
    best, best_r = 1, 0.0
    for nt in cands:
        n, el = rate(nt, 0.8)
        if n * cap / el > best_r:
            best, best_r = nt, n * cap / el
    nthreads = best
    n, el = rate(nthreads, budget_s)
    return dict(value=n * cap / el, unit="trajectories/s", cores=nthreads, kind="port",
                sample=f"{n} passes of fwd+adjoint over {cap} trajectories of the same workload, {el:.1f} s wall, "
                       f"{nthreads} OpenMP thread(s) (fastest of {cands} on {avail} available cores)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="goku_pendulum", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also report a large-batch sweep (extra keys, rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from latentdiffeq_amd import _lib as L
    lib = L.load()
    w = WORKLOADS[args.workload]
    B = args.batch or w["B"]
    d, ts, z0, theta, W, dz = build_problem(w, B, seed_shift=rank)
    T, D, P = w["T"], w["D"], w["P"]

    h = C.c_void_p()
    L.check(lib.lde_create(C.byref(d), C.byref(h)), None, "lde_create")
    nW = int(lib.lde_num_weights(C.byref(d)))
    if nW:
        L.check(lib.lde_set_weights(h, W.ctypes.data_as(C.c_void_p), nW), h, "lde_set_weights")
    L.check(lib.lde_reserve(h, B, T), h, "lde_reserve")

    dev = torch.device("cuda", local)
    z0d = torch.from_numpy(z0).to(dev)
    thd = torch.from_numpy(theta).to(dev) if theta is not None else None
    dzd = torch.from_numpy(dz).to(dev)
    zout = torch.empty((T, B, D), device=dev)
    ret = torch.empty((B,), device=dev, dtype=torch.int32)
    dz0 = torch.empty((B, D), device=dev)
    dth = torch.empty((B, P), device=dev) if P else None
    dW = torch.zeros((nW,), device=dev) if nW else None
    tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def fwd():
        L.check(lib.lde_forward(h, p(z0d), p(thd), tsp, T, B, p(zout), p(ret), sp), h, "lde_forward")

    def bwd():
        if dW is not None:
            dW.zero_()
        L.check(lib.lde_adjoint(h, p(zout), p(thd), tsp, T, B, p(dzd), p(dz0), p(dth), p(dW), sp), h, "lde_adjoint")
        if dW is not None and world > 1:  # the one collective of the path: shared RHS-MLP gradient
            dist.all_reduce(dW)

    def step():
        fwd()
        bwd()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    ms_per_step = el / args.steps * 1e3
    value = B * world * args.steps / el

    # per-kernel launch durations, HIP events on the launch stream (separate pass, same launches)
    def kernel_ms(fn, n):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record(stream)
            fn()
            b.record(stream)
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    nprobe = min(args.steps, 100)
    fwd()
    fwd_ms = kernel_ms(fwd, nprobe)
    bwd_ms = kernel_ms(bwd, nprobe)

    st = L.Stats()
    lib.lde_get_stats(h, 0, C.byref(st), sp)
    fstat = dict(nfe=st.nfe, naccept=st.naccept, nreject=st.nreject, nfailed=st.nfailed, max_steps=st.max_steps)
    lib.lde_get_stats(h, 1, C.byref(st), sp)
    bstat = dict(nfe=st.nfe, naccept=st.naccept, nreject=st.nreject, nfailed=st.nfailed, max_steps=st.max_steps)

    fb, bb = alg_bytes_per_traj(w)
    dom, dom_ms, dom_bytes = ("lde_adjoint", bwd_ms, bb) if bwd_ms >= fwd_ms else ("lde_forward", fwd_ms, fb)
    Ff = flops_per_eval(w)
    if Ff and w["batching"] == "coupled":
        flops = (fstat["nfe"] * Ff + bstat["nfe"] * 3 * Ff) * B
    else:
        flops = fstat["nfe"] * Ff + bstat["nfe"] * 3 * Ff
    if Ff:  # MLP right-hand side: compute-bound on the f32 MFMA/VALU rate
        ach = flops / ((fwd_ms + bwd_ms) * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel="lde_forward+lde_adjoint", achieved=ach, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=ach / FP32_PEAK_TFLOPS, traffic=None)
    else:
        ach = dom_bytes * B / (dom_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                    traffic=None, alg_bytes_per_launch=dom_bytes * B, avg_launch_ms=dom_ms)

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    # (profiles/collect.sh + profiles/summarize.py; FETCH_SIZE ×2 on gfx950, WRITE_SIZE as is — MI355X_MICROARCH.md §HBM)
    prof = os.path.join(ROOT, "profiles", f"r1_{args.workload}_b{B}_summary.json")
    if os.path.exists(prof) and not Ff:
        kn = {"lde_forward": "k_pend_forward", "lde_adjoint": "k_pend_adjoint"}[dom]
        for name, kd in json.load(open(prof))["kernels"].items():
            if name.startswith(kn) and "write_bytes" in kd:
                roof["traffic"] = kd["fetch_bytes_x2_gfx950"] + kd["write_bytes"]
                roof["traffic_source"] = os.path.relpath(prof, ROOT)
                roof["rocprof_avg_launch_ms"] = kd["avg_ns"] * 1e-6

    out = {
        "metric": "trajectories/sec (fwd+adjoint) GOKU pendulum, batch=256, 1/2/4/8 GPU"
        if args.workload == "goku_pendulum" else f"trajectories/sec (fwd+adjoint) {args.workload}",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {w['desc']}", "batch_per_gpu": B, "global_batch": B * world,
                   "save_points": T, "parallelism": f"dp{world} (batch sharded by trajectory, no data-path collective)"
                   if not nW else f"dp{world} (batch sharded; one all-reduce of dW per step)"},
        "roofline": roof,
        "kernel_ms": {"lde_forward": fwd_ms, "lde_adjoint": bwd_ms},
        "solver_stats": {"forward": fstat, "adjoint": bstat},
    }

    if rank == 0 and args.sweep and not nW:
        sweep = {}
        for Bs in (1 << 12, 1 << 16, 1 << 20):
            d2, ts2, z02, th2, _, dz2 = build_problem(w, Bs)
            a = torch.from_numpy(z02).to(dev); b_ = torch.from_numpy(th2).to(dev); c = torch.from_numpy(dz2).to(dev)
            zo = torch.empty((T, Bs, D), device=dev); r = torch.empty((Bs,), device=dev, dtype=torch.int32)
            g0 = torch.empty((Bs, D), device=dev); gt = torch.empty((Bs, P), device=dev)
            lib.lde_reserve(h, Bs, T)

            def f2():
                L.check(lib.lde_forward(h, p(a), p(b_), tsp, T, Bs, p(zo), p(r), sp), h, "fwd")

            def b2():
                L.check(lib.lde_adjoint(h, p(zo), p(b_), tsp, T, Bs, p(c), p(g0), p(gt), C.c_void_p(), sp), h, "bwd")
            f2(); b2(); torch.cuda.synchronize()
            fm, bm = kernel_ms(f2, 10), kernel_ms(b2, 10)
            sweep[str(Bs)] = dict(traj_per_s=Bs / ((fm + bm) * 1e-3), fwd_ms=fm, bwd_ms=bm,
                                  fwd_GBs=fb * Bs / (fm * 1e-3) / 1e9, bwd_GBs=bb * Bs / (bm * 1e-3) / 1e9)
        out["batch_sweep"] = sweep

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, d, ts, z0, theta, W, dz)
    elif rank == 0:
        out["cpu_baseline"] = None

    lib.lde_destroy(h)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
