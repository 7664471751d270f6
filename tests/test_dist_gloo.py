"""The N>1 path on CPU: world_size-2 `gloo` processes.

The solve itself is GPU-only, so the ranks here use the CPU oracle as the stand-in solver (tests may); what is
under test is the multi-process logic the GPU path shares: contiguous column sharding with no data-path collective,
and ONE flat sum-all-reduce of the shared weight gradient that reproduces the single-process gradient of the
global batch (gradient all-reduce equivalence 1 vs N ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from latentdiffeq_amd import dist as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_tile_the_batch():
    for B in (1, 7, 64, 256, 1000):
        for world in (1, 2, 3, 8):
            blocks = [D.shard_bounds(B, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    x = torch.arange(2 * 10 * 3).reshape(2, 10, 3)
    parts = [D.shard_columns(x, r, 4, 1) for r in range(4)]
    assert torch.equal(torch.cat(parts, dim=1), x) and parts[0].data_ptr() == x.data_ptr()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world)
    from oracle import oracle as O
    o32 = O.Oracle("f32")
    layers = (2, 16, 16, 2)
    W = O.mlp_weights(layers, seed=3)
    B, T = 24, 20
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)          # already carries the 1/(B·T) of the GLOBAL batch
    d = O.make_desc(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
    lo, hi = D.shard_bounds(B, rank, world)
    z, _, _ = o32.forward(d, z0[lo:hi], L[lo:hi], ts, W=W, nthreads=1)
    g0, gL, gW, _ = o32.adjoint(d, z, L[lo:hi], ts, dz[:, lo:hi], W=W, nthreads=1)
    # (1) the one collective of the path: flat sum of the shared weight gradient
    buf = torch.from_numpy(gW.copy())
    D.allreduce_flat_(buf)
    # (2) the same through parameter .grad fields (what a trainer uses), as ONE message
    p1, p2 = torch.nn.Parameter(torch.zeros(100)), torch.nn.Parameter(torch.zeros(gW.size - 100))
    p1.grad, p2.grad = torch.from_numpy(gW[:100].copy()), torch.from_numpy(gW[100:].copy())
    D.FlatGradAllReduce([p1, p2])()
    q.put((rank, lo, hi, z, g0, gL, buf.numpy(), torch.cat([p1.grad, p2.grad]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_matches_single_process():
    from oracle import oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the global batch
    o32 = O.Oracle("f32")
    layers = (2, 16, 16, 2)
    W = O.mlp_weights(layers, seed=3)
    B, T = 24, 20
    z0, L = O.pendulum_inputs(B)
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, 2)
    d = O.make_desc(rhs_kind=O.RHS_PENDULUM_PLUS_MLP, layers=layers)
    z, _, _ = o32.forward(d, z0, L, ts, W=W, nthreads=1)
    g0, gL, gW, _ = o32.adjoint(d, z, L, ts, dz, W=W, nthreads=1)
    zs = np.concatenate([r[3] for r in res], axis=1)
    assert np.array_equal(zs, z), "per-trajectory solves are shard-invariant, bit for bit"
    assert np.array_equal(np.concatenate([r[4] for r in res]), g0) and np.array_equal(np.concatenate([r[5] for r in res]), gL)
    for r in res:
        assert np.allclose(r[6], gW, rtol=2e-5, atol=1e-9), "sum over ranks of shard dW == global-batch dW"
        assert np.array_equal(r[6], r[7]), "flat-buffer and per-parameter paths agree exactly"
    assert np.array_equal(res[0][6], res[1][6]), "every rank ends with identical gradients"


def _grad_worker(rank, world, port, q):
    """A small torch model, different data per rank: both modes of FlatGradAllReduce must leave Σ_ranks grad in every .grad."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init("gloo")
    out = {}
    for mode in ("packed", "attached"):
        torch.manual_seed(0)                       # identical weights on every rank
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Tanh(), torch.nn.Linear(3, 2))
        params = list(net.parameters())
        sync = D.FlatGradAllReduce(params, buckets=3, attach=(mode == "attached"))
        grads = []
        for it in range(3):                        # several steps: the per-step state (pending counts, handles) must reset
            if mode == "attached":
                sync.zero_()
            else:
                for p in params:
                    p.grad = None
            g = torch.Generator().manual_seed(100 * it + rank)
            x = torch.randn(11, 5, generator=g)
            net(x).square().sum().backward()
            local = [p.grad.detach().clone() for p in params]
            if mode == "attached":
                # the hooks may already have summed the early buckets in place: the local part is recomputed without them
                ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Tanh(), torch.nn.Linear(3, 2))
                ref.load_state_dict(net.state_dict())
                ref(x).square().sum().backward()
                local = [p.grad.detach().clone() for p in ref.parameters()]
            sync()
            grads.append(([l.numpy() for l in local], [p.grad.detach().clone().numpy() for p in params]))
        if mode == "attached":
            assert sync._attached()                # the views survived three steps
            # gradient accumulation is not silently wrong: a second backward() before the reduce of the first one raises (its
            # buckets' sums are already on the wire); collecting afterwards drains the outstanding handles on every rank alike
            sync.zero_()
            net(x).square().sum().backward()
            try:
                net(x).square().sum().backward()
                raised = False
            except RuntimeError as e:
                raised = "backward() ran again" in str(e)
            assert raised
            sync()
        out[mode] = grads
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_grad_allreduce_packed_and_attached_modes():
    """Both modes against Σ_ranks of the local gradients: the packed one (multi-tensor copy in, one message, copy out) and the
    attached one (.grad fields are views of the flat buffer, bucketed asynchronous all-reduce issued from the
    post-accumulate hooks while the backward is still running)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for mode in ("packed", "attached"):
        for it in range(3):
            want = [a + b for a, b in zip(res[0][mode][it][0], res[1][mode][it][0])]
            for r in range(world):
                for g, w in zip(res[r][mode][it][1], want):
                    assert np.abs(g - w).max() <= 1e-6 * max(1.0, np.abs(w).max()), (mode, it, r)


# ---- LDE_BATCH_COUPLED_GLOBAL: one coupled solve whose batch is sharded over the ranks (SURVEY.md §8e option (ii)) ---------------------
def _worker_global(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init("gloo")
    from oracle import oracle as O
    o64 = O.Oracle("f64")
    layers = (4, 24, 24, 4)
    W = O.mlp_weights(layers, seed=3).astype(np.float64)
    B, T, Dm = 20, 16, 4
    z0 = 0.5 * np.random.default_rng(1).standard_normal((B, Dm))
    ts = O.time_grid(T)
    dz = O.cotangent(T, B, Dm).astype(np.float64)
    kw = dict(rhs_kind=O.RHS_MLP, state_dim=Dm, param_dim=0, layers=layers, activation=O.ACT_TANH)
    lo, hi = D.shard_bounds(B, rank, world)
    calls = [0]
    base = D.global_sum_hook()

    def hook(vals):
        calls[0] += 1
        base(vals)
    o64.set_sum_hook(hook, nscale=B / (hi - lo))
    d = O.make_desc(batching=O.BATCH_COUPLED_GLOBAL, **kw)
    z, _, st = o64.forward(d, z0[lo:hi], None, ts, W=W, nthreads=1)
    g0, _, gW, sb = o64.adjoint(d, z, None, ts, dz[:, lo:hi], W=W, nthreads=1)
    o64.set_sum_hook(None)
    buf = torch.from_numpy(gW.copy())
    D.allreduce_flat_(buf)                                   # the shared weight gradient: the path's one data collective
    if rank == 0:   # the unsharded coupled solve of the whole batch, and the shard-LOCAL norm for contrast
        dc = O.make_desc(batching=O.BATCH_COUPLED, **kw)
        zf, _, stf = o64.forward(dc, z0, None, ts, W=W, nthreads=1)
        f0, _, fW, sbf = o64.adjoint(dc, zf, None, ts, dz, W=W, nthreads=1)
        zl, _, _ = o64.forward(dc, z0[lo:hi], None, ts, W=W, nthreads=1)
        q.put(dict(ez=float(np.abs(z - zf[:, lo:hi]).max()), eg=float(np.abs(g0 - f0[lo:hi]).max() / np.abs(f0).max()),
                   eW=float(np.abs(buf.numpy() - fW).max() / np.abs(fW).max()), steps=(st["naccept"], stf["naccept"]),
                   bsteps=(sb["naccept"], sbf["naccept"]), calls=calls[0], elocal=float(np.abs(zl - zf[:, lo:hi]).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_coupled_global_norm_sharded_equals_unsharded():
    """Two gloo ranks share ONE coupled adaptive solve (float64 oracle as the stand-in solver, its step-control sums going through
    dist.global_sum_hook): rank 0's shard of ẑ, of ∂L/∂ẑ₀ and the all-reduced dW equal the single-process coupled solve of the whole
    batch to 1e-6 (they differ by summation order only: ≈ 1e-13), with the same step counts — while the shard-LOCAL norm
    (LDE_BATCH_COUPLED on the shard) is off at the level of the solver tolerance."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["ez"] <= 1e-6 and res["eg"] <= 1e-6 and res["eW"] <= 1e-6, res
    assert res["steps"][0] == res["steps"][1] and res["bsteps"][0] == res["bsteps"][1], res
    assert res["calls"] >= res["steps"][0] + res["bsteps"][0], res            # one exchange per attempt (+ the initial-step sums)
    assert res["elocal"] > 10 * max(res["ez"], 1e-12), res                    # the mode matters: local control is measurably different
