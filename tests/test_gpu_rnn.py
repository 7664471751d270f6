"""GPU parity of the recurrent pattern extractor (lde_rnn_*, scope row f-2) against the CPU oracle.

Tolerance: f32 round-off only (hardware exp/rcp in σ, ≈1e-7): forward ≤ 2e-5, gradients ≤ 1e-4 of the float64 scale and
no farther from float64 than 2× the f32 oracle + 2e-5."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

CASES = [
    (O.CELL_RNN_RELU, (32, 16, 16), True),      # pe_z₀          [REF src/models/GOKU.jl:229-230]
    (O.CELL_LSTM, (32, 16, 16), False),         # pe_θ_forward   [REF src/models/GOKU.jl:233-234]
    (O.CELL_LSTM, (32, 16, 16), True),          # pe_θ_backward  [REF src/models/GOKU.jl:236-237]
    (O.CELL_RNN_RELU, (32, 32, 32), True),      # LatentODE      [REF src/models/LatentODE.jl:120-121]
    (O.CELL_RNN_TANH, (32, 16, 16), False),     # the third compile-time instantiation of k_rnn
    (O.CELL_RNN_TANH, (5, 7, 3, 9), False),
    (O.CELL_LSTM, (3, 10), True),
    (O.CELL_LSTM, (40, 16, 11), False),
    (O.CELL_RNN_TANH, (40, 64, 33), True),
]


def _run(cell, sizes, reverse, T, B, o32, o64, seed=4):
    from tests.gpu_util import NativeRnn
    d = O.make_rnn_desc(cell, sizes, reverse)
    W = O.rnn_weights(cell, sizes, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = rng.standard_normal((T, B, sizes[0])).astype(np.float32)
    dy = (rng.standard_normal((B, sizes[-1])) / B).astype(np.float32)
    nat = NativeRnn(cell, sizes, reverse)
    assert nat.nW == W.size
    nat.set_weights(W)
    y = nat.forward(x)
    yr = o32.rnn_forward(d, W, x)
    y64 = o64.rnn_forward(d, W.astype(np.float64), x.astype(np.float64))
    assert np.abs(y - yr).max() <= 2e-5 and np.abs(y - y64).max() <= 2e-5
    dx, dW = nat.backward(x, dy)
    rx, rW = o32.rnn_backward(d, W, x, dy)
    tx, tW = o64.rnn_backward(d, W.astype(np.float64), x.astype(np.float64), dy.astype(np.float64))
    for g, r, t, what in ((dx, rx, tx, "dx"), (dW, rW, tW, "dW")):
        s = np.abs(t).max()
        assert np.isfinite(g).all(), what
        assert np.abs(g - r).max() <= 1e-4 * s, what
        assert np.abs(g - t).max() <= 2 * np.abs(r - t).max() + 2e-5 * s, what
    return nat, (x, dy, dx, dW)


@pytest.mark.parametrize("cell,sizes,reverse", CASES)
@pytest.mark.parametrize("T,B", [(1, 1), (9, 37), (50, 256)])
def test_rnn_forward_backward_parity(o32, o64, cell, sizes, reverse, T, B):
    _run(cell, sizes, reverse, T, B, o32, o64)


def test_rnn_dw_accumulates_dx_optional_and_repeatable(o32, o64):
    nat, (x, dy, dx, dW) = _run(O.CELL_LSTM, (32, 16, 16), True, 12, 40, o32, o64)
    base = np.full(nat.nW, 0.5, np.float32)
    dx2, dW2 = nat.backward(x, dy, need_dx=False, dW0=base)
    assert dx2 is None and np.abs((dW2 - base) - dW).max() <= 1e-6 * max(1.0, np.abs(dW).max())
    assert np.array_equal(nat.backward(x, dy)[1], nat.backward(x, dy)[1])


def test_rnn_errors_are_reported_not_thrown():
    from latentdiffeq_amd import _lib as L
    from tests.gpu_util import NativeRnn
    with pytest.raises(L.LdeError, match="UNSUPPORTED"):
        NativeRnn(O.CELL_LSTM, (8, 32))             # 4·32 gate rows > 64
    with pytest.raises(L.LdeError, match="INVALID_ARG"):
        NativeRnn(7, (8, 8))


def test_torch_encoder_path_end_to_end(o64):
    """encode(encoder, x) = feature extractor → pattern extractor → latent_in through autograd: (μ, logσ²) and the gradients of
    every parameter match the float64 oracle composition (GOKU default layers at reduced input size)."""
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd.recurrent import Encoder, default_encoder_layers, encode
    torch.manual_seed(1)
    NI, B, T = 48, 20, 7
    mt = M.GOKU_basic()
    fe, pe, li = default_encoder_layers(mt, NI, hidden_dim_resnet=40, device="cuda")
    with torch.no_grad():      # biases / initial states away from zero so that every term is exercised
        for m in [fe, *pe, *li]:
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(torch.empty_like(p).uniform_(-0.2, 0.2))
    enc = Encoder(mt, (fe, pe, li))
    x = torch.rand(NI, B, T, device="cuda")
    (mu_z, mu_t), (ls_z, ls_t) = encode(enc, x)
    assert mu_z.shape == (16, B) and mu_t.shape == (16, B) and ls_z.shape == (16, B) and ls_t.shape == (16, B)
    cz, ct, dz_, dt_ = (torch.randn(16, B, device="cuda") for _ in range(4))
    ((mu_z * cz).sum() + (mu_t * ct).sum() + (ls_z * dz_).sum() + (ls_t * dt_).sum()).backward()

    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    cdesc = lambda ch: (O.make_chain_desc(ch.sizes, ch.acts, ch.skips), f64(ch.flat_weights()))
    rdesc = lambda r: (O.make_rnn_desc(r.code, r.sizes, r.reverse), f64(r.flat_weights()))
    dfe, Wfe = cdesc(fe)
    xb = f64(x).transpose(2, 1, 0).reshape(T * B, NI)
    fo = o64.chain_forward(dfe, Wfe, xb)                       # (T*B, 32)
    fo3 = fo.reshape(T, B, -1)
    outs, recs = [], [rdesc(r) for r in pe]
    for (dr, Wr) in recs:
        outs.append(o64.rnn_forward(dr, Wr, fo3))
    pe_z, pe_t = outs[0], np.concatenate([outs[1], outs[2]], axis=1)
    lis = [cdesc(c) for c in li]
    got = [f64(mu_z).T, f64(ls_z).T, f64(mu_t).T, f64(ls_t).T]
    ins = [pe_z, pe_z, pe_t, pe_t]
    cts = [f64(cz).T, f64(dz_).T, f64(ct).T, f64(dt_).T]
    d_pe_z, d_pe_t = np.zeros_like(pe_z), np.zeros_like(pe_t)
    flat_grad = lambda ch: ch.theta.grad.cpu().numpy()
    for i, ((dd, Wl), xin, c_) in enumerate(zip(lis, ins, cts)):
        ref = o64.chain_forward(dd, Wl, xin)
        assert np.abs(got[i] - ref).max() <= 2e-5, f"latent_in {i}"
        dxi, dWl = o64.chain_backward(dd, Wl, xin, c_)
        assert np.abs(flat_grad(li[i]) - dWl).max() <= 2e-4 * np.abs(dWl).max(), f"latent_in {i} dW"
        if i < 2:
            d_pe_z += dxi
        else:
            d_pe_t += dxi
    dys = [d_pe_z, d_pe_t[:, :16], d_pe_t[:, 16:]]
    dfo = np.zeros_like(fo3)
    for r, (dr, Wr), dy in zip(pe, recs, dys):
        dxr, dWr = o64.rnn_backward(dr, Wr, fo3, dy)
        dfo += dxr
        gW = r.theta.grad.cpu().numpy()
        assert np.abs(gW - dWr).max() <= 2e-4 * np.abs(dWr).max(), "recurrent dW"
    _, dWfe = o64.chain_backward(dfe, Wfe, xb, dfo.reshape(T * B, -1), need_dx=False)
    assert np.abs(flat_grad(fe) - dWfe).max() <= 2e-4 * np.abs(dWfe).max(), "feature extractor dW"


def test_rnn_large_batch(o32, o64):
    """B = 5 003 trajectories (313 workgroups, ragged last one), T = 30."""
    _run(O.CELL_LSTM, (32, 16, 16), True, 30, 5003, o32, o64, seed=8)


def test_one_rnn_handle_changing_shapes(o32):
    """One handle with sequence lengths and batches growing and shrinking (progressive sequence length, ragged minibatch)."""
    from tests.gpu_util import NativeRnn
    for cell, sizes in ((O.CELL_LSTM, (32, 16, 16)), (O.CELL_RNN_TANH, (5, 7, 3, 9))):
        d = O.make_rnn_desc(cell, sizes, True)
        W = O.rnn_weights(cell, sizes, seed=3)
        nat = NativeRnn(cell, sizes, True)
        nat.set_weights(W)
        for T, B in ((3, 5), (20, 40), (7, 300), (50, 16), (51, 301), (2, 1)):
            rng = np.random.default_rng(T * 1000 + B)
            x = rng.standard_normal((T, B, sizes[0])).astype(np.float32)
            dy = (rng.standard_normal((B, sizes[-1])) / B).astype(np.float32)
            assert np.abs(nat.forward(x) - o32.rnn_forward(d, W, x)).max() <= 2e-5, (T, B)
            dx, dW = nat.backward(x, dy)
            rx, rW = o32.rnn_backward(d, W, x, dy)
            assert np.abs(dx - rx).max() <= 1e-4 * np.abs(rx).max() and np.abs(dW - rW).max() <= 1e-4 * np.abs(rW).max(), (T, B)


def test_instantiated_and_run_time_shaped_kernels_agree():
    """The reference's default stacks (32 → 16 → 16) run compile-time instantiations of k_rnn; lde_rnn_set_option("generic", 1) forces
    the run-time-shaped kernel every other shape uses. Same source, same order of operations: the two must agree to round-off
    (the compiler may contract multiply-adds differently, nothing more)."""
    from tests.gpu_util import NativeRnn
    for cell in (O.CELL_LSTM, O.CELL_RNN_RELU, O.CELL_RNN_TANH):
        sizes, T, B = (32, 16, 16), 23, 70
        W = O.rnn_weights(cell, sizes, seed=6)
        rng = np.random.default_rng(7)
        x = rng.standard_normal((T, B, 32)).astype(np.float32)
        dy = (rng.standard_normal((B, 16)) / B).astype(np.float32)
        res = []
        for generic in (0, 1):
            nat = NativeRnn(cell, sizes, True)
            nat.set_option("generic", generic)
            nat.set_weights(W)
            y = nat.forward(x)
            dx, dW = nat.backward(x, dy)
            res.append((y, dx, dW))
        for a, b, what in zip(res[0], res[1], ("y", "dx", "dW")):
            assert np.abs(a - b).max() <= 2e-6 * max(np.abs(b).max(), 1e-30), (cell, what)


@pytest.mark.parametrize("cell", [O.CELL_LSTM, O.CELL_RNN_RELU, O.CELL_RNN_TANH])
def test_register_and_lds_weight_rows_agree(cell):
    """Small batches of the default stacks run one wave per workgroup with the weight rows in registers (packed FMAs);
    lde_rnn_set_option("regw", 0) keeps them in LDS (the instantiation larger batches use). Same sums in a different association: round-off."""
    from tests.gpu_util import NativeRnn
    sizes, T, B = (32, 16, 16), 23, 37
    W = O.rnn_weights(cell, sizes, seed=9)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((T, B, sizes[0])).astype(np.float32)
    dy = (rng.standard_normal((B, sizes[-1])) / B).astype(np.float32)
    res = []
    for flag in (1, 0):
        nat = NativeRnn(cell, sizes, True)
        nat.set_option("regw", flag)
        nat.set_weights(W)
        y = nat.forward(x)
        dx, dW = nat.backward(x, dy)
        res.append((y, dx, dW))
    for a, b, what in zip(res[0], res[1], ("y", "dx", "dW")):
        assert np.abs(a - b).max() <= 5e-6 * max(np.abs(b).max(), 1e-30), what


@pytest.mark.parametrize("cell", [O.CELL_LSTM, O.CELL_RNN_RELU, O.CELL_RNN_TANH])
@pytest.mark.parametrize("T,B,rev", [(23, 37, True), (1, 5, False), (8, 16, True), (9, 64, False), (50, 256, True)])
def test_one_wave_per_cell_equals_one_wave_per_stack(cell, T, B, rev):
    """The default stacks run a workgroup of two waves, one per cell, handing h¹ₛ forward and ∂L/∂h¹ₛ backward through an LDS ring
    (rnn_body2); lde_rnn_set_option("pipe", 0) keeps the single wave that walks both cells (rnn_body). Same arithmetic per cell, in the same
    order ⇒ the same bits — outputs, input gradients and weight gradients, on sweeps shorter than, equal to and longer than the
    ring, with ragged batches, both directions of time; the training forward + pullback from kept records as well."""
    from tests.gpu_util import NativeRnn
    sizes = (32, 16, 16)
    W = O.rnn_weights(cell, sizes, seed=9)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((T, B, sizes[0])).astype(np.float32)
    dy = (rng.standard_normal((B, sizes[-1])) / B).astype(np.float32)
    res = []
    for flag in (1, 0):
        nat = NativeRnn(cell, sizes, rev)
        nat.set_option("pipe", flag)
        nat.set_weights(W)
        y = nat.forward(x)
        dx, dW = nat.backward(x, dy)
        res.append((y, dx, dW))
    for a, b, what in zip(res[0], res[1], ("y", "dx", "dW")):
        assert np.array_equal(a, b), what


def test_branch_streams_match_joined_streams():
    """encode() three ways — the three stacks as one autograd node on raw side streams (opt-in), the z₀ / θ branches kept on
    their own HIP streams to the end (default), and the variant that joins the streams after the recurrent stacks — over several
    iterations with changing (B, T): allocator reuse across streams would show up as differing outputs or gradients (round-1
    advisor finding on record_stream coverage). Outputs agree bit for bit; so do the gradients of the two older variants; the
    grouped nodes (raw side streams; round 3: one launch per stage through lde_rnn_group_*) add the three input gradients in their own
    order, so their gradients agree to f32 rounding (1e-6 of the largest)."""
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd import recurrent as R
    torch.manual_seed(2)
    NI = 48
    mt = M.GOKU_basic()
    fe, pe, li = R.default_encoder_layers(mt, NI, hidden_dim_resnet=40, device="cuda")
    enc = R.Encoder(mt, (fe, pe, li))
    params = [p for m in [fe, *pe, *li] for p in m.parameters()]
    keep = R._BRANCH_STREAMS, R._RNN_GROUP, R._RNN_LAUNCH_GROUP, R._ENCODER_FUSED
    R._ENCODER_FUSED = False     # (the one-node encoder has a test of its own below; this one is about the streams of the separate nodes)
    try:
        for it, (B, T) in enumerate([(20, 7), (64, 12), (16, 5), (33, 9), (64, 12), (256, 20)]):
            x = torch.rand(NI, B, T, device="cuda")
            cts = [torch.randn(16, B, device="cuda") for _ in range(4)]
            res = []
            for group, branch, launch in ((True, True, False), (False, True, False), (False, False, False), (False, False, True)):
                R._RNN_GROUP, R._BRANCH_STREAMS, R._RNN_LAUNCH_GROUP = group, branch, launch   # (the last: the stacks through lde_rnn_group_*, round 3)
                for p in params:
                    p.grad = None
                (mz, mt_), (lz, lt) = R.encode(enc, x)
                junk = [torch.randn(64, 64, device="cuda") for _ in range(8)]       # churn the main stream's allocator between the two
                ((mz * cts[0]).sum() + (mt_ * cts[1]).sum() + (lz * cts[2]).sum() + (lt * cts[3]).sum()).backward()
                torch.cuda.synchronize()
                res.append(([t.detach().clone() for t in (mz, mt_, lz, lt)], [p.grad.clone() for p in params]))
                del junk
            for other in res[1:]:
                for a, b in zip(res[0][0], other[0]):
                    assert torch.equal(a, b), (it, B, T)
            for a, b in zip(res[1][1], res[2][1]):
                assert torch.equal(a, b), (it, B, T)
            for grouped in (res[0][1], res[3][1]):      # the two grouped nodes add the three frame gradients in their own order
                for a, b in zip(grouped, res[1][1]):
                    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12, (it, B, T)
    finally:
        R._BRANCH_STREAMS, R._RNN_GROUP, R._RNN_LAUNCH_GROUP, R._ENCODER_FUSED = keep


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_encoder_node_equals_separate_nodes(dtype):
    """encode() on one stream (what a captured step runs) two ways: feature extractor, grouped stacks, vcat, grouped heads as separate
    autograd nodes — and recurrent._GokuEncoderFn, ONE node that makes the same library calls on the same values with the vcat written
    in place by the stacks (lde_rnn_group_forward_ld) and the fan-out sums of the pullback formed where the next kernel reads them
    (lde_rnn_group_backward_ld, lde_chain_backward_saved_sum). Outputs, the frames' gradient and every weight gradient: bit for bit
    (the separate nodes add the three frame gradients as (a + b) + c, like the kernel), ragged and full tiles, f32 and bf16 chains."""
    import torch
    import latentdiffeq_amd as M
    from latentdiffeq_amd import recurrent as R
    torch.manual_seed(5)
    NI = 64
    mt = M.GOKU_basic()
    fe, pe, li = R.default_encoder_layers(mt, NI, hidden_dim_resnet=48, device="cuda")
    for m in (fe, *li):
        m.set_dtype(dtype)
    enc = R.Encoder(mt, (fe, pe, li))
    params = [p for m in [fe, *pe, *li] for p in m.parameters()]
    keep = R._BRANCH_STREAMS, R._RNN_GROUP, R._RNN_LAUNCH_GROUP, R._ENCODER_FUSED
    try:
        R._BRANCH_STREAMS, R._RNN_GROUP, R._RNN_LAUNCH_GROUP = False, False, True
        for B, T in ((20, 7), (64, 12), (33, 9), (256, 50)):
            x0 = torch.rand(T, B, NI, device="cuda")
            cts = [torch.randn(16, B, device="cuda") for _ in range(4)]
            res = []
            for fused in (True, False):
                R._ENCODER_FUSED = fused
                for p in params:
                    p.grad = None
                x = x0.clone().requires_grad_(True)
                (mz, mt_), (lz, lt) = R.encode(enc, x.permute(2, 1, 0))
                ((mz * cts[0]).sum() + (mt_ * cts[1]).sum() + (lz * cts[2]).sum() + (lt * cts[3]).sum()).backward()
                torch.cuda.synchronize()
                res.append(([t.detach().clone() for t in (mz, mt_, lz, lt)], [p.grad.clone() for p in params] + [x.grad.clone()]))
            for a, b in zip(res[0][0], res[1][0]):
                assert torch.equal(a, b), (B, T, "outputs")
            for i, (a, b) in enumerate(zip(res[0][1], res[1][1])):
                assert torch.equal(a, b), (B, T, "gradient", i)
        # the node is the one that ran
        R._ENCODER_FUSED = True
        (mz, _), _ = R.encode(enc, x0.permute(2, 1, 0))
        assert "GokuEncoderFn" in type(mz.grad_fn.next_functions[0][0]).__name__
    finally:
        R._BRANCH_STREAMS, R._RNN_GROUP, R._RNN_LAUNCH_GROUP, R._ENCODER_FUSED = keep


def test_launch_grouped_stacks_equal_separate_calls():
    """lde_rnn_group_forward / lde_rnn_group_backward (recurrent._RecurrentLaunchGroupFn): the GOKU encoder's three pattern extractors
    on the same frames — one RNN stack and two LSTM stacks of the default shape — with every stage of the call ONE launch. Per stack
    the same kernels on the same arguments as the separate calls: outputs, the frame gradient and every stack's weight gradient equal
    them bit for bit; a non-default shape in the group (run one after the other by the library) too."""
    import torch
    from latentdiffeq_amd.recurrent import LSTM, RNN, Recurrent, _RecurrentLaunchGroupFn
    torch.manual_seed(3)
    dev = "cuda"
    for shapes, (T, B) in (([("rnn", 32, 16), ("lstm", 32, 16), ("lstm", 32, 16)], (50, 256)), ([("rnn", 32, 16), ("lstm", 32, 16)], (7, 37)),
                           ([("lstm", 32, 16), ("lstm", 32, 12), ("rnn", 32, 16)], (9, 40))):
        stacks = []
        for kind, i, h in shapes:
            cells = (RNN(i, h, "relu"), RNN(h, h, "relu")) if kind == "rnn" else (LSTM(i, h), LSTM(h, h))
            stacks.append(Recurrent(*cells, reverse=(len(stacks) % 2 == 0)).to(dev))
        n_in = max(i for _, i, _ in shapes)
        x = torch.randn(n_in, B, T, device=dev)
        gs = [torch.randn(m.sizes[-1], B, device=dev) for m in stacks]

        def run(grouped):
            xr = x.clone().requires_grad_(True)
            for m in stacks:
                m.theta.grad = None
            if grouped:
                buf = xr.permute(2, 1, 0).contiguous().float()
                ys = [y.t() for y in _RecurrentLaunchGroupFn.apply(tuple(stacks), buf, *[m.flat_weights() for m in stacks])]
            else:
                ys = [m(xr) for m in stacks]
            torch.autograd.backward(ys, gs)
            torch.cuda.synchronize()
            return [y.detach().clone() for y in ys], xr.grad.clone(), [m.theta.grad.clone() for m in stacks]

        a, b = run(False), run(True)
        for u, v in zip(a[0], b[0]):
            assert torch.equal(u, v)
        # the frame gradient is a sum of the stacks' contributions: the grouped node adds them in stack order, autograd in its own
        assert torch.allclose(a[1], b[1], rtol=0, atol=1e-6 * float(a[1].abs().max()))
        for u, v in zip(a[2], b[2]):
            assert torch.equal(u, v)


@pytest.mark.parametrize("cell", ["lstm", "rnn"])
def test_training_forward_keeps_the_records_for_the_pullback(cell):
    """lde_rnn_forward_train + lde_rnn_backward: the pullback back-propagates from the records and input panels the forward sweep left in
    the handle (mode 2 / mode 3 of k_rnn) instead of sweeping again — y, dx and dW equal the plain pair (lde_rnn_forward, then a
    pullback that sweeps itself) bit for bit; default shape (register-resident weights) and a run-time shape; a pullback for OTHER frames,
    or after a weight upload, does not use stale records."""
    import ctypes as C
    import torch
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd.recurrent import LSTM, RNN, Recurrent
    torch.manual_seed(9)
    dev = "cuda"
    for (i, h), (T, B) in (((32, 16), (50, 256)), ((20, 12), (9, 37))):
        cells = (LSTM(i, h), LSTM(h, h)) if cell == "lstm" else (RNN(i, h, "tanh"), RNN(h, h, "tanh"))
        m = Recurrent(*cells, reverse=True).to(dev)
        hnd, lib = m._native(), m._lib
        W = m.theta.detach().contiguous()
        s = L.raw_stream(0)
        p = lambda t: C.c_void_p(t.data_ptr())
        assert lib.lde_rnn_set_weights_device(hnd, p(W), W.numel(), s) == 0
        x = torch.randn(T, B, i, device=dev)
        x2 = torch.randn(T, B, i, device=dev)
        dy = torch.randn(B, h, device=dev)

        def pull(xx, train):
            y = torch.empty(B, h, device=dev)
            dx, dW = torch.empty_like(xx), torch.empty(m.num_weights, device=dev)
            fwd = lib.lde_rnn_forward_train if train else lib.lde_rnn_forward
            assert fwd(hnd, p(xx), T, B, p(y), s) == 0
            return y, dx, dW

        def back(xx, dx, dW):
            assert lib.lde_rnn_backward(hnd, p(xx), p(dy), T, B, p(dx), p(dW), s) == 0
            torch.cuda.synchronize()
            return dx.clone(), dW.clone()

        y0, dx0, dW0 = pull(x, False)
        dx0, dW0 = back(x, dx0, dW0)                     # the plain pair
        y1, dx1, dW1 = pull(x, True)
        dx1, dW1 = back(x, dx1, dW1)                     # records kept
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1) and torch.equal(dW0, dW1)
        # records of x, pullback asked for x2: swept again on x2
        yr, dxr, dWr = pull(x2, False)
        dxr, dWr = back(x2, dxr, dWr)
        _ = pull(x, True)
        dxs, dWs = back(x2, torch.empty_like(x2), torch.empty(m.num_weights, device=dev))
        assert torch.equal(dxr, dxs) and torch.equal(dWr, dWs)
        # a weight upload between the two drops the records
        _ = pull(x, True)
        W2 = (W * 1.01).contiguous()
        assert lib.lde_rnn_set_weights_device(hnd, p(W2), W2.numel(), s) == 0
        dxa, dWa = back(x, torch.empty_like(x), torch.empty(m.num_weights, device=dev))
        _ = pull(x, False)
        dxb, dWb = back(x, torch.empty_like(x), torch.empty(m.num_weights, device=dev))
        assert torch.equal(dxa, dxb) and torch.equal(dWa, dWb)

