"""GPU: the dense chains in bf16 mode (lde_chain_set_dtype; BASELINE.json configs[4] "mixed fp32 solve / bf16
encoder-decoder"; round 3: native — csrc/lde_chain_bf16.h): both operands of every matrix product — forward, input gradient,
weight gradient — are bfloat16 (round-to-nearest-even) multiplied with f32 accumulation on v_mfma_f32_16x16x32_bf16 /
32x32x16_bf16; hidden activations and δ are stored in bf16; master weights, biases, y, dx and dW stay f32.

Checked against a numpy restatement of exactly that arithmetic (Flux's Dense / SkipConnection definitions as in
oracle/lde_chain_oracle.c, operands rounded to bf16 with integer arithmetic, products accumulated in float64):
agreement to f32 accumulation order (median ≤ 2e-6, worst entry ≤ 2e-3 of the output scale — a hidden value sitting on a rounding
boundary may round the other way), and against the f32 path at the bf16 level (≤ 3e-2)."""
import numpy as np
import pytest

from latentdiffeq_amd import _lib as L

pytestmark = pytest.mark.gpu


def bf16r(a):
    """float32 → nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def _act(kind, x):
    if kind == L.CACT_RELU:
        return np.maximum(x, 0)
    if kind == L.CACT_TANH:
        return np.tanh(x)
    if kind == L.CACT_SIGMOID:
        return 1 / (1 + np.exp(-x))
    if kind == L.CACT_SOFTPLUS:
        return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))
    return x


def _act_grad_out(kind, f):
    if kind == L.CACT_RELU:
        return (f > 0).astype(f.dtype)
    if kind == L.CACT_TANH:
        return 1 - f * f
    if kind == L.CACT_SIGMOID:
        return f * (1 - f)
    if kind == L.CACT_SOFTPLUS:
        return 1 - np.exp(-f)
    return np.ones_like(f)


def _split(sizes, W):
    out, off = [], 0
    for i, o in zip(sizes[:-1], sizes[1:]):
        Wl = W[off:off + i * o].reshape(i, o).T          # vec(W) column-major [out×in]
        off += i * o
        out.append((Wl, W[off:off + o]))
        off += o
    return out


def chain_ref(sizes, acts, skips, W, x, dy, rnd):
    """x (N, in), dy (N, out) → y, dx, dW of the mode `rnd` defines (identity: the f32 chain; bf16r: the native bf16 mode of
    csrc/lde_chain_bf16.h): both operands of every product are `rnd`-ed, and the hidden activations, the activation before a skip
    addition and δ are STORED `rnd`-ed; biases, the gradient that flows around a skip connection, y, dx and dW are f32."""
    layers = _split(sizes, W.astype(np.float64))
    L = len(layers)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    h = [x.astype(np.float64).T]
    f = []
    for l, ((Wl, bl), a, s) in enumerate(zip(layers, acts, skips)):
        pre = rnd(Wl).astype(np.float64) @ rnd(h[-1]).astype(np.float64) + bl[:, None]
        fl = f32(_act(a, pre))                                        # the epilogue's f32 value
        if l == L - 1:
            f.append(fl)
            h.append(fl)                                               # y stays f32
        else:
            f.append(rnd(fl).astype(np.float64))                       # stored: what the activation derivative is taken from
            h.append(rnd(f32(rnd(h[-1]).astype(np.float64) + fl) if s else fl).astype(np.float64))
    G = dy.astype(np.float64).T
    dW = []
    for l in range(L - 1, -1, -1):
        Wl, _ = layers[l]
        d = rnd(f32(G * _act_grad_out(acts[l], f[l]))).astype(np.float64)   # δ_l as stored
        gW = d @ rnd(h[l]).astype(np.float64).T
        dW = [gW.T.reshape(-1), d.sum(axis=1)] + dW
        G = rnd(Wl.T).astype(np.float64) @ d + (G if skips[l] else 0)
    return h[-1].T, G.T, np.concatenate(dW)


SPECS = {
    "reconstructor": ((2, 200, 200, 200, 784), (L.CACT_RELU, L.CACT_RELU, L.CACT_RELU, L.CACT_SIGMOID), (0, 1, 1, 0)),
    "feature_extractor": ((784, 200, 200, 200, 32), (L.CACT_RELU, L.CACT_RELU, L.CACT_RELU, L.CACT_RELU), (0, 1, 1, 0)),
    "latent_out": ((16, 200, 1), (L.CACT_RELU, L.CACT_SOFTPLUS), (0, 0)),
    "odd_tanh": ((7, 33, 50, 21), (L.CACT_TANH, L.CACT_TANH, L.CACT_IDENTITY), (0, 0, 0)),
}


@pytest.mark.parametrize("spec", sorted(SPECS))
@pytest.mark.parametrize("N", [40, 1000])
def test_bf16_chain_matches_the_rounded_operand_restatement(spec, N):
    from tests.gpu_util import NativeChain
    from latentdiffeq_amd import synthetic as S
    sizes, acts, skips = SPECS[spec]
    rng = np.random.default_rng(7)
    W = S.mlp_weights(sizes, seed=11)
    x = (rng.uniform(0, 1, (N, sizes[0])) if sizes[0] == 784 else 0.6 * rng.standard_normal((N, sizes[0]))).astype(np.float32)
    dy = (rng.standard_normal((N, sizes[-1])) / N).astype(np.float32)
    ch = NativeChain(sizes, acts, skips)
    ch.set_weights(W)
    y32 = ch.forward(x)
    dx32, dW32 = ch.backward(x, y32, dy)
    ch.set_dtype("bf16")
    y = ch.forward(x)
    dx, dW = ch.backward(x, y, dy)
    ys, saved = ch.forward_save(x)
    dxs, dWs = ch.backward_saved(x, ys, dy, saved)
    assert np.array_equal(y, ys) and np.array_equal(dx, dxs) and np.array_equal(dW, dWs)      # training variant: same numbers
    yr, dxr, dWr = chain_ref(sizes, acts, skips, W, x, dy, bf16r)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    # most entries agree to 7 digits; a hidden value that sits on a bf16 rounding boundary rounds the other way when the f32
    # accumulation order differs in its last bit, and moves what follows by one bf16 ulp of one term: ≤ 2e-3 of the scale over
    # four layers of K = 200 … 784 (measured 9e-4 on the 784-200-200-200-32 chain at N = 1000, 1e-5 … 2e-4 elsewhere)
    assert rel(y, yr) <= 2e-3, rel(y, yr)
    assert np.median(np.abs(y - yr)) <= 2e-6 * max(np.abs(yr).max(), 1e-30)
    assert rel(dx, dxr) <= 5e-3 and rel(dW, dWr) <= 5e-3, (rel(dx, dxr), rel(dW, dWr))
    # bf16 is bf16: the mode moves the results by its rounding (2⁻⁹ per operand), no more — and it is not the f32 path
    l2 = lambda a, b: np.linalg.norm((a - b).astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)
    assert 1e-6 < rel(y, y32) <= 3e-2 and l2(dx, dx32) <= 0.3 and l2(dW, dW32) <= 0.3   # (sanity only: relu kinks flip for a few units, gradients move more than values; the parity is the restatement above)
    ch.set_dtype("f32")
    assert np.array_equal(ch.forward(x), y32)                                                  # and back: nothing was re-uploaded


def test_set_dtype_validates():
    from tests.gpu_util import NativeChain
    ch = NativeChain((4, 8, 2), (L.CACT_RELU, L.CACT_IDENTITY))
    assert ch.lib.lde_chain_set_dtype(ch.h, 7) == -1 and ch.lib.lde_chain_set_dtype(None, 0) == -1


@pytest.mark.parametrize("N", [12800, 1000])
def test_reconstructor_pullback_from_the_delta_the_forward_pass_left(N):
    """lde_chain_forward_save_mse_delta / lde_chain_backward_saved_delta (bf16 chains): the forward launch's last epilogue leaves
    δ_L′ = 2·scale·(x̂ − x)·σ′(x̂) as the pullback's bf16 δ matrix, the pullback starts there (no pass over x̂ and the frames) and multiplies
    dx / dW by the loss's cotangent g at the end. Against lde_chain_forward_save_mse + lde_chain_backward_saved_mse on the same chain
    [REF src/models/GOKU.jl:252-269], [REF examples/pendulum_friction-less/model_train.jl:225-238]: with g = 1 — the loss is the
    objective — the SAME BITS (loss, dx, dW; and x̂ when it is asked for); with g ≠ 1 the results are g times those of g = 1 exactly
    (δ is linear in g: the rounded-operand arithmetic is that of g = 1), which is the two-call path's to bf16 rounding of δ."""
    import ctypes as C
    import torch
    from tests.gpu_util import NativeChain
    from latentdiffeq_amd import synthetic as S
    sizes, acts, skips = (2, 200, 200, 200, 784), (L.CACT_RELU, L.CACT_RELU, L.CACT_RELU, L.CACT_SIGMOID), (0, 1, 1, 0)
    rng = np.random.default_rng(3)
    W = S.mlp_weights(sizes, seed=11)
    x = torch.from_numpy((0.7 * rng.standard_normal((N, 2))).astype(np.float32)).cuda()
    tgt = torch.from_numpy(rng.uniform(0, 1, (N, 784)).astype(np.float32)).cuda()
    base = torch.tensor([0.25], device="cuda")
    ch = NativeChain(sizes, acts, skips)
    ch.set_weights(W)
    ch.set_dtype("bf16")
    lib, h = ch.lib, ch.h
    lib.lde_chain_saved_floats.restype = C.c_int64
    lib.lde_chain_mse_scratch_floats.restype = C.c_int64
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / N
    nsv, nws = int(lib.lde_chain_saved_floats(h, N)), int(lib.lde_chain_mse_scratch_floats(h, N)) + 1

    def two_calls(g):
        y, saved, ws = torch.empty((N, 784), device="cuda"), torch.empty((nsv,), device="cuda"), torch.empty((nws,), device="cuda")
        L.check(lib.lde_chain_forward_save_mse(h, p(x), N, p(y), p(saved), p(tgt), scale, p(base), p(ws), C.c_void_p(ws.data_ptr() + 4), s), h, "fwd", chain=True)
        dx, dW, gd = torch.empty_like(x), torch.zeros((ch.nW,), device="cuda"), torch.tensor([g], device="cuda")
        L.check(lib.lde_chain_backward_saved_mse(h, p(x), p(y), p(tgt), p(gd), scale, C.c_void_p(), p(saved), N, p(dx), p(dW), s), h, "bwd", chain=True)
        torch.cuda.synchronize()
        return ws[0].item(), y.cpu().numpy(), dx.cpu().numpy(), dW.cpu().numpy()

    def delta(g, want_y):
        y = torch.full((N, 784), 7.0, device="cuda") if want_y else None
        saved, ws = torch.empty((nsv,), device="cuda"), torch.empty((nws,), device="cuda")
        L.check(lib.lde_chain_forward_save_mse_delta(h, p(x), N, p(y), p(saved), p(tgt), scale, p(base), p(ws), C.c_void_p(ws.data_ptr() + 4), s), h, "fwd", chain=True)
        dx, dW, gd = torch.empty_like(x), torch.zeros((ch.nW,), device="cuda"), torch.tensor([g], device="cuda")
        L.check(lib.lde_chain_backward_saved_delta(h, p(x), p(gd), p(saved), N, p(dx), p(dW), s), h, "bwd", chain=True)
        torch.cuda.synchronize()
        return ws[0].item(), (y.cpu().numpy() if want_y else None), dx.cpu().numpy(), dW.cpu().numpy()

    l0, y0, dx0, dW0 = two_calls(1.0)
    l1, y1, dx1, dW1 = delta(1.0, True)
    l2, y2, dx2, dW2 = delta(1.0, False)
    assert l0 == l1 == l2 and np.array_equal(y0, y1) and y2 is None
    relm = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert np.array_equal(dx0, dx1) and np.array_equal(dW0, dW1) and np.array_equal(dx0, dx2) and np.array_equal(dW0, dW2), \
        (relm(dx1, dx0), relm(dW1, dW0), relm(dx2, dx0), relm(dW2, dW0))
    g = 0.5                                                          # (a power of two: g·v is exact, so "g times the g = 1 result" is a bit statement)
    _, _, dxg, dWg = delta(g, False)
    assert np.array_equal(dxg, g * dx0) and np.array_equal(dWg, g * dW0)
    _, _, dxr, dWr = two_calls(g)                                    # the two-call path rounds g·δ instead of δ: equal to bf16 rounding of δ
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(dxg, dxr) <= 2e-2 and rel(dWg, dWr) <= 2e-2 and np.linalg.norm(dWg - dWr) <= 4e-3 * np.linalg.norm(dWr)
    # a pullback without a staged δ of this size is refused, not guessed
    gd = torch.tensor([1.0], device="cuda")
    sv = torch.empty((nsv,), device="cuda")
    dW = torch.zeros((ch.nW,), device="cuda")
    assert lib.lde_chain_backward_saved_delta(h, p(x), p(gd), p(sv), N - 16, C.c_void_p(), p(dW), s) != 0
    ch.set_dtype("f32")
    y = torch.empty((N, 784), device="cuda")
    ws = torch.empty((nws,), device="cuda")
    assert lib.lde_chain_forward_save_mse_delta(h, p(x), N, p(y), p(sv), p(tgt), scale, p(base), p(ws), C.c_void_p(ws.data_ptr() + 4), s) == -2   # LDE_ERR_UNSUPPORTED


def test_two_forwards_of_one_decoder_before_their_backwards():
    """the bf16 δ-staging lives in the chain's single workspace. Two `decode_loss`-style forwards of the SAME chain (same N)
    before their pullbacks: the first graph must not run its pullback from the second forward's δ_L′. The staging is tokenised by the
    forward call's saved-activation buffer (lde_chain_delta_is_staged): the second forward's pullback takes the staged δ, the first one's
    falls back to the two-pass pullback from x̂ and the frames — both give the gradients of their OWN forward; the C entry point refuses a
    stale token instead of guessing. [REF src/models/GOKU.jl:252-269], [REF examples/pendulum_friction-less/model_train.jl:225-238]"""
    import ctypes as C
    import torch
    from latentdiffeq_amd import chain as CH
    from latentdiffeq_amd import synthetic as S
    sizes = (2, 200, 200, 200, 784)
    N = 1024
    rng = np.random.default_rng(5)
    dec = CH.Chain(CH.Dense(2, 200, "relu"), CH.SkipConnection(CH.Dense(200, 200, "relu")), CH.SkipConnection(CH.Dense(200, 200, "relu")),
                   CH.Dense(200, 784, "sigmoid")).cuda().set_dtype("bf16")
    assert tuple(dec.sizes) == sizes
    xs = [torch.from_numpy((0.7 * rng.standard_normal((N, 2))).astype(np.float32)).cuda().requires_grad_() for _ in range(2)]
    tg = [torch.from_numpy(rng.uniform(0, 1, (N, 784)).astype(np.float32)).cuda() for _ in range(2)]

    def grads(order):
        for x in xs:
            x.grad = None
        dec.theta.grad = None
        losses = [CH._ChainMseFn.apply(dec, xs[i], dec.theta, tg[i], 1.0 / N, None, True)[0] for i in range(2)]
        out = {}
        for i in order:
            dec.theta.grad = None
            losses[i].backward()
            out[i] = (xs[i].grad.clone(), dec.theta.grad.clone())
        return out

    # reference: each forward followed at once by its own backward
    ref = {}
    for i in range(2):
        xs[i].grad = None
        dec.theta.grad = None
        CH._ChainMseFn.apply(dec, xs[i], dec.theta, tg[i], 1.0 / N, None, True)[0].backward()
        ref[i] = (xs[i].grad.clone(), dec.theta.grad.clone())
    for order in ((0, 1), (1, 0)):
        got = grads(order)
        for i in range(2):
            for a, b in zip(got[i], ref[i]):
                rel = float((a - b).abs().max() / b.abs().max())
                assert rel <= 2e-2, (order, i, rel)           # (the fall-back rounds g·δ instead of δ: equal to bf16 rounding; a stale δ would be O(1) off)
    # and the C entry point: a stale token is refused
    h, lib = dec._native(), dec._lib
    sv = torch.empty((int(lib.lde_chain_saved_floats(h, N)),), device="cuda")
    gd = torch.tensor([1.0], device="cuda")
    dW = torch.zeros((dec.num_weights,), device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.lde_chain_delta_is_staged(h, C.c_void_p(sv.data_ptr()), N) == 0
    assert lib.lde_chain_backward_saved_delta(h, C.c_void_p(xs[0].data_ptr()), C.c_void_p(gd.data_ptr()), C.c_void_p(sv.data_ptr()), N, C.c_void_p(),
                                              C.c_void_p(dW.data_ptr()), s) != 0
