"""GPU: the variational sample and the loss terms (lde_sample_* / lde_kl_* / lde_mse_*, scope row f-3) through the C ABI,
against the f64 CPU oracle (oracle/lde_loss_oracle.c) on the same seeded inputs.

Tolerances (f32 kernels vs the f64 oracle): elementwise outputs 2e-6 relative to the largest entry (one v_exp_f32 and a few
roundings); the reductions 1e-5 relative (f32 partial sums in a fixed tree; the terms are non-negative)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

E_ELEM, E_SUM = 2e-6, 1e-5


@pytest.fixture(scope="module")
def o64():
    return O.Oracle("f64")


@pytest.fixture(scope="module")
def nat():
    from tests.gpu_util import NativeLoss
    return NativeLoss()


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


# latent sizes of the GOKU default (16 × B), ragged sizes, one past / one short of the 16-byte and workgroup boundaries,
# and an operand offset of one float (unaligned pointers take the scalar path)
SIZES = [(1, 0), (3, 0), (16 * 256, 0), (16 * 256, 1), (4097, 0), (8191, 3), (8192, 0), (100003, 0), (100003, 2)]


@pytest.mark.parametrize("n,offset", SIZES)
def test_sample_and_its_pullback(nat, o64, n, offset):
    rng = np.random.default_rng(n + offset)
    mu, ls, eps, dl = (rng.standard_normal(n).astype(np.float32) for _ in range(4))
    want = o64.sample_forward(mu, ls, eps)
    assert _rel(nat.sample_forward(mu, ls, eps, offset), want) <= E_ELEM
    _, dlv = o64.sample_backward(ls, eps, dl)
    assert _rel(nat.sample_backward(ls, eps, dl, offset), dlv) <= E_ELEM


@pytest.mark.parametrize("n,offset", SIZES)
def test_kl_and_its_pullback(nat, o64, n, offset):
    rng = np.random.default_rng(7 * n + offset)
    mu, ls = rng.standard_normal(n).astype(np.float32), (0.7 * rng.standard_normal(n)).astype(np.float32)
    B = 13
    want = o64.kl_forward(mu, ls, 1 / B)
    got = nat.kl_forward(mu, ls, 1 / B, offset)
    assert abs(got - want) <= E_SUM * abs(want), (got, want)
    dmu, dlv = o64.kl_backward(mu, ls, 1 / B, 1.7)
    gm, gl = nat.kl_backward(mu, ls, 1 / B, 1.7, offset)
    assert _rel(gm, dmu) <= E_ELEM and _rel(gl, dlv) <= E_ELEM


@pytest.mark.parametrize("n,offset", SIZES)
def test_mse_and_its_pullback(nat, o64, n, offset):
    rng = np.random.default_rng(11 * n + offset)
    x, xh = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    scale = 1 / 50.0
    want = o64.mse_forward(x, xh, scale)
    got = nat.mse_forward(x, xh, scale, offset)
    assert abs(got - want) <= E_SUM * abs(want), (got, want)
    assert _rel(nat.mse_backward(x, xh, scale, 0.3, offset), o64.mse_backward(x, xh, scale, 0.3)) <= E_ELEM


@pytest.mark.parametrize("n,offset", SIZES)
def test_sample_with_kl_in_one_pass(nat, o64, n, offset):
    """lde_sample_kl_forward / _backward and lde_mse_forward_add = the separate entry points composed as loss_batch composes
    them: against the oracle's separate terms, and bit for bit against the separate kernels where the arithmetic is the same
    (the sample, both reductions before `base` is added, and the summed cotangents)."""
    rng = np.random.default_rng(13 * n + offset)
    mu, eps, dl = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    ls = (0.7 * rng.standard_normal(n)).astype(np.float32)
    scale, base, g = 1e-3 / 13, 0.625, 1.7
    l, tot = nat.sample_kl_forward(mu, ls, eps, scale, base, offset)
    assert np.array_equal(l, nat.sample_forward(mu, ls, eps, offset))
    assert _rel(l, o64.sample_forward(mu, ls, eps)) <= E_ELEM
    kl = nat.kl_forward(mu, ls, scale, offset)
    assert tot == float(np.float32(base) + np.float32(kl))
    assert abs((tot - base) - o64.kl_forward(mu, ls, scale)) <= E_SUM * abs(o64.kl_forward(mu, ls, scale)) + 1e-7
    _, tot0 = nat.sample_kl_forward(mu, ls, eps, scale, None, offset)
    assert tot0 == kl
    dm, dv = nat.sample_kl_backward(mu, ls, eps, dl, g, scale, offset)
    km, kv = nat.kl_backward(mu, ls, scale, g, offset)
    sv = nat.sample_backward(ls, eps, dl, offset)
    assert np.array_equal(dm, dl + km) and np.array_equal(dv, sv + kv)
    om, ov = o64.kl_backward(mu, ls, scale, g)
    _, osv = o64.sample_backward(ls, eps, dl)
    assert _rel(dm, dl.astype(np.float64) + om) <= E_ELEM and _rel(dv, osv + ov) <= E_ELEM
    x, xh = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    assert nat.mse_forward_add(x, xh, 1 / 50.0, base, offset) == float(np.float32(base) + np.float32(nat.mse_forward(x, xh, 1 / 50.0, offset)))


def test_loss_batch_fused_equals_the_separate_terms():
    """train.loss_batch with the sample and β·KL in one pass (default) vs the reference's composition of sample / vector_kl /
    reconstruction_loss and torch additions (train._FUSED_LOSS = False): same ε (same generator state), same loss to f32 rounding of the
    scalar additions, same gradients to 1e-6 of their largest entry."""
    import torch
    import latentdiffeq_amd as la
    from latentdiffeq_amd import train as TR
    torch.manual_seed(3)
    mt, diffeq = la.GOKU_basic(), la.Pendulum()
    enc, dec = TR.default_layers(mt, 64, diffeq, device="cuda", hidden_dim_resnet=48)
    with torch.no_grad():
        dec[0][1]._dense[-1].bias.fill_(1.0)
    model = TR.LatentDiffEqModel(mt, enc, dec)
    B, T = 24, 12
    ts = np.arange(T) * 0.05
    x = torch.rand(T, B, 64, device="cuda").permute(2, 1, 0)
    res = []
    for fused in (True, False):
        TR._FUSED_LOSS = fused
        try:
            for p in model.parameters():
                p.grad = None
            torch.manual_seed(11)
            loss = TR.loss_batch(model, x, ts, 0.5, True)
            loss.backward()
            res.append((float(loss.detach()), [p.grad.detach().clone() for p in model.parameters()]))
        finally:
            TR._FUSED_LOSS = True
    (lf, gf), (ls_, gs) = res
    assert abs(lf - ls_) <= 2e-6 * abs(ls_), (lf, ls_)
    for a, b in zip(gf, gs):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12


def test_known_answers_and_edges(nat):
    z = np.zeros(64, np.float32)
    assert nat.kl_forward(z, z, 1.0) == 0.0                      # kl(0, 0) = 0 exactly
    x = np.linspace(0, 1, 1000).astype(np.float32)
    assert nat.mse_forward(x, x, 1.0) == 0.0
    assert np.array_equal(nat.sample_forward(x, x, np.zeros_like(x)), x)    # ε = 0 ⇒ l̃ = μ, bit for bit
    e = np.zeros(0, np.float32)
    assert nat.kl_forward(e, e, 1.0) == 0.0 and nat.mse_forward(e, e, 1.0) == 0.0   # empty input: the sum is 0
    assert nat.sample_forward(e, e, e).size == 0


def test_reductions_are_reproducible_and_full_size(nat, o64):
    """The reconstruction loss at the size of the training step (784 × 256 × 50 ≈ 10 M terms): two calls agree bit for bit
    (fixed summation order, no atomics), and the value matches the oracle."""
    rng = np.random.default_rng(3)
    n = 784 * 256 * 50
    x, xh = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    a = nat.mse_forward(x, xh, 1 / (256 * 50))
    b = nat.mse_forward(x, xh, 1 / (256 * 50))
    assert a == b
    want = o64.mse_forward(x, xh, 1 / (256 * 50))
    assert abs(a - want) <= E_SUM * want, (a, want)


def test_torch_level_functions_match_the_oracle(o64):
    """loss.sample / vector_kl / reconstruction_loss on the reference-shaped (transposed) arrays, gradients through autograd."""
    import torch
    from latentdiffeq_amd import loss as LS
    torch.manual_seed(5)
    B, T, P = 40, 7, 12
    mu_b = torch.randn(B, 16, device="cuda", requires_grad=True)                 # batch-major buffers, as the chains return them
    ls_b = (0.5 * torch.randn(B, 16, device="cuda")).requires_grad_(True)
    mu, ls = mu_b.t(), ls_b.t()                                                  # [16, B] views: the reference's layout
    k = LS.vector_kl((mu, mu), (ls, ls))
    k.backward()
    m64, l64 = mu_b.detach().cpu().numpy().astype(np.float64), ls_b.detach().cpu().numpy().astype(np.float64)
    want = 2 * o64.kl_forward(m64, l64, 1 / B)
    assert abs(float(k) - want) <= E_SUM * want
    dmu, dlv = o64.kl_backward(m64, l64, 1 / B, 2.0)
    assert _rel(mu_b.grad.cpu().numpy(), dmu) <= E_ELEM and _rel(ls_b.grad.cpu().numpy(), dlv) <= E_ELEM
    mu_b.grad = ls_b.grad = None
    torch.manual_seed(9)
    l = LS.sample(mu, ls)
    assert l.shape == mu.shape
    torch.manual_seed(9)
    eps = LS.randn((B, 16), "cuda")                                               # the draw sample() made (same generator state)
    want_l = o64.sample_forward(m64, l64, eps.cpu().numpy().astype(np.float64))
    assert _rel(l.detach().t().cpu().numpy(), want_l) <= E_ELEM
    ct = torch.randn_like(l)
    (l * ct).sum().backward()
    _, dlv = o64.sample_backward(l64, eps.cpu().numpy().astype(np.float64), ct.t().cpu().numpy().astype(np.float64))
    assert _rel(ls_b.grad.cpu().numpy(), dlv) <= E_ELEM
    assert torch.equal(mu_b.grad, ct.t())
    x = torch.rand(T, B, P, device="cuda").permute(2, 1, 0)                       # [pixels, B, T], pixels fastest
    xh_b = torch.rand(T, B, P, device="cuda", requires_grad=True)
    r = LS.reconstruction_loss(x, xh_b.permute(2, 1, 0))
    r.backward()
    x64, h64 = x.permute(2, 1, 0).cpu().numpy().astype(np.float64), xh_b.detach().cpu().numpy().astype(np.float64)
    want = o64.mse_forward(x64, h64, 1 / (B * T))
    assert abs(float(r) - want) <= E_SUM * want
    assert np.isclose(want, ((x64 - h64) ** 2).transpose(2, 1, 0).mean(axis=(1, 2)).sum(), rtol=1e-12)   # sum(mean(·, dims=(2,3)))
    assert _rel(xh_b.grad.cpu().numpy(), o64.mse_backward(x64, h64, 1 / (B * T), 1.0)) <= E_ELEM
    # one rank's shard of a global batch of 4·B: the per-rank terms are the reference's divided by 4 (they add up over ranks)
    with torch.no_grad():
        assert abs(float(LS.reconstruction_loss(x, xh_b.permute(2, 1, 0), 4 * B)) - want / 4) <= E_SUM * want
        k1, k4 = float(LS.vector_kl(mu, ls)), float(LS.vector_kl(mu, ls, 4 * B))
        assert abs(k4 - k1 / 4) <= E_SUM * abs(k1)


@pytest.mark.parametrize("B", [16, 17])
def test_sample_when_batch_equals_latent_dim(o64, B):
    """B == latent dim (16, the default latent_dim_z0 / latent_dim_theta): the [16, 16] transposed views permute without
    changing shape — the result must still come back in the caller's layout (round-1 advisor finding)."""
    import torch
    from latentdiffeq_amd import loss as LS
    torch.manual_seed(3)
    mu_b = torch.randn(B, 16, device="cuda", requires_grad=True)
    ls_b = (0.5 * torch.randn(B, 16, device="cuda")).requires_grad_(True)
    mu, ls = mu_b.t(), ls_b.t()                                                   # [16, B] non-contiguous views
    torch.manual_seed(11)
    l = LS.sample(mu, ls)
    torch.manual_seed(11)
    eps = LS.randn((B, 16), "cuda")
    want = o64.sample_forward(mu_b.detach().cpu().numpy().astype(np.float64), ls_b.detach().cpu().numpy().astype(np.float64),
                              eps.cpu().numpy().astype(np.float64))
    assert l.shape == (16, B) and _rel(l.detach().t().cpu().numpy(), want) <= E_ELEM
    ct = torch.randn(16, B, device="cuda")
    (l * ct).sum().backward()
    assert torch.equal(mu_b.grad, ct.t())


def test_loss_errors_are_reported_not_thrown():
    """NULL operands and negative sizes come back as LDE_ERR_INVALID_ARG; nothing is launched, nothing crashes."""
    import ctypes as C
    import torch
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    t = torch.zeros(64, device="cuda")
    p, null, s = C.c_void_p(t.data_ptr()), C.c_void_p(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = L.STATUS_CODE["LDE_ERR_INVALID_ARG"] if hasattr(L, "STATUS_CODE") else -1
    assert lib.lde_sample_forward(null, p, p, 64, p, s) == bad
    assert lib.lde_sample_backward(p, p, null, 64, p, s) == bad
    assert lib.lde_kl_forward(p, p, 64, 1.0, null, p, s) == bad
    assert lib.lde_kl_forward(p, null, 64, 1.0, p, p, s) == bad
    assert lib.lde_kl_forward(p, p, -1, 1.0, p, p, s) == bad
    assert lib.lde_kl_backward(p, p, 64, 1.0, null, p, p, s) == bad
    assert lib.lde_mse_forward(p, p, 64, 1.0, p, null, s) == bad
    assert lib.lde_mse_backward(p, p, 64, 1.0, p, null, s) == bad
    torch.cuda.synchronize()
    assert float(t.abs().max()) == 0.0


# ---- ε: lde_randn / loss.randn (round 3) ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,seed,offset,call,epoch", [(1, 0, 0, 0, None), (4096, 0x0123456789ABCDEF, 12, 0, None), (4099, 77, (1 << 40) + 8, 3, 5),
                                                      (10, (1 << 64) - 1, (1 << 64) - 2, (1 << 32) - 1, 9)])
def test_randn_is_philox_and_box_muller(n, seed, offset, call, epoch):
    """The raw words equal the numpy restatement of Philox4x32-10 (tests/philox_ref.py, pinned by the generator's known-answer vectors in
    tests/test_philox.py) bit for bit — counter (block, call, offset + *epoch), key = seed — and the normals are Box–Muller of them
    (f32 logf / sincospif against f64: 2e-6 of the value's scale)."""
    import ctypes as C
    import torch
    from latentdiffeq_amd import _lib as L
    from tests import philox_ref as P
    lib = L.load()
    out = torch.empty(n, device="cuda")
    raw = torch.empty(n, device="cuda", dtype=torch.int32)
    ep = torch.tensor(epoch, device="cuda", dtype=torch.int64) if epoch is not None else None
    rc = lib.lde_randn(C.c_void_p(out.data_ptr()), n, seed, offset, call, C.c_void_p(ep.data_ptr()) if ep is not None else None,
                       C.c_void_p(raw.data_ptr()), L.raw_stream())
    assert rc == 0
    torch.cuda.synchronize()
    want = P.words(n, seed, offset, call, epoch or 0)
    assert np.array_equal(raw.cpu().numpy().view(np.uint32), want)
    z = P.normals(P.words(4 * ((n + 3) // 4), seed, offset, call, epoch or 0))[:n]     # (whole blocks: a pair's second word exists past n)
    assert np.abs(out.cpu().numpy() - z).max() <= 2e-6 * max(1.0, np.abs(z).max())


def test_randn_moments_and_generator_semantics():
    """loss.randn: standard normal moments over 2²⁰ draws; torch.manual_seed reproduces a sequence and consecutive draws differ (the
    draw advances torch's CUDA generator like a torch draw of its size, with no launch of its own); inside a stream capture the draw
    is keyed by the registered device counter — a replay after the counter moved gives new noise, a replay at the same count the same."""
    import torch
    from latentdiffeq_amd import loss as LS
    torch.manual_seed(5)
    z = LS.randn((1 << 20,), "cuda")
    m, v, k = float(z.mean()), float(z.var()), float((z ** 4).mean())
    assert abs(m) < 4e-3 and abs(v - 1) < 6e-3 and abs(k - 3) < 0.05, (m, v, k)
    torch.manual_seed(9)
    a, b = LS.randn((16, 256), "cuda"), LS.randn((16, 256), "cuda")
    t = torch.randn(8, device="cuda")                       # a torch draw in between sees an advanced generator too
    torch.manual_seed(9)
    a2, b2 = LS.randn((16, 256), "cuda"), LS.randn((16, 256), "cuda")
    t2 = torch.randn(8, device="cuda")
    assert torch.equal(a, a2) and torch.equal(b, b2) and torch.equal(t, t2) and not torch.equal(a, b)
    keep = dict(LS._noise_epoch)
    try:
        cnt = torch.zeros((), device="cuda", dtype=torch.int64)
        LS.set_noise_epoch(cnt)
        buf = torch.zeros(2, 1000, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                buf[0].copy_(LS.randn((1000,), "cuda"))
                buf[1].copy_(LS.randn((1000,), "cuda"))
        torch.cuda.current_stream().wait_stream(s)
        g.replay(); torch.cuda.synchronize(); r0 = buf.clone()
        g.replay(); torch.cuda.synchronize(); r0b = buf.clone()
        cnt += 1
        g.replay(); torch.cuda.synchronize(); r1 = buf.clone()
        assert torch.equal(r0, r0b) and not torch.equal(r0, r1) and not torch.equal(r0[0], r0[1])
        assert abs(float(r1.mean())) < 0.1 and abs(float(r1.var()) - 1) < 0.15
    finally:
        LS._noise_epoch.clear()
        LS._noise_epoch.update(keep)


@pytest.mark.parametrize("B", [1, 37, 256, 512])
def test_sample_pair_equals_the_separate_calls(B):
    """sample_with_kl of the GOKU tuple: one launch each way (lde_sample_kl_pair_*: both parts' ε, samples and the running KL total) against
    the separate calls (randn, sample+KL, randn, sample+KL; two pullbacks) from the same generator state — samples, total and every
    gradient bit for bit; the generator ends in the same state."""
    import torch
    from latentdiffeq_amd import loss as LS
    torch.manual_seed(12)
    raw = [torch.randn(B, 16, device="cuda") for _ in range(4)]
    cts = [torch.randn(16, B, device="cuda") for _ in range(2)]
    res = []
    keep = LS._SAMPLE_PAIR
    try:
        for pair in (True, False):
            LS._SAMPLE_PAIR = pair
            leaves = [t.clone().requires_grad_(True) for t in raw]
            mu, ls = (leaves[0].t(), leaves[1].t()), (0.5 * leaves[2].t(), 0.5 * leaves[3].t())
            torch.manual_seed(21)
            (la, lb), total = LS.sample_with_kl(mu, ls, 0.7, 4 * B)
            after = torch.randn(4, device="cuda")
            ((la * cts[0]).sum() + (lb * cts[1]).sum() + 1.3 * total).backward()
            torch.cuda.synchronize()
            res.append([la.detach().clone(), lb.detach().clone(), total.detach().clone(), after] + [t.grad.clone() for t in leaves])
    finally:
        LS._SAMPLE_PAIR = keep
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), (B, i)
    assert res[0][0].shape == (16, B)
