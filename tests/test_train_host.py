"""CPU: the host-side training harness (scope row f-3) against values worked out by hand from the reference's code
[REF src/utils/utils.jl], [REF examples/pendulum_friction-less/model_train.jl:225-238]."""
import numpy as np
import torch

from latentdiffeq_amd import train as TR


def julia_frange(n_iter, start, stop, n_cycle, ratio):
    """The reference loop with 1-based indices, written independently of train.py (dict as a 1-based array)."""
    L = {i: stop for i in range(1, n_iter + 1)}
    period = n_iter / n_cycle
    step = np.float32((stop - start) / (period * ratio))
    for c in range(0, n_cycle):
        v, i = np.float32(start), 1
        while (v <= stop) and (int(round(i + c * period)) < n_iter):
            L[int(round(i + c * period))] = v
            v = np.float32(v + step)
            i += 1
    return np.array([L[i] for i in range(1, n_iter + 1)], dtype=np.float32)


def test_frange_cycle_linear_matches_the_reference_loop():
    for args in [(900, 0.0, 1.0, 3, 0.9), (10, 0.0, 1.0, 4, 0.5), (37, 0.0, 0.5, 5, 0.3), (8, 0.0, 1.0, 3, 0.9)]:
        got = TR.frange_cycle_linear(*args)
        assert got.dtype == np.float32 and np.array_equal(got, julia_frange(*args)), args
    s = TR.frange_cycle_linear(12, 0.0, 1.0, 3, 0.5)          # period 4, ramp over 2 steps: 0, .5, 1, 1 | 0, .5, 1, 1 | 0, .5, 1, 1
    assert np.allclose(s, [0, .5, 1, 1, 0, .5, 1, 1, 0, .5, 1, 1])


def test_kl_follows_the_reference_definition_and_the_loss_terms_have_no_cpu_path():
    import pytest
    from latentdiffeq_amd._lib import LdeError
    mu = torch.tensor([[0.5, -1.0], [0.0, 2.0]])
    ls = torch.tensor([[0.0, 0.3], [-0.2, 0.1]])
    e = (np.exp(ls.numpy()) + mu.numpy() ** 2 - ls.numpy() - 1) / 2
    assert np.allclose(TR.kl(mu, ls).numpy(), e)
    # vector_kl / sample / reconstruction_loss are liblde.so kernels (tests/test_gpu_loss.py); on CPU tensors they refuse
    for call in (lambda: TR.vector_kl(mu, ls), lambda: TR.sample(mu, ls), lambda: TR.reconstruction_loss(mu[:, :, None], ls[:, :, None])):
        with pytest.raises(LdeError):
            call()


def test_time_loader_and_normalisation():
    x = torch.arange(2 * 3 * 10, dtype=torch.float32).reshape(2, 3, 10)
    starts = set()
    rng = np.random.default_rng(0)
    for _ in range(200):
        w = TR.time_loader(x, 10, 4, rng)
        assert w.shape == (2, 3, 4)
        s0 = int(w[0, 0, 0])
        assert torch.equal(w, x[:, :, s0:s0 + 4])
        starts.add(s0)
    assert starts == set(range(0, 6))          # rand(1:full−seq) ⇒ 0-based starts 0..5; the last window (6) is never drawn
    X = np.array([[2.0, 4.0], [6.0, 10.0]])
    Xh, lo, hi = TR.normalize_to_unit_segment(X)
    assert (lo, hi) == (2.0, 10.0) and Xh.min() == 0 and Xh.max() == 1
    assert np.allclose(TR.denormalize_unit_segment(Xh, lo, hi), X)


def test_pendulum_frames_geometry():
    """create_frames: the reference drawing's geometry [REF examples/pendulum_friction-less/create_data.jl:90-111] on a 28×28
    canvas — pivot disc at (0, −8.5) with a dark centre, bob 19 px away along π/2 + θ (y down), a 3.75-thick rod between."""
    from latentdiffeq_amd.data import create_frames
    f = create_frames(torch.tensor([0.0, 0.5, -0.5]))
    assert f.shape == (3, 28, 28) and float(f.min()) >= 0 and float(f.max()) <= 1
    f0 = f[0]
    sub = 4.0 / 16 + 1e-6   # the rod's edge x = ±1.875 lies exactly on a sub-sample column and cos(π/2) is −4e-8 in f32: one column of 4
    assert float((f0 - f0.flip(1)).abs().max()) <= sub                     # θ = 0: mirror-symmetric in x
    assert float((f[1] - f[2].flip(1)).abs().max()) <= sub                 # ±θ are mirror images
    # bob centre at y = −8.5 + 19 = 10.5 → rows 24/25, columns 13/14 fully lit; pivot centre (row 5, cols 13/14) dark
    assert float(f0[24, 13]) == 1 and float(f0[24, 14]) == 1 and float(f0[5:6, 13:15].mean()) < 0.5
    assert float(f0[15, 13]) == 1                                          # the rod runs down the middle
    assert float(f0[15, 5]) == 0 and float(f0[2, 2]) == 0                  # background
    # lit area ≈ rod (19 × 3.75) + two half discs at its ends + the parts of the discs outside the rod − inner disc
    area = float(f0.sum())
    assert 70 < area < 90
    # θ = 0.5: the bob moves to x = −19·sin(0.5) ≈ −9.1 → column ≈ 4–5, y = −8.5 + 19·cos(0.5) ≈ 8.2 → row ≈ 22
    r, c = np.unravel_index(int(f[1][18:].argmax()), f[1][18:].shape)
    assert abs((c + 0.5 - 14) - (-19 * np.sin(0.5))) < 2.5


def test_sample_layout_is_tracked_not_inferred_from_shapes(monkeypatch):
    """loss._sample1 undoes the permutation it applied even when the permutation does not change the shape ([16, 16])."""
    from latentdiffeq_amd import loss as LS
    monkeypatch.setattr(LS._SampleFn, "apply", staticmethod(lambda m, s, eps: m.clone()))     # stand-in kernel: l̃ = μ
    monkeypatch.setattr(LS, "randn", lambda shape, device: torch.zeros(shape))                 # (ε comes from a kernel too: lde_randn)
    for B in (16, 32):
        mu_b, ls_b = torch.randn(B, 16), torch.randn(B, 16)
        mu, ls = mu_b.t(), ls_b.t()
        m, s, order = LS._same_layout(mu, ls)
        assert order == [1, 0] and m.is_contiguous() and s.is_contiguous()
        assert torch.equal(LS._sample1(mu, ls), mu)
    a, b, order = LS._same_layout(torch.randn(4, 5), torch.randn(4, 5))
    assert order is None


def test_flux_flavour_adamw_update():
    """ADAMW(η, β, decay) = Optimiser(ADAM(η, β), WeightDecay(decay)) in the pinned Flux [REF Manifest.toml:452]: the ADAM step
    η·m̂/(√v̂ + ε) and then `Δ += decay·x` — the decay is NOT scaled by η  [REF examples/pendulum_friction-less/model_train.jl:138]."""
    from latentdiffeq_amd.train import FluxADAMW
    torch.manual_seed(0)
    x = torch.nn.Parameter(torch.randn(7, dtype=torch.float64))
    x0 = x.detach().clone()
    eta, b1, b2, decay, eps = 1e-3, 0.9, 0.999, 1e-3, 1e-8
    opt = FluxADAMW([x], lr=eta, betas=(b1, b2), decay=decay)
    m = torch.zeros_like(x0)
    v = torch.zeros_like(x0)
    xr = x0.clone()
    for k in range(1, 4):
        g = torch.sin(xr * k) + 0.3
        x.grad = torch.sin(x.detach() * k) + 0.3
        opt.step()
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        delta = m / (1 - b1 ** k) / (torch.sqrt(v / (1 - b2 ** k)) + eps) * eta     # Flux ADAM: mt/(1-β1^t) / (√(vt/(1-β2^t)) + ε) · η
        xr = xr - (delta + decay * xr)                                              # WeightDecay: Δ += decay·x, then x -= Δ
        assert torch.allclose(x.detach(), xr, rtol=1e-12, atol=1e-14), k
    # with decay = η = 1e-3 torch's AdamW (lr·wd·x) would decay 1000× more weakly: the two flavours must differ
    y = torch.nn.Parameter(x0.clone())
    o2 = torch.optim.AdamW([y], lr=eta, betas=(b1, b2), weight_decay=decay, eps=eps)
    y.grad = torch.sin(y.detach()) + 0.3
    o2.step()
    z = torch.nn.Parameter(x0.clone())
    o3 = FluxADAMW([z], lr=eta, betas=(b1, b2), decay=decay)
    z.grad = torch.sin(z.detach()) + 0.3
    o3.step()
    assert float((y.detach() - z.detach()).abs().max()) > 1e-4


def test_dataset_container_round_trip_and_loader(tmp_path):
    """f-4's on-disk format: the (latent_data, u0s, ps, high_dim_data) tuple written and read back, then the example script's
    reshape / 90-10 split / loader  [REF examples/pendulum_friction-less/model_train.jl:86-121]."""
    from latentdiffeq_amd import data as D
    rng = np.random.default_rng(0)
    T, n = 6, 20
    latent, u0s, ps = rng.standard_normal((2, T, n)), rng.standard_normal((2, n)), rng.uniform(1, 2, (1, n))
    high = rng.uniform(0, 1, (28, 28, T, n))
    path = D.save_dataset(str(tmp_path / "data"), torch.from_numpy(latent), u0s, ps, high)
    got = D.load_dataset(path)
    for a, b in zip(got, (latent, u0s, ps, high)):
        assert a.dtype == np.float32 and np.array_equal(a, b.astype(np.float32))
    prep = D.prepare_training_data(got)
    assert prep["input_dim"] == 784 and prep["full_seq_len"] == T
    assert prep["train_set"].shape == (784, T, 18) and prep["val_set"].shape == (784, 2, T)
    # column-major vectorisation: pixel (i, j) of frame (t, k) is row i + 28·j
    assert prep["train_set"][3 + 28 * 5, 2, 7] == np.float32(high[3, 5, 2, 7])
    assert prep["val_set"][3 + 28 * 5, 1, 4] == np.float32(high[3, 5, 4, 19])
    assert prep["train_set_params"].shape == (1, 18) and prep["val_set_latent"].shape == (2, T, 2)
    batches = list(D.data_loader(prep["train_set"], 4, np.random.default_rng(1)))
    assert len(batches) == 4 and all(b.shape == (784, 4, T) for b in batches)          # partial=false: 18 // 4 full batches
    import pytest
    with pytest.raises(KeyError):
        np.savez(str(tmp_path / "bad.npz"), x=np.zeros(3))
        D.load_dataset(str(tmp_path / "bad.npz"))
