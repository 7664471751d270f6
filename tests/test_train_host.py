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
