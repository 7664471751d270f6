"""GPU: lde_adamw_flux_step — the training step's parameter update, Flux's `ADAMW(η, β, decay)` = `Optimiser(ADAM, WeightDecay)`
[REF examples/pendulum_friction-less/model_train.jl:138, :190-192], one launch for all arrays — against the formula evaluated
in float64 on the host (the same hand computation tests/test_train_host.py pins the torch path with), and against that torch
path. Tolerance: 2e-6 of the largest parameter after five steps (f32 state and arithmetic; the update is a few roundings)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flux_reference(x0, grads, eta, b1, b2, eps, decay):
    x, m, v = x0.astype(np.float64), np.zeros_like(x0, np.float64), np.zeros_like(x0, np.float64)
    for k, g in enumerate(grads, 1):
        g = g.astype(np.float64)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        delta = m / (1 - b1 ** k) / (np.sqrt(v / (1 - b2 ** k)) + eps) * eta
        x = x - (delta + decay * x)
    return x


@pytest.mark.parametrize("decay", [0.0, 1e-3])
def test_native_update_matches_the_flux_formula_and_the_torch_path(decay):
    import torch
    from latentdiffeq_amd.train import FluxADAMW
    rng = np.random.default_rng(5)
    sizes = [1, 3, 784 * 200 + 200, 2048, 2049, 4097, 16 * 32 + 16]
    eta, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    x0 = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    grads = [[(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 2)).astype(np.float32) for n in sizes] for _ in range(5)]

    def params(offset):
        out = []
        for i, a in enumerate(x0):
            buf = torch.zeros(a.size + 1, device="cuda")
            v = buf[offset if i % 2 else 0:][:a.size]      # every second array starts one float into its buffer (unaligned pointers)
            v.copy_(torch.from_numpy(a))
            out.append(torch.nn.Parameter(v))
        return out

    pn, pt = params(1), params(0)
    on, ot = FluxADAMW(pn, lr=eta, betas=(b1, b2), decay=decay, eps=eps), FluxADAMW(pt, lr=eta, betas=(b1, b2), decay=decay, eps=eps, native=False)
    assert on.native and not ot.native
    for gs in grads:
        for p, q, g in zip(pn, pt, gs):
            p.grad = torch.from_numpy(g).cuda()
            q.grad = torch.from_numpy(g).cuda()
        on.step()
        ot.step()
    for i, (p, q) in enumerate(zip(pn, pt)):
        want = _flux_reference(x0[i], [gs[i] for gs in grads], eta, b1, b2, eps, decay)
        scale = np.abs(want).max()
        assert np.abs(p.detach().cpu().numpy() - want).max() <= 2e-6 * scale, sizes[i]
        assert np.abs(q.detach().cpu().numpy() - want).max() <= 2e-6 * scale, sizes[i]
        assert on.state[p]["step"] == 5


def test_native_update_skips_parameters_without_gradient_and_rejects_bad_arguments():
    import torch
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd.train import FluxADAMW
    a, b = torch.nn.Parameter(torch.ones(10, device="cuda")), torch.nn.Parameter(torch.ones(10, device="cuda"))
    opt = FluxADAMW([a, b], lr=0.1)
    a.grad = torch.ones(10, device="cuda")
    opt.step()
    assert torch.equal(b.detach(), torch.ones(10, device="cuda")) and float(a.detach()[0]) < 1.0
    # second step: both have gradients; a is at its step 2, b at its step 1 — each with its own bias correction (Flux keeps β₁ᵗ, β₂ᵗ per array)
    a.grad = torch.full((10,), 0.5, device="cuda")
    b.grad = torch.full((10,), 0.5, device="cuda")
    opt.step()
    one = np.ones(10, np.float32)
    want_a = _flux_reference(one, [one, 0.5 * one], 0.1, 0.9, 0.999, 1e-8, 0.0)
    want_b = _flux_reference(one, [0.5 * one], 0.1, 0.9, 0.999, 1e-8, 0.0)
    assert np.abs(a.detach().cpu().numpy() - want_a).max() <= 2e-6 and np.abs(b.detach().cpu().numpy() - want_b).max() <= 2e-6
    assert opt.state[a]["step"] == 2 and opt.state[b]["step"] == 1
    lib = L.load()
    t = (L.AdamTensor * 1)()
    t[0].n = 4
    assert lib.lde_adamw_flux_step(1, t, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1       # null pointers
    assert lib.lde_adamw_flux_step(0, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None) == -1    # step counts from 1
    assert lib.lde_adamw_flux_step(0, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == 0
    with pytest.raises(ValueError):
        FluxADAMW([torch.nn.Parameter(torch.ones(3))], native=True)                         # CPU parameters: no native path


def test_capturable_step_count_survives_a_checkpoint():
    """The capturable form keeps t in one device word: a state_dict carries it as every array's `step`, a loaded one restores it —
    a resumed run continues the bias corrections at t (restarting at 0 scales m̂ by 10 and v̂ by 1000 on the first resumed step) —
    and the checkpoint is the one the non-capturable form writes and reads. One parameter group only (one launch bumps the word)."""
    import torch
    from latentdiffeq_amd.train import FluxADAMW
    rng = np.random.default_rng(11)
    eta, b1, b2, eps, decay = 1e-3, 0.9, 0.999, 1e-8, 1e-3
    x0 = rng.standard_normal(1000).astype(np.float32)
    grads = [rng.standard_normal(1000).astype(np.float32) for _ in range(6)]

    def run(opt, p, gs):
        for g in gs:
            p.grad = torch.from_numpy(g).cuda()
            opt.step()

    p1 = torch.nn.Parameter(torch.from_numpy(x0).cuda())
    o1 = FluxADAMW([p1], lr=eta, betas=(b1, b2), decay=decay, eps=eps, capturable=True)
    run(o1, p1, grads[:3])
    import io
    buf = io.BytesIO()
    torch.save(o1.state_dict(), buf)          # through a file image: load_state_dict does not copy tensors that are already on the device
    sd = torch.load(io.BytesIO(buf.getvalue()))
    assert all(st["step"] == 3 for st in sd["state"].values())
    p2 = torch.nn.Parameter(p1.detach().clone())
    o2 = FluxADAMW([p2], lr=eta, betas=(b1, b2), decay=decay, eps=eps, capturable=True)
    o2.load_state_dict(sd)
    assert int(o2._step_dev.item()) == 3
    p3 = torch.nn.Parameter(p1.detach().clone())
    o3 = FluxADAMW([p3], lr=eta, betas=(b1, b2), decay=decay, eps=eps)          # the non-capturable form resumes from the same file
    o3.load_state_dict(torch.load(io.BytesIO(buf.getvalue())))
    run(o1, p1, grads[3:])
    run(o2, p2, grads[3:])
    run(o3, p3, grads[3:])
    want = _flux_reference(x0, grads, eta, b1, b2, eps, decay)
    for p in (p1, p2, p3):
        assert np.abs(p.detach().cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
    assert torch.equal(p1, p2)
    with pytest.raises(ValueError):
        FluxADAMW([{"params": [p1]}, {"params": [p2]}], capturable=True)
    with pytest.raises(ValueError):
        o1.add_param_group({"params": [p3]})
