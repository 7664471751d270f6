"""GPU: lde_refresh_weights — the weights of many modules handed to the library in ONE launch after an optimiser step
(include/lde.h) — is the same hand-over as the per-module lde_chain_set_weights_device / lde_rnn_set_weights_device calls:
identical outputs and gradients bit for bit, and a parameter that changes afterwards is picked up again by the module
itself (the reference has no such step: Flux layers read their arrays in place [REF model_train.jl:190-192])."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(seed):
    import torch
    import latentdiffeq_amd as la
    from latentdiffeq_amd import train as TR
    torch.manual_seed(seed)
    mt, diffeq = la.GOKU_basic(), la.Pendulum()
    enc, dec = TR.default_layers(mt, 64, diffeq, device="cuda", hidden_dim_resnet=48)
    with torch.no_grad():
        dec[0][1]._dense[-1].bias.fill_(1.0)
    return TR.LatentDiffEqModel(mt, enc, dec)


def _loss_and_grads(model, x, ts):
    import torch
    from latentdiffeq_amd import train as TR
    for p in model.parameters():
        p.grad = None
    loss = TR.loss_batch(model, x, ts, 1e-3, False)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), [p.grad.detach().clone() for p in model.parameters()]


def test_refresh_weights_equals_per_module_uploads_and_tracks_changes():
    import torch
    B, T = 24, 12
    ts = np.arange(T) * 0.05
    x = torch.rand(T, B, 64, device="cuda").permute(2, 1, 0)
    a, b = _model(7), _model(7)
    for _ in range(3):
        la_, ga = _loss_and_grads(a, x, ts)                 # a: every module uploads its own weights at every call
        assert b.refresh_weights() == 11                    # b: one launch for the eleven modules, calls skip the upload
        assert all(m._wkey is not None for m in b.modules())
        lb, gb = _loss_and_grads(b, x, ts)
        assert la_ == lb
        for u, v in zip(ga, gb):
            assert torch.equal(u, v)
        with torch.no_grad():                               # an "optimiser step" on both models
            for pa, pb, g in zip(a.parameters(), b.parameters(), ga):
                pa.add_(g, alpha=-1e-3)
                pb.add_(g, alpha=-1e-3)
    # b's parameters changed after its last refresh: the keys no longer match, every module re-uploads by itself
    la_, ga = _loss_and_grads(a, x, ts)
    lb, gb = _loss_and_grads(b, x, ts)
    assert la_ == lb and all(torch.equal(u, v) for u, v in zip(ga, gb))
    assert all(m._wkey is None for m in b.modules())


def test_native_optimiser_steps_without_refresh_are_seen_by_the_modules():
    """refresh_weights() ONCE, then native FluxADAMW steps (lde_adamw_flux_step writes the parameters through raw pointers) without
    another refresh: the optimiser bumps the parameters' version counters, so every module notices its (storage, version) key no
    longer matches and uploads by itself — same losses and gradients as a model that is refreshed after every step. (Before round 3
    the native update left the key unchanged and such a loop trained on the weights of the first refresh.)"""
    import torch
    from latentdiffeq_amd.train import FluxADAMW
    B, T = 24, 12
    ts = np.arange(T) * 0.05
    x = torch.rand(T, B, 64, device="cuda").permute(2, 1, 0)
    a, b = _model(11), _model(11)
    oa = FluxADAMW(list(a.parameters()), lr=1e-2, decay=1e-10)
    ob = FluxADAMW(list(b.parameters()), lr=1e-2, decay=1e-10)
    assert b.refresh_weights() == 11
    losses = []
    for _ in range(3):
        la_, _ = _loss_and_grads(a, x, ts)
        oa.step()
        a.refresh_weights()                                 # a: the documented loop — refresh after every update
        lb, _ = _loss_and_grads(b, x, ts)
        ob.step()                                           # b: never refreshed again
        assert la_ == lb, (la_, lb)
        losses.append(la_)
    assert len(set(losses)) == 3                            # (the weights did move: three different losses)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def test_refresh_weights_rejects_bad_arguments():
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    kinds, handles, ptrs = (C.c_int32 * 1)(7), (C.c_void_p * 1)(1), (C.c_void_p * 1)(1)
    assert lib.lde_refresh_weights(1, kinds, handles, ptrs, None) == -1          # unknown module kind
    kinds[0] = L.MODULE_CHAIN
    handles[0] = None
    assert lib.lde_refresh_weights(1, kinds, handles, ptrs, None) == -1          # null handle
    assert lib.lde_refresh_weights(0, None, None, None, None) == 0
    assert lib.lde_refresh_weights(-1, None, None, None, None) == -1


def test_handover_in_bf16_mode_skips_the_f32_fragments_and_a_switch_back_rebuilds_them():
    """lde_refresh_weights for a chain in bf16 mode rebuilds its bf16 fragments only (round 3: half the hand-over's work in a mixed-precision
    step); lde_chain_set_dtype back to f32 rebuilds the f32 fragments from the handle's copy of the weights — the f32 forward after
    hand-over + switch equals the f32 forward of a fresh handle given the same weights."""
    import torch
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd.chain import Chain, Dense, SkipConnection
    torch.manual_seed(8)
    mk = lambda: Chain(Dense(24, 40, "relu"), SkipConnection(Dense(40, 40, "relu")), Dense(40, 12, "sigmoid")).to("cuda")
    c = mk()
    x = torch.randn(24, 96, device="cuda")
    c.set_dtype("bf16")
    y_b0 = c(x).clone()
    with torch.no_grad():
        c.theta.mul_(1.5).add_(0.01)                    # an optimiser step's worth of change
    torch.autograd.graph.increment_version(c.theta)
    assert L.refresh_weights([c]) == 1                  # hand-over in bf16 mode
    y_b1 = c(x).clone()
    assert not torch.equal(y_b0, y_b1)                  # the bf16 fragments are the new weights'
    c.set_dtype("f32")                                  # … and now the f32 ones must be too
    y_f = c(x).clone()
    fresh = mk()
    with torch.no_grad():
        fresh.theta.copy_(c.theta)
    assert torch.equal(y_f, fresh(x))
