"""GPU: lde_set_dw_stream / lde_join_dw — the chain and recurrent pullbacks put their weight-gradient kernels on a stream of
their own (include/lde.h). Same kernels, same inputs, another stream: after the join every gradient must equal the default
mode's bit for bit, step after step (the handles reuse their workspaces, so a missing wait shows up as a changed gradient),
also while the main stream is kept busy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(seed):
    import torch
    import latentdiffeq_amd as la
    from latentdiffeq_amd import train as TR
    torch.manual_seed(seed)
    mt, diffeq = la.GOKU_basic(), la.Pendulum()
    enc, dec = TR.default_layers(mt, 784, diffeq, device="cuda")
    with torch.no_grad():
        dec[0][1]._dense[-1].bias.fill_(1.0)
    return TR.LatentDiffEqModel(mt, enc, dec)


def _grads(model, x, ts, seed):
    import torch
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd import train as TR
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(seed)
    loss = TR.loss_batch(model, x, ts, 1e-3, True)
    loss.backward()
    L.join_weight_gradients()
    torch.cuda.synchronize()
    return float(loss.detach()), [p.grad.detach().clone() for p in model.parameters()]


def test_weight_gradients_on_their_own_stream_equal_the_default():
    import torch
    from latentdiffeq_amd import _lib as L
    B, T = 64, 20
    ts = np.arange(T) * 0.05
    x = torch.rand(T, B, 784, device="cuda").permute(2, 1, 0)
    a, b = _model(5), _model(5)
    busy = torch.rand(2048, 2048, device="cuda")
    try:
        for step in range(4):
            L.set_async_weight_gradients(False)
            la_, ga = _grads(a, x, ts, 100 + step)
            L.set_async_weight_gradients(True)
            assert L.dw_stream is not None
            if step % 2:
                for _ in range(4):
                    busy = busy @ busy * 1e-3          # keep the main stream busy behind the pullback
            lb, gb = _grads(b, x, ts, 100 + step)
            assert la_ == lb
            for u, v in zip(ga, gb):
                assert torch.equal(u, v)
            with torch.no_grad():
                for pa, pb, g in zip(a.parameters(), b.parameters(), ga):
                    pa.add_(g, alpha=-1e-3)
                    pb.add_(g, alpha=-1e-3)
    finally:
        L.set_async_weight_gradients(False)
    assert L.dw_stream is None


def test_join_without_a_stream_is_a_no_op():
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    assert lib.lde_set_dw_stream(None) == 0 and lib.lde_join_dw(None) == 0
    L.join_weight_gradients()


def test_the_weight_gradient_stream_is_a_branch_of_a_captured_step():
    """Stream capture with the weight-gradient stream set: the capture forks into it and joins back (lde_join_dw); the events the eager
    warm-up steps left behind are not waited for across the capture boundary (the runtime refuses such a wait once the dw stream itself is
    being captured — found in round 5), and grouped calls stay available. Replayed gradients equal the eager, synchronous ones bit for bit
    (the non-variational loss: no ε to keep in step)."""
    import torch
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd import recurrent as R
    from latentdiffeq_amd import train as TR
    B, T = 32, 20
    ts = np.arange(T) * 0.05
    x = torch.rand(T, B, 784, device="cuda").permute(2, 1, 0)
    m = _model(7)
    old = R._BRANCH_STREAMS
    R._BRANCH_STREAMS = False           # (a captured step runs the encoder on one stream)

    def grads():
        for p in m.parameters():
            p.grad = None
        loss = TR.loss_batch(m, x, ts, 1e-3, False)
        loss.backward()
        L.join_weight_gradients()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in m.parameters()]
    try:
        L.set_async_weight_gradients(False)
        g0 = grads()
        L.set_async_weight_gradients(True)
        for _ in range(2):
            g1 = grads()                # eager, asynchronous: leaves pending events behind
        assert all(torch.equal(u, v) for u, v in zip(g0, g1))
        for p in m.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(graph, stream=side):
            loss = TR.loss_batch(m, x, ts, 1e-3, False)
            loss.backward()
            L.join_weight_gradients()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        g2 = [p.grad.detach().clone() for p in m.parameters()]
        assert all(torch.equal(u, v) for u, v in zip(g0, g2))
    finally:
        L.set_async_weight_gradients(False)
        R._BRANCH_STREAMS = old
