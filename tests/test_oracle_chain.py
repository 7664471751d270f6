"""CPU known-answer tests for the dense-chain oracle (oracle/lde_chain_oracle.c): pinned against torch (an independent
implementation of the same published layer definitions — Flux Dense / SkipConnection / σ / softplus), float64."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

ACT = {O.CACT_IDENTITY: lambda v: v, O.CACT_RELU: torch.relu, O.CACT_TANH: torch.tanh, O.CACT_SIGMOID: torch.sigmoid,
       O.CACT_SOFTPLUS: torch.nn.functional.softplus}

# the reference's default decoder chains  [REF src/models/GOKU.jl:252-269]
RECON = ((2, 200, 200, 200, 784), (O.CACT_RELU, O.CACT_RELU, O.CACT_RELU, O.CACT_SIGMOID), (0, 1, 1, 0))
LO_Z0 = ((16, 200, 2), (O.CACT_RELU, O.CACT_IDENTITY), (0, 0))
LO_TH = ((16, 200, 1), (O.CACT_RELU, O.CACT_SOFTPLUS), (0, 0))
ODD = ((5, 33, 33, 7, 7, 19), (O.CACT_TANH, O.CACT_SOFTPLUS, O.CACT_RELU, O.CACT_SIGMOID, O.CACT_IDENTITY), (0, 1, 0, 1, 0))


def torch_chain(sizes, acts, skips, Wflat, x):
    """x (N, in) torch tensor; weights in Flux.destructure order (vec(W) column-major [out×in], then b)."""
    off, h = 0, x
    for l in range(len(sizes) - 1):
        ni, no = sizes[l], sizes[l + 1]
        W = Wflat[off:off + no * ni].reshape(ni, no).T      # column-major [out×in]
        off += no * ni
        b = Wflat[off:off + no]
        off += no
        a = ACT[acts[l]](h @ W.T + b)
        h = a + h if skips[l] else a
    assert off == Wflat.numel()
    return h


@pytest.mark.parametrize("spec", [RECON, LO_Z0, LO_TH, ODD], ids=["reconstructor", "lo_z0", "lo_theta", "odd"])
def test_chain_oracle_matches_torch_f64(o64, o32, spec):
    sizes, acts, skips = spec
    d = O.make_chain_desc(sizes, acts, skips)
    W = O.mlp_weights(sizes, seed=7).astype(np.float64)
    assert o64.chain_num_weights(d) == W.size
    rng = np.random.default_rng(5)
    N = 37
    x = rng.standard_normal((N, sizes[0]))
    dy = rng.standard_normal((N, sizes[-1]))
    y = o64.chain_forward(d, W, x)
    xt = torch.tensor(x, requires_grad=True)
    Wt = torch.tensor(W, requires_grad=True)
    yt = torch_chain(sizes, acts, skips, Wt, xt)
    assert np.abs(y - yt.detach().numpy()).max() <= 1e-12 * max(1.0, np.abs(y).max())
    (yt * torch.tensor(dy)).sum().backward()
    dx, dW = o64.chain_backward(d, W, x, dy)
    assert np.abs(dx - xt.grad.numpy()).max() <= 1e-11 * max(1.0, np.abs(dx).max())
    assert np.abs(dW - Wt.grad.numpy()).max() <= 1e-11 * max(1.0, np.abs(dW).max())
    # the f32 build agrees with float64 to fp32 round-off
    y32 = o32.chain_forward(d, W.astype(np.float32), x.astype(np.float32))
    assert np.abs(y32 - y).max() <= 2e-5 * max(1.0, np.abs(y).max())
    dx32, dW32 = o32.chain_backward(d, W.astype(np.float32), x.astype(np.float32), dy.astype(np.float32))
    assert np.abs(dx32 - dx).max() <= 5e-5 * max(1.0, np.abs(dx).max())
    assert np.abs(dW32 - dW).max() <= 5e-5 * max(1.0, np.abs(dW).max())


def test_chain_oracle_accumulates_dw_and_threads_agree(o64):
    sizes, acts, skips = LO_TH
    d = O.make_chain_desc(sizes, acts, skips)
    W = O.mlp_weights(sizes, seed=8).astype(np.float64)
    rng = np.random.default_rng(1)
    x, dy = rng.standard_normal((64, 16)), rng.standard_normal((64, 1))
    dx1, dW1 = o64.chain_backward(d, W, x, dy, nthreads=1)
    dx4, dW4 = o64.chain_backward(d, W, x, dy, nthreads=4)
    assert np.array_equal(dx1, dx4) and np.abs(dW1 - dW4).max() <= 1e-13 * np.abs(dW1).max()
    assert o64.chain_backward(d, W, x, dy, need_dx=False)[0] is None


def test_chain_golden_fixture(o32, o64):
    """tests/golden/chain_decoder.npz (made by tests/golden/make_golden.py): inputs, f32 oracle outputs, f64 truth."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "chain_decoder.npz"))
    sizes, acts, skips = tuple(g["sizes"]), tuple(g["acts"]), tuple(g["skips"])
    d = O.make_chain_desc(sizes, acts, skips)
    y = o32.chain_forward(d, g["W"], g["x"])
    assert np.array_equal(y, g["y_f32"])                       # bit-reproducible build flags (oracle/Makefile)
    assert np.abs(y - g["y_f64"]).max() <= 2e-6
    dx, dW = o32.chain_backward(d, g["W"], g["x"], g["dy"])
    assert np.abs(dx - g["dx_f64"]).max() <= 2e-5 * np.abs(g["dx_f64"]).max()
    assert np.abs(dW - g["dW_f64"]).max() <= 2e-5 * np.abs(g["dW_f64"]).max()
