"""CPU known-answer tests for the recurrent-stack oracle (oracle/lde_rnn_oracle.c): pinned against torch autograd, float64,
on a hand-written restatement of Flux 0.13's RNNCell / LSTMCell (gate order input, forget, cell, output; trainable state0)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O


def torch_stack(cell, sizes, reverse, Wflat, x):
    """x (T, B, in) tensor; returns the top cell's output after the last processed frame, (B, h_last)."""
    G = 4 if cell == O.CELL_LSTM else 1
    T, B, _ = x.shape
    cells, off = [], 0
    for l in range(len(sizes) - 1):
        n_in, h = sizes[l], sizes[l + 1]
        Wi = Wflat[off:off + G * h * n_in].reshape(n_in, G * h).T; off += G * h * n_in
        Wh = Wflat[off:off + G * h * h].reshape(h, G * h).T; off += G * h * h
        b = Wflat[off:off + G * h]; off += G * h
        h0 = Wflat[off:off + h]; off += h
        c0 = None
        if cell == O.CELL_LSTM:
            c0 = Wflat[off:off + h]; off += h
        cells.append((Wi, Wh, b, h0, c0, h))
    assert off == Wflat.numel()
    hs = [c[3].expand(B, -1) for c in cells]
    cs = [c[4].expand(B, -1) if c[4] is not None else None for c in cells]
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        inp = x[t]
        for l, (Wi, Wh, b, _, _, h) in enumerate(cells):
            g = inp @ Wi.T + hs[l] @ Wh.T + b
            if cell == O.CELL_LSTM:
                i, f, c_, o = torch.sigmoid(g[:, :h]), torch.sigmoid(g[:, h:2 * h]), torch.tanh(g[:, 2 * h:3 * h]), torch.sigmoid(g[:, 3 * h:])
                cs[l] = f * cs[l] + i * c_
                hs[l] = o * torch.tanh(cs[l])
            else:
                hs[l] = torch.tanh(g) if cell == O.CELL_RNN_TANH else torch.relu(g)
            inp = hs[l]
    return hs[-1]


CASES = [
    (O.CELL_RNN_RELU, (32, 16, 16), True),      # pe_z₀               [REF src/models/GOKU.jl:229-230]
    (O.CELL_LSTM, (32, 16, 16), False),         # pe_θ_forward        [REF src/models/GOKU.jl:233-234]
    (O.CELL_LSTM, (32, 16, 16), True),          # pe_θ_backward       [REF src/models/GOKU.jl:236-237]
    (O.CELL_RNN_TANH, (5, 7, 3, 9), False),
    (O.CELL_LSTM, (3, 10), True),
]


@pytest.mark.parametrize("cell,sizes,reverse", CASES)
def test_rnn_oracle_matches_torch_f64(o64, o32, cell, sizes, reverse):
    d = O.make_rnn_desc(cell, sizes, reverse)
    W = O.rnn_weights(cell, sizes, seed=3).astype(np.float64)
    assert o64.rnn_num_weights(d) == W.size
    rng = np.random.default_rng(4)
    T, B = 9, 11
    x = rng.standard_normal((T, B, sizes[0]))
    dy = rng.standard_normal((B, sizes[-1]))
    y = o64.rnn_forward(d, W, x)
    xt, Wt = torch.tensor(x, requires_grad=True), torch.tensor(W, requires_grad=True)
    yt = torch_stack(cell, sizes, reverse, Wt, xt)
    assert np.abs(y - yt.detach().numpy()).max() <= 1e-12
    (yt * torch.tensor(dy)).sum().backward()
    dx, dW = o64.rnn_backward(d, W, x, dy)
    assert np.abs(dx - xt.grad.numpy()).max() <= 1e-11 * max(1.0, np.abs(dx).max())
    assert np.abs(dW - Wt.grad.numpy()).max() <= 1e-11 * max(1.0, np.abs(dW).max())
    y32 = o32.rnn_forward(d, W.astype(np.float32), x.astype(np.float32))
    assert np.abs(y32 - y).max() <= 2e-5
    dx32, dW32 = o32.rnn_backward(d, W.astype(np.float32), x.astype(np.float32), dy.astype(np.float32))
    assert np.abs(dx32 - dx).max() <= 5e-5 * max(1.0, np.abs(dx).max())
    assert np.abs(dW32 - dW).max() <= 5e-5 * max(1.0, np.abs(dW).max())


def test_rnn_oracle_threads_agree_and_dx_optional(o64):
    cell, sizes = O.CELL_LSTM, (6, 8, 8)
    d = O.make_rnn_desc(cell, sizes, True)
    W = O.rnn_weights(cell, sizes).astype(np.float64)
    rng = np.random.default_rng(1)
    x, dy = rng.standard_normal((5, 20, 6)), rng.standard_normal((20, 8))
    dx1, dW1 = o64.rnn_backward(d, W, x, dy, nthreads=1)
    dx4, dW4 = o64.rnn_backward(d, W, x, dy, nthreads=4)
    assert np.array_equal(dx1, dx4) and np.abs(dW1 - dW4).max() <= 1e-13 * np.abs(dW1).max()
    assert o64.rnn_backward(d, W, x, dy, need_dx=False)[0] is None
