"""Isolated timing of lde_rnn_forward / lde_rnn_backward (HIP events), LSTM 32-16-16 and RNN 32-16-16, B=256, T=50."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentdiffeq_amd import _lib as L
from latentdiffeq_amd import synthetic  # noqa
lib = L.load()
B, T = int(os.environ.get('RB_B', 256)), int(os.environ.get('RB_T', 50))
for cell, name in ((L.CELL_LSTM, "lstm"), (L.CELL_RNN_RELU, "rnn")):
    d = L.RnnDesc(); d.abi_version, d.cell, d.n_layers, d.reverse = 1, cell, 2, 1
    for i, s in enumerate((32, 16, 16)): d.sizes[i] = s
    h = C.c_void_p(); assert lib.lde_rnn_create(C.byref(d), C.byref(h)) == 0
    nW = lib.lde_rnn_num_weights(C.byref(d))
    W = (np.random.default_rng(0).standard_normal(nW) * 0.2).astype(np.float32)
    lib.lde_rnn_set_weights(h, W.ctypes.data_as(C.c_void_p), nW)
    x = torch.randn(T, B, 32, device="cuda"); y = torch.empty(B, 16, device="cuda"); dy = torch.randn(B, 16, device="cuda")
    dx = torch.empty_like(x); dW = torch.zeros(nW, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    f = lambda: lib.lde_rnn_forward(h, p(x), T, B, p(y), s)
    b = lambda: lib.lde_rnn_backward(h, p(x), p(dy), T, B, p(dx), p(dW), s)
    for fn, nm in ((f, "forward"), (b, "backward")):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(name, nm, round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")
    lib.lde_rnn_destroy(h)
