"""Host-side mirror of the encoder side of the model (scope row f-2): recurrent pattern extractor + the dense layers
around it.

    reference                                                                    here
    ---------------------------------------------------------------------------  ------------------------------------
    RNN(in, out, relu; init), LSTM(in, out; init)      [Flux; REF GOKU.jl:229-238]  RNN(in, out, act), LSTM(in, out)
    Chain(RNN(..), RNN(..)) applied frame by frame     [REF GOKU.jl:229-238]        Recurrent(RNN(..), RNN(..), reverse=)
    apply_feature_extractor(encoder, x)                [REF GOKU.jl:19]             same  (a Chain: chain.py)
    apply_pattern_extractor(encoder, fe_out)           [REF GOKU.jl:32-51], [REF LatentODE.jl:24-33]   same
    apply_latent_in(encoder, pe_out)                   [REF GOKU.jl:61-72], [REF LatentODE.jl:35-43]   same
    Encoder(model_type, (fe, pe, li)); encoder(x)      [REF LatentDiffEqModel.jl:52-75]                Encoder, encode
    default_layers(...) (encoder part)                 [REF GOKU.jl:214-245], [REF LatentODE.jl:108-128]  default_encoder_layers
    sample(μ, logσ², model)                            [REF GOKU.jl:155-163], [REF LatentODE.jl:82-89]  sample

`Recurrent.__call__` is liblde.so (lde_rnn_forward / lde_rnn_backward, include/lde.h); the feature extractor and
latent_in are `Chain`s (lde_chain_*). `sample` (loss.py) draws ε with torch's generator and forms l̃ in one kernel (lde_sample_*).
"""
from __future__ import annotations

import ctypes as C
import os
import math

import torch

from . import _lib as L
from .api import GOKU, LatentODE
from .chain import Chain, Dense, SkipConnection


class _Cell:
    G, S, code = 1, 1, L.CELL_RNN_RELU

    def __init__(self, n_in: int, n_out: int):
        self.n_in, self.n_out = n_in, n_out
        bi, bh = 1.0 / math.sqrt(n_in), 1.0 / math.sqrt(n_out)       # kaiming_uniform(gain = 1/√3) ⇒ U(±1/√fan_in)  [REF GOKU.jl:204]
        self._init = [torch.empty(self.G * n_out, n_in).uniform_(-bi, bi), torch.empty(self.G * n_out, n_out).uniform_(-bh, bh),
                      torch.zeros(self.G * n_out), torch.zeros(self.S * n_out)]   # Wi, Wh, b, state0 (trainable in Flux 0.13)
        self._owner, self._off = None, 0

    def _piece(self, src, i):
        R, n_in, h = self.G * self.n_out, self.n_in, self.n_out
        sizes = [R * n_in, R * h, R, self.S * h]
        o = self._off + sum(sizes[:i])
        v = src[o:o + sizes[i]]
        return v.view(n_in, R).t() if i == 0 else v.view(h, R).t() if i == 1 else v

    def _get(self, i):
        return self._init[i] if self._owner is None else self._piece(self._owner.theta, i)

    Wi = property(lambda self: self._get(0))
    Wh = property(lambda self: self._get(1))
    b = property(lambda self: self._get(2))
    state0 = property(lambda self: self._get(3))

    def grads(self):
        """(∂Wi, ∂Wh, ∂b, ∂state0) as views of the owner's theta.grad."""
        return tuple(self._piece(self._owner.theta.grad, i) for i in range(4))

    def num_weights(self):
        R = self.G * self.n_out
        return R * self.n_in + R * self.n_out + R + self.S * self.n_out

    def flat(self):
        return torch.cat([self.Wi.t().reshape(-1), self.Wh.t().reshape(-1), self.b, self.state0])


class RNN(_Cell):
    """RNN(in, out, act): h' = act.(Wi*x .+ Wh*h .+ b)  [Flux RNNCell]."""

    def __init__(self, n_in: int, n_out: int, act: str = "relu"):
        super().__init__(n_in, n_out)
        if act not in ("relu", "tanh"):
            raise ValueError("RNN activation: relu or tanh")
        self.act = act
        self.code = L.CELL_RNN_RELU if act == "relu" else L.CELL_RNN_TANH


class LSTM(_Cell):
    """LSTM(in, out)  [Flux LSTMCell: gates input, forget, cell, output; forget-gate bias initialised to 1]."""
    G, S, code = 4, 2, L.CELL_LSTM

    def __init__(self, n_in: int, n_out: int):
        super().__init__(n_in, n_out)
        self._init[2][n_out:2 * n_out] = 1.0


class _RecurrentFn(torch.autograd.Function):
    """y = stack(x) on batch-major buffers: x (T, B, in) → y (B, h_last); backward = lde_rnn_backward."""

    @staticmethod
    def forward(ctx, rec: "Recurrent", x: torch.Tensor, W: torch.Tensor):
        if not x.is_cuda:
            raise L.LdeError("Recurrent needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        h = rec._native()
        lib = rec._lib
        stream = L.raw_stream(x.device.index)
        if rec._wkey != L.weights_key(W):   # not handed over by refresh_weights() since the parameter last changed
            Wc = W.detach().contiguous().float()
            L.check(lib.lde_rnn_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h, "lde_rnn_set_weights_device", rnn=True)
            rec._wkey = None
        T, B, _ = x.shape
        y = torch.empty((B, rec.sizes[-1]), device=x.device, dtype=torch.float32)
        # a pullback will follow (grad mode is off inside Function.forward: ctx.needs_input_grad says so): the sweep keeps its records
        fwd = lib.lde_rnn_forward_train if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else lib.lde_rnn_forward
        L.check(fwd(h, C.c_void_p(x.data_ptr()), T, B, C.c_void_p(y.data_ptr()), stream), h, "lde_rnn_forward", rnn=True)
        ctx.rec, ctx.need_dx = rec, x.requires_grad
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        rec = ctx.rec
        h = rec._native()
        lib = rec._lib
        (x,) = ctx.saved_tensors
        T, B, _ = x.shape
        dy = dy.contiguous().float()
        stream = L.raw_stream(x.device.index)
        dx = torch.empty_like(x) if ctx.need_dx else None
        dW = torch.empty((rec.num_weights,), device=x.device, dtype=torch.float32)       # written, not accumulated (set_accumulate(0))
        if L.dw_stream is not None:
            dW.record_stream(L.dw_stream)          # written on the weight-gradient stream (set_async_weight_gradients)
        L.check(lib.lde_rnn_backward(h, C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), T, B,
                                     C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(), C.c_void_p(dW.data_ptr()), stream),
                h, "lde_rnn_backward", rnn=True)
        return None, dx, dW


class _RecurrentGroupFn(torch.autograd.Function):
    """(y_1, …, y_n) = (stack_1(x), …, stack_n(x)): independent stacks on the same frames — the GOKU pattern extractor
    [REF src/models/GOKU.jl:32-51] — as ONE autograd node. Every stack runs on a HIP stream of its own (each fills only B/16
    workgroups), the raw stream handed straight to liblde.so: no stream context switches, one node instead of three each way.
    The pullback issues every stack's sweep (lde_rnn_backward_dx) before the first weight-gradient tail (lde_rnn_backward_dw), so the
    three long kernels start within a few microseconds of each other instead of one host round trip apart, and returns Σ dx.
    Tensors are allocated on the current stream's pool and recorded on the side stream that touches them."""

    @staticmethod
    def forward(ctx, recs, x, *Ws):
        if not x.is_cuda:
            raise L.LdeError("Recurrent needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        dev = x.device
        main = torch.cuda.current_stream(dev)
        streams = _side_streams.setdefault((dev.index, len(recs)), [torch.cuda.Stream(dev) for _ in recs])
        T, B, _ = x.shape
        ready = main.record_event()
        ys = []
        for rec, W, st in zip(recs, Ws, streams):
            h, raw = rec._native(), C.c_void_p(st.cuda_stream)
            st.wait_event(ready)
            if rec._wkey != L.weights_key(W):
                Wc = W.detach().contiguous().float()
                L.check(rec._lib.lde_rnn_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), raw), h, "lde_rnn_set_weights_device", rnn=True)
                Wc.record_stream(st)
                rec._wkey = None
            y = torch.empty((B, rec.sizes[-1]), device=dev, dtype=torch.float32)
            y.record_stream(st)
            L.check(rec._lib.lde_rnn_forward(h, C.c_void_p(x.data_ptr()), T, B, C.c_void_p(y.data_ptr()), raw), h, "lde_rnn_forward", rnn=True)
            ys.append(y)
        for st in streams:
            x.record_stream(st)
        for st in streams:
            main.wait_stream(st)
        ctx.recs, ctx.streams, ctx.need_dx = recs, streams, x.requires_grad
        ctx.save_for_backward(x)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        recs, streams = ctx.recs, ctx.streams
        (x,) = ctx.saved_tensors
        dev = x.device
        T, B, _ = x.shape
        main = torch.cuda.current_stream(dev)
        dys = [dy.contiguous().float() for dy in dys]
        ready = main.record_event()
        dxs, dWs = [], []
        for rec, dy, st in zip(recs, dys, streams):               # the sweeps first …
            h, raw = rec._native(), C.c_void_p(st.cuda_stream)
            st.wait_event(ready)
            dx = torch.empty_like(x) if ctx.need_dx else None
            dy.record_stream(st)
            x.record_stream(st)
            if dx is not None:
                dx.record_stream(st)
            L.check(rec._lib.lde_rnn_backward_dx(h, C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), T, B,
                                                 C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(), raw), h, "lde_rnn_backward_dx", rnn=True)
            dxs.append(dx)
        for rec, st in zip(recs, streams):                        # … then the weight-gradient tails
            dW = torch.empty((rec.num_weights,), device=dev, dtype=torch.float32)     # written, not accumulated (set_accumulate(0))
            dW.record_stream(st)
            if L.dw_stream is not None:
                dW.record_stream(L.dw_stream)
            L.check(rec._lib.lde_rnn_backward_dw(rec._native(), C.c_void_p(dW.data_ptr()), C.c_void_p(st.cuda_stream)), rec._native(),
                    "lde_rnn_backward_dw", rnn=True)
            dWs.append(dW)
        for st in streams:
            main.wait_stream(st)
        dx = None
        if ctx.need_dx:
            dx = dxs[0]
            for d in dxs[1:]:
                dx = dx + d
        return (None, dx, *dWs)


class _RecurrentLaunchGroupFn(torch.autograd.Function):
    """(y_1, …, y_n) = (stack_1(x), …, stack_n(x)) on ONE stream through lde_rnn_group_forward / lde_rnn_group_backward: every stage of
    the call — the sweeps, the weight-gradient products of all (stack, cell) pairs, their fixed-order sums, the initial-state sums — is
    one launch (round 3). The same kernels on the same arguments per stack as _RecurrentFn: bit-equal results
    (tests/test_gpu_rnn.py::test_launch_grouped_stacks_equal_separate_calls). Inside a captured step this replaces the three side
    streams of _run_concurrently, whose kernels started 15 and 60 µs apart and whose weight-gradient tails queued behind each other."""

    @staticmethod
    def forward(ctx, recs, x, *Ws):
        if not x.is_cuda:
            raise L.LdeError("Recurrent needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        n = len(recs)
        dev = x.device
        stream = L.raw_stream(dev.index)
        T, B, _ = x.shape
        hs = []
        for rec, W in zip(recs, Ws):
            h = rec._native()
            if rec._wkey != L.weights_key(W):
                Wc = W.detach().contiguous().float()
                L.check(rec._lib.lde_rnn_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h, "lde_rnn_set_weights_device", rnn=True)
                rec._wkey = None
            hs.append(h)
        lib = recs[0]._lib
        ys = [torch.empty((B, rec.sizes[-1]), device=dev, dtype=torch.float32) for rec in recs]
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
        ctx.c_handles = arr([h.value for h in hs])
        train = any(ctx.needs_input_grad[1:])          # a pullback will follow: the sweeps keep their records (lde_rnn_forward_train)
        fwd = lib.lde_rnn_group_forward_train if train else lib.lde_rnn_group_forward
        rc = fwd(n, ctx.c_handles, arr([x.data_ptr()] * n), T, B, arr([y.data_ptr() for y in ys]), stream)
        L.check(rc, hs[0], "lde_rnn_group_forward", rnn=True)
        ctx.recs, ctx.need_dx = recs, x.requires_grad
        ctx.save_for_backward(x)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        recs = ctx.recs
        (x,) = ctx.saved_tensors
        n = len(recs)
        dev = x.device
        T, B, _ = x.shape
        stream = L.raw_stream(dev.index)
        dys = [dy.contiguous().float() for dy in dys]
        dxs = [torch.empty_like(x) for _ in recs] if ctx.need_dx else None
        dWs = [torch.empty((rec.num_weights,), device=dev, dtype=torch.float32) for rec in recs]     # written, not accumulated (set_accumulate(0))
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
        lib = recs[0]._lib
        rc = lib.lde_rnn_group_backward(n, ctx.c_handles, arr([x.data_ptr()] * n), arr([d.data_ptr() for d in dys]), T, B,
                                        arr([d.data_ptr() for d in dxs]) if dxs is not None else None, arr([d.data_ptr() for d in dWs]), stream)
        L.check(rc, recs[0]._native(), "lde_rnn_group_backward", rnn=True)
        dx = None
        if dxs is not None:
            dx = dxs[0]
            for d in dxs[1:]:
                dx = dx + d
        return (None, dx, *dWs)


class _GokuEncoderFn(torch.autograd.Function):
    """(μ_z₀, μ_θ, log σ²_z₀, log σ²_θ) = latent_in(pattern_extractor(feature_extractor(x))) for GOKU as ONE autograd node (round 3) —
    the same library calls on the same values as the separate nodes (so the same bits: tests/test_gpu_rnn.py::
    test_fused_encoder_node_equals_separate_nodes), minus the launches autograd and the tensor plumbing put between them in a
    training step:
      * the two LSTM stacks write the halves of vcat(pe_forward, pe_backward) [REF src/models/GOKU.jl:47] in place
        (lde_rnn_group_forward_ld) and read the halves of its gradient in place — no concatenation, no two strided copies;
      * a stack's output feeds the μ head and the log σ² head [REF src/models/GOKU.jl:61-72]: the pullback of the stacks takes both
        heads' input gradients as the two sources of its output gradient (lde_rnn_group_backward_ld) — no two additions;
      * the frames' features feed the three stacks [REF src/models/GOKU.jl:32-51]: the feature extractor's pullback takes their three
        input gradients as the sources of its output gradient (lde_chain_backward_saved_sum) — no two additions over [32 × B·T].
    Seven launches of ≈ 5 µs each per step. Inputs: (fe, pes, lis), x (T·B, pixels) batch-major, then the flat weights of fe, the three
    stacks and the four heads. Outputs batch-major: (B, ·)."""

    @staticmethod
    def forward(ctx, mods, x, *Ws):
        fe, pes, lis, (T, B) = mods
        if not x.is_cuda:
            raise L.LdeError("the encoder needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        dev = x.device
        stream = L.raw_stream(dev.index)
        N = T * B
        W_fe, W_pes, W_lis = Ws[0], Ws[1:4], Ws[4:8]
        # weights not handed over by refresh_weights() since the parameters last changed
        h_fe = fe._native()
        lib = fe._lib
        if fe._wkey != L.weights_key(W_fe):
            Wc = W_fe.detach().contiguous().float()
            L.check(lib.lde_chain_set_weights_device(h_fe, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h_fe, "lde_chain_set_weights_device", chain=True)
            fe._wkey = None
        h_pes, h_lis = [], []
        for rec, W in zip(pes, W_pes):
            h = rec._native()
            if rec._wkey != L.weights_key(W):
                Wc = W.detach().contiguous().float()
                L.check(lib.lde_rnn_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h, "lde_rnn_set_weights_device", rnn=True)
                rec._wkey = None
            h_pes.append(h)
        for c, W in zip(lis, W_lis):
            h = c._native()
            if c._wkey != L.weights_key(W):
                Wc = W.detach().contiguous().float()
                L.check(lib.lde_chain_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h, "lde_chain_set_weights_device", chain=True)
                c._wkey = None
            h_lis.append(h)
        train = any(ctx.needs_input_grad[1:])
        f32 = dict(device=dev, dtype=torch.float32)
        # 1. the frames' features, (T, B, f) = the stacks' [f × B × T]
        y_fe = torch.empty((N, fe.sizes[-1]), **f32)
        sv_fe = torch.empty((int(lib.lde_chain_saved_floats(h_fe, N)),), **f32) if train else None
        if train:
            L.check(lib.lde_chain_forward_save(h_fe, C.c_void_p(x.data_ptr()), N, C.c_void_p(y_fe.data_ptr()), C.c_void_p(sv_fe.data_ptr()), stream),
                    h_fe, "lde_chain_forward_save", chain=True)
        else:
            L.check(lib.lde_chain_forward(h_fe, C.c_void_p(x.data_ptr()), N, C.c_void_p(y_fe.data_ptr()), stream), h_fe, "lde_chain_forward", chain=True)
        # 2. the three stacks; the θ stacks write the two column blocks of their vcat
        hz, hf, hb = (rec.sizes[-1] for rec in pes)
        y_z0 = torch.empty((B, hz), **f32)
        y_th = torch.empty((B, hf + hb), **f32)
        arr3 = lambda ptrs: (C.c_void_p * 3)(*ptrs)
        ctx.c_pes = arr3([h.value for h in h_pes])
        ctx.c_ld = (C.c_int32 * 3)(hz, hf + hb, hf + hb)
        rc = lib.lde_rnn_group_forward_ld(3, ctx.c_pes, arr3([y_fe.data_ptr()] * 3), T, B,
                                          arr3([y_z0.data_ptr(), y_th.data_ptr(), y_th.data_ptr() + 4 * hf]), ctx.c_ld, int(train), stream)
        L.check(rc, h_pes[0], "lde_rnn_group_forward_ld", rnn=True)
        # 3. the four heads
        xs_li = (y_z0, y_th, y_z0, y_th)
        outs = [torch.empty((B, c.sizes[-1]), **f32) for c in lis]
        sv_li = [torch.empty((int(lib.lde_chain_saved_floats(h, B)),), **f32) for h in h_lis] if train else None
        arr4 = lambda ptrs: (C.c_void_p * 4)(*ptrs)
        ctx.c_lis = arr4([h.value for h in h_lis])
        ctx.c_Ns = (C.c_int64 * 4)(B, B, B, B)
        rc = lib.lde_chain_group_forward_save(4, ctx.c_lis, arr4([t.data_ptr() for t in xs_li]), ctx.c_Ns, arr4([o.data_ptr() for o in outs]),
                                              arr4([t.data_ptr() for t in sv_li]) if train else None, stream)
        L.check(rc, h_lis[0], "lde_chain_group_forward_save", chain=True)
        ctx.mods, ctx.train, ctx.TB, ctx.need_dx = mods, train, (T, B), bool(ctx.needs_input_grad[1])
        if train:
            ctx.save_for_backward(x, y_fe, sv_fe, y_z0, y_th, *outs, *sv_li)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        fe, pes, lis, _ = ctx.mods
        T, B = ctx.TB
        N = T * B
        t = ctx.saved_tensors
        x, y_fe, sv_fe, y_z0, y_th = t[:5]
        outs, sv_li = t[5:9], t[9:13]
        lib = fe._lib
        dev = x.device
        stream = L.raw_stream(dev.index)
        f32 = dict(device=dev, dtype=torch.float32)
        arr3 = lambda ptrs: (C.c_void_p * 3)(*ptrs)
        arr4 = lambda ptrs: (C.c_void_p * 4)(*ptrs)
        douts = [d.contiguous().float() for d in douts]
        # 1. the heads: input gradients per head (two per stack output), weight gradients written (set_accumulate(0))
        xs_li = (y_z0, y_th, y_z0, y_th)
        d_li = [torch.empty_like(xi) for xi in xs_li]
        dW_li = [torch.empty((c.num_weights,), **f32) for c in lis]
        rc = lib.lde_chain_group_backward_saved(4, ctx.c_lis, arr4([t_.data_ptr() for t_ in xs_li]), arr4([o.data_ptr() for o in outs]),
                                                arr4([d.data_ptr() for d in douts]), arr4([s_.data_ptr() for s_ in sv_li]), ctx.c_Ns,
                                                arr4([d.data_ptr() for d in d_li]), arr4([d.data_ptr() for d in dW_li]), stream)
        L.check(rc, lis[0]._native(), "lde_chain_group_backward_saved", chain=True)
        # 2. the stacks: output gradient = μ head's + log σ² head's input gradient, the θ stacks' as column blocks of the (B, 2h) arrays
        hf = pes[1].sizes[-1]
        dz_a, dth_a, dz_b, dth_b = d_li
        dxs = [torch.empty_like(y_fe) for _ in pes]
        dW_pe = [torch.empty((rec.num_weights,), **f32) for rec in pes]
        rc = lib.lde_rnn_group_backward_ld(3, ctx.c_pes, arr3([y_fe.data_ptr()] * 3),
                                           arr3([dz_a.data_ptr(), dth_a.data_ptr(), dth_a.data_ptr() + 4 * hf]),
                                           arr3([dz_b.data_ptr(), dth_b.data_ptr(), dth_b.data_ptr() + 4 * hf]), ctx.c_ld, T, B,
                                           arr3([d.data_ptr() for d in dxs]), arr3([d.data_ptr() for d in dW_pe]), stream)
        L.check(rc, pes[0]._native(), "lde_rnn_group_backward_ld", rnn=True)
        # 3. the feature extractor: output gradient = the three stacks' input gradients
        dx = torch.empty_like(x) if ctx.need_dx else None
        dW_fe = torch.empty((fe.num_weights,), **f32)
        if L.dw_stream is not None:
            for dW in (dW_fe, *dW_li):
                dW.record_stream(L.dw_stream)
        rc = lib.lde_chain_backward_saved_sum(fe._native(), C.c_void_p(x.data_ptr()), C.c_void_p(y_fe.data_ptr()), 3, arr3([d.data_ptr() for d in dxs]),
                                              C.c_void_p(sv_fe.data_ptr()), N, C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(),
                                              C.c_void_p(dW_fe.data_ptr()), stream)
        L.check(rc, fe._native(), "lde_chain_backward_saved_sum", chain=True)
        return (None, dx, dW_fe, *dW_pe, *dW_li)


_ENCODER_FUSED = True   # encode() without branch streams (what a captured step runs): the GOKU encoder as one autograd node


def _encode_goku_fused(encoder: Encoder, x):
    """encode() for GOKU through _GokuEncoderFn, or None when the layers are not the kinds it handles."""
    fe, pes, lis = encoder.feature_extractor, encoder.pattern_extractor, encoder.latent_in
    ok = (isinstance(fe, Chain) and x.dim() == 3 and x.is_cuda and isinstance(pes, (tuple, list)) and len(pes) == 3
          and all(isinstance(m, Recurrent) for m in pes) and isinstance(lis, (tuple, list)) and len(lis) == 4
          and all(isinstance(m, Chain) for m in lis) and len({id(m) for m in pes}) == 3 and len({id(m) for m in lis}) == 4
          and all(m.sizes[0] == fe.sizes[-1] for m in pes)
          and lis[0].sizes[0] == lis[1].sizes[0] == pes[0].sizes[-1]
          and lis[2].sizes[0] == lis[3].sizes[0] == pes[1].sizes[-1] + pes[2].sizes[-1])
    if not ok:
        return None
    n_in, B, T = x.shape
    buf = x.permute(2, 1, 0).reshape(T * B, n_in).contiguous().float()       # (T·B, pixels): in place when x is a view of a (T, B, pixels) buffer
    li_mu_z0, li_ls_z0, li_mu_th, li_ls_th = lis
    heads = (li_mu_z0, li_mu_th, li_ls_z0, li_ls_th)                       # the order apply_latent_in issues them in
    mu_z0, mu_th, ls_z0, ls_th = _GokuEncoderFn.apply((fe, tuple(pes), heads, (T, B)), buf, fe.flat_weights(), *[m.flat_weights() for m in pes],
                                                      *[m.flat_weights() for m in heads])
    return (mu_z0.t(), mu_th.t()), (ls_z0.t(), ls_th.t())


class Recurrent(torch.nn.Module):
    """A stack of cells of one kind applied to the frames of x [in, B, T] (reverse=True: frames T..1), returning the output
    after the last frame, [h_last, B] — `[pe(x) for x in frames][end]` followed by `Flux.reset!`  [REF GOKU.jl:40-47]."""

    def __init__(self, *cells, reverse: bool = False):
        super().__init__()
        if not cells or len({type(c) for c in cells}) != 1 or len({c.code for c in cells}) != 1:
            raise TypeError("Recurrent takes cells of one kind (all RNN with one activation, or all LSTM)")
        for a, b in zip(cells[:-1], cells[1:]):
            if a.n_out != b.n_in:
                raise ValueError("cell sizes do not chain")
        self.cells = list(cells)
        self.reverse = bool(reverse)
        self.sizes = [cells[0].n_in] + [c.n_out for c in cells]
        self.code = cells[0].code
        self.num_weights = sum(c.num_weights() for c in cells)
        # ONE flat parameter in Flux.destructure order (per cell vec(Wi), vec(Wh), b, state0): what lde_rnn_* consumes
        off = 0
        parts = []
        for c in cells:
            parts.append(c.flat().detach())
            c._off = off
            off += c.num_weights()
        self.theta = torch.nn.Parameter(torch.cat(parts).float())
        for c in cells:
            c._owner, c._init = self, None
        self._handle, self._lib = None, None
        self._wkey = None          # set by _lib.refresh_weights: the parameter value the handle already holds
        self._is_recurrent = True

    def _native(self):
        if self._handle is None:
            self._lib = L.load()
            d = L.RnnDesc()
            d.abi_version, d.cell, d.n_layers, d.reverse = L.LDE_ABI_VERSION, self.code, len(self.cells), int(self.reverse)
            for i, s in enumerate(self.sizes):
                d.sizes[i] = s
            h = C.c_void_p()
            rc = self._lib.lde_rnn_create(C.byref(d), C.byref(h))
            if rc != 0:
                try:
                    L.check(rc, h if h else None, "lde_rnn_create", rnn=True)
                finally:
                    if h:
                        self._lib.lde_rnn_destroy(h)
            L.check(self._lib.lde_rnn_set_accumulate(h, 0), h, "lde_rnn_set_accumulate", rnn=True)   # the pullback hands autograd a fresh gradient
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None and self._lib is not None:
                self._lib.lde_rnn_destroy(self._handle)
        except Exception:
            pass

    def flat_weights(self) -> torch.Tensor:
        return self.theta

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [in, B, T] → [h_last, B]."""
        if x.dim() != 3:
            raise ValueError("Recurrent takes [in, B, T] arrays")
        buf = x.permute(2, 1, 0).contiguous().float()              # (T, B, in) == column-major [in × B × T]
        return _RecurrentFn.apply(self, buf, self.flat_weights()).t()


class Encoder:
    """Encoder(model_type, (feature_extractor, pattern_extractor, latent_in))  [REF src/models/LatentDiffEqModel.jl:30-61]."""

    def __init__(self, model_type, encoder_layers):
        self.model_type = model_type
        self.feature_extractor, self.pattern_extractor, self.latent_in = encoder_layers


def apply_feature_extractor(encoder: Encoder, x):
    """fe_out = encoder.feature_extractor(x), x [pixels, B, T]  [REF src/models/GOKU.jl:19]."""
    return encoder.feature_extractor(x)


_side_streams = {}
_RNN_GROUP = False           # encode(): the three stacks as ONE autograd node on raw side streams, sweeps issued before
                                                                       # the weight-gradient tails. Opt-in: measured 1.80 vs 1.64 ms per goku_step at B = 256 (six
                                                                       # alternations on one box) — the z₀ branch's latent_in chains then run after the join instead of
                                                                       # beside the θ branch's longer stacks, and the per-tensor record_stream calls cost the host more
                                                                       # than the two autograd nodes they save
_RNN_LAUNCH_GROUP = True   # apply_pattern_extractor (the path without branch streams — what a captured step
                                                                           # runs): the stacks through lde_rnn_group_* on one stream instead of three side streams
_STACKS_FIRST = True       # encode(): issue the three recurrent stacks before the latent_in chains
_BRANCH_STREAMS = True   # encode(): keep the z₀ / θ branches on their own streams (diagnostic switch)


def _run_concurrently(stacks, x):
    """The three stacks of the GOKU pattern extractor are independent and each fills only B/16 workgroups: run them on
    side HIP streams (autograd replays each pullback on the stream its forward ran on, so those overlap too)."""
    if not x.is_cuda or len(stacks) < 2:
        return [m(x) for m in stacks]
    dev = x.device
    main = torch.cuda.current_stream(dev)
    streams = _side_streams.setdefault((dev.index, len(stacks)), [torch.cuda.Stream(dev) for _ in stacks])
    ready = main.record_event()
    outs = []
    for m, st in zip(stacks, streams):
        st.wait_event(ready)
        with torch.cuda.stream(st):
            y = m(x)
        x.record_stream(st)
        outs.append(y)
    for st, y in zip(streams, outs):
        main.wait_stream(st)
        y.record_stream(main)
    return outs


def apply_pattern_extractor(encoder: Encoder, fe_out):
    """[REF src/models/GOKU.jl:32-51]: pe_z₀ on the reversed frames; pe_θ forward ⊕ pe_θ backward (reversed frames).
    [REF src/models/LatentODE.jl:24-33]: one stack on the reversed frames."""
    if isinstance(encoder.model_type, GOKU):
        pes = encoder.pattern_extractor
        if _RNN_LAUNCH_GROUP and fe_out.is_cuda and fe_out.dim() == 3 and 2 <= len(pes) <= 3 and all(isinstance(m, Recurrent) for m in pes):
            buf = fe_out.permute(2, 1, 0).contiguous().float()            # (T, B, in) == column-major [in × B × T], once for the three stacks
            ys = _RecurrentLaunchGroupFn.apply(tuple(pes), buf, *[m.flat_weights() for m in pes])
            outs = [y.t() for y in ys]
        else:
            outs = _run_concurrently(pes, fe_out)
        return outs[0], torch.cat([outs[1].t(), outs[2].t()], dim=1).t()   # vcat, kept in batch-major memory (what latent_in reads)
    if isinstance(encoder.model_type, LatentODE):
        return encoder.pattern_extractor(fe_out)
    raise TypeError(f"no apply_pattern_extractor method for model type {type(encoder.model_type).__name__}")


def apply_latent_in(encoder: Encoder, pe_out):
    """(μ, logσ²)  [REF src/models/GOKU.jl:61-72], [REF src/models/LatentODE.jl:35-43]."""
    if isinstance(encoder.model_type, GOKU):
        pe_z0, pe_th = pe_out
        li_mu_z0, li_ls_z0, li_mu_th, li_ls_th = encoder.latent_in
        heads = [(li_mu_z0, pe_z0), (li_mu_th, pe_th), (li_ls_z0, pe_z0), (li_ls_th, pe_th)]
        from .chain import apply_chains_grouped      # the four heads as one autograd node, one launch per stage
        mu_z0, mu_th, ls_z0, ls_th = apply_chains_grouped(heads)
        return (mu_z0, mu_th), (ls_z0, ls_th)
    li_mu, li_ls = encoder.latent_in
    from .chain import apply_chains_grouped
    mu, ls = apply_chains_grouped([(li_mu, pe_out), (li_ls, pe_out)])
    return mu, ls


def _encode_goku_group(encoder: Encoder, fe_out):
    """apply_latent_in(apply_pattern_extractor(fe_out)) for GOKU with the three stacks as one autograd node
    (_RecurrentGroupFn) and everything between the stacks and (μ, logσ²) kept in batch-major buffers — no transposed views in
    between, the θ branch's vcat [REF GOKU.jl:47] is a concatenation along the feature axis of (B, h) buffers. Same arithmetic
    as the two reference functions."""
    recs = tuple(encoder.pattern_extractor)
    li_mu_z0, li_ls_z0, li_mu_th, li_ls_th = encoder.latent_in
    buf = fe_out.permute(2, 1, 0).contiguous().float()            # (T, B, in) == column-major [in × B × T]
    y_z0, y_f, y_b = _RecurrentGroupFn.apply(recs, buf, *[m.flat_weights() for m in recs])
    y_th = torch.cat([y_f, y_b], dim=1)                           # (B, 2h)
    return ((li_mu_z0.apply_batch_major(y_z0).t(), li_mu_th.apply_batch_major(y_th).t()),
            (li_ls_z0.apply_batch_major(y_z0).t(), li_ls_th.apply_batch_major(y_th).t()))


def _encode_goku_branches(encoder: Encoder, fe_out):
    """apply_latent_in(apply_pattern_extractor(fe_out)) for GOKU with the two independent branches kept on their own HIP
    streams to the end: z₀ branch = pe_z₀ → (li_μ_z₀, li_logσ²_z₀); θ branch = (pe_θ forward, pe_θ backward) → vcat →
    (li_μ_θ, li_logσ²_θ). The z₀ branch's small chains then run beside the θ branch's longer LSTM stacks instead of after
    them (and so do their pullbacks: autograd replays every node on its forward stream). Same events as running only the
    three stacks on side streams; same arithmetic as the two reference functions.

    Allocator invariant (torch's caching allocator ties a block to the stream it was allocated on): every tensor that
    ESCAPES its `with torch.cuda.stream(...)` block is recorded on the stream that consumes it — fe_out on the three side
    streams, pe_b on sB, the four results on main. pe_z0, pe_f and pe_th are produced and consumed on ONE stream and are
    not returned; anything that starts using them elsewhere (e.g. returning pe_out for diagnostics) must add
    `record_stream` for that stream. tests/test_gpu_rnn.py::test_branch_streams_match_joined_streams runs both paths
    over changing (B, T) and compares outputs and gradients."""
    pe_z0_m, pe_f_m, pe_b_m = encoder.pattern_extractor
    li_mu_z0, li_ls_z0, li_mu_th, li_ls_th = encoder.latent_in
    dev = fe_out.device
    main = torch.cuda.current_stream(dev)
    sA, sB, sC = _side_streams.setdefault((dev.index, 3), [torch.cuda.Stream(dev) for _ in range(3)])
    ready = main.record_event()
    for st in (sA, sB, sC):
        st.wait_event(ready)
        fe_out.record_stream(st)
    # the three stacks first, the small chains after them: autograd replays the nodes in reverse order, so the pullback
    # enqueues the four cheap latent_in pullbacks first and then the three long recurrent pullbacks back to back — they
    # overlap fully instead of the last one starting when the host gets to it (≈ 0.1 ms of a 1.8 ms step)
    with torch.cuda.stream(sA):
        pe_z0 = pe_z0_m(fe_out)
        if not _STACKS_FIRST:
            mu_z0, ls_z0 = li_mu_z0(pe_z0), li_ls_z0(pe_z0)
    with torch.cuda.stream(sC):
        pe_b = pe_b_m(fe_out)
    with torch.cuda.stream(sB):
        pe_f = pe_f_m(fe_out)
    if _STACKS_FIRST:
        with torch.cuda.stream(sA):
            mu_z0, ls_z0 = li_mu_z0(pe_z0), li_ls_z0(pe_z0)
    with torch.cuda.stream(sB):
        sB.wait_stream(sC)
        pe_b.record_stream(sB)
        pe_th = torch.cat([pe_f.t(), pe_b.t()], dim=1).t()    # vcat [REF GOKU.jl:47] in batch-major memory: latent_in reads it in place
        mu_th, ls_th = li_mu_th(pe_th), li_ls_th(pe_th)
    main.wait_stream(sA)
    main.wait_stream(sB)
    main.wait_stream(sC)      # (already ordered through sB; joined explicitly so that a stream capture sees every fork return to its origin)
    for y in (mu_z0, ls_z0, mu_th, ls_th):
        y.record_stream(main)
    return (mu_z0, mu_th), (ls_z0, ls_th)


def encode(encoder: Encoder, x):
    """(μ, logσ²) = encoder(x)  [REF src/models/LatentDiffEqModel.jl:63-75]."""
    if isinstance(encoder.model_type, GOKU) and _ENCODER_FUSED and not _RNN_GROUP and not _BRANCH_STREAMS and torch.is_tensor(x) and x.is_cuda:
        out = _encode_goku_fused(encoder, x)
        if out is not None:
            return out
    fe_out = apply_feature_extractor(encoder, x)
    if isinstance(encoder.model_type, GOKU) and fe_out.is_cuda and _RNN_GROUP and isinstance(encoder.pattern_extractor, tuple) \
            and len(encoder.pattern_extractor) == 3 and all(isinstance(m, Recurrent) for m in encoder.pattern_extractor):
        return _encode_goku_group(encoder, fe_out)
    if isinstance(encoder.model_type, GOKU) and fe_out.is_cuda and _BRANCH_STREAMS:
        return _encode_goku_branches(encoder, fe_out)
    return apply_latent_in(encoder, apply_pattern_extractor(encoder, fe_out))


from .loss import sample  # noqa: E402,F401  (l̃ = μ + ε·exp(logσ²/2): lde_sample_forward / _backward)


def default_encoder_layers(model_type, input_dim: int, diffeq=None, hidden_dim_resnet: int = 200, rnn_input_dim: int = 32,
                           rnn_output_dim: int = None, latent_dim_z0: int = 16, latent_dim_theta: int = 16, device=None):
    """The encoder half of default_layers  [REF src/models/GOKU.jl:214-245], [REF src/models/LatentODE.jl:108-128]."""
    fe = Chain(Dense(input_dim, hidden_dim_resnet, "relu"),
               SkipConnection(Dense(hidden_dim_resnet, hidden_dim_resnet, "relu")),
               SkipConnection(Dense(hidden_dim_resnet, hidden_dim_resnet, "relu")),
               Dense(hidden_dim_resnet, rnn_input_dim, "relu"))
    if isinstance(model_type, GOKU):
        ro = rnn_output_dim or 16
        pe = (Recurrent(RNN(rnn_input_dim, ro, "relu"), RNN(ro, ro, "relu"), reverse=True),
              Recurrent(LSTM(rnn_input_dim, ro), LSTM(ro, ro), reverse=False),
              Recurrent(LSTM(rnn_input_dim, ro), LSTM(ro, ro), reverse=True))
        li = (Chain(Dense(ro, latent_dim_z0)), Chain(Dense(ro, latent_dim_z0)),
              Chain(Dense(2 * ro, latent_dim_theta)), Chain(Dense(2 * ro, latent_dim_theta)))
    elif isinstance(model_type, LatentODE):
        ro = rnn_output_dim or 32
        pe = Recurrent(RNN(rnn_input_dim, ro, "relu"), RNN(ro, ro, "relu"), reverse=True)
        li = (Chain(Dense(ro, diffeq.latent_dim_in)), Chain(Dense(ro, diffeq.latent_dim_in)))
    else:
        raise TypeError("default_encoder_layers: GOKU_basic() or LatentODE()")
    if device is not None:
        fe = fe.to(device)
        pe = tuple(m.to(device) for m in pe) if isinstance(pe, tuple) else pe.to(device)
        li = tuple(m.to(device) for m in li)
    return fe, pe, li
