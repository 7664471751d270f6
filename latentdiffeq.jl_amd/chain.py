"""Host-side mirror of the dense chains either side of the solve (scope row f-1):

    reference                                                                      here
    -----------------------------------------------------------------------------  ---------------------------------
    Dense(in, out, act; init)                        [Flux; REF GOKU.jl:252-266]    Dense(in, out, act)
    SkipConnection(Dense(...), +)                    [REF GOKU.jl:262-263]          SkipConnection(Dense(...))
    Chain(l1, SkipConnection(l2,+), ...)             [REF GOKU.jl:252-266]          Chain(l1, SkipConnection(l2), ...)
    apply_latent_out(decoder, l̃)                     [REF GOKU.jl:83-91], [REF LatentODE.jl:53]   same
    apply_reconstructor(decoder, ẑ)                  [REF GOKU.jl:148], [REF LatentODE.jl:80]     same
    decoder(l̃, t) → (x̂, ẑ, l̂)                       [REF LatentDiffEqModel.jl:101-113]           decode(decoder, l̃, t)
    default_layers(GOKU_basic(), input_dim, diffeq)  [REF GOKU.jl:201-273] (decoder part)         default_decoder_layers

Everything under `Chain.__call__` is liblde.so (lde_chain_forward / lde_chain_backward, include/lde.h). Arrays keep the
reference's shapes ([rows, B] or [rows, B, T]); their memory is the reference's column-major layout, i.e. transposed
views of batch-major torch buffers, exactly like `diffeq_layer` (api.py).
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Optional, Sequence

import torch

from . import _lib as L
from .api import GOKU, LatentODE, Decoder, diffeq_layer

_ACT = {"identity": L.CACT_IDENTITY, "relu": L.CACT_RELU, "tanh": L.CACT_TANH, "sigmoid": L.CACT_SIGMOID, "σ": L.CACT_SIGMOID,
        "softplus": L.CACT_SOFTPLUS}


class Dense:
    """Dense(in, out, act): y = act.(W*x .+ b). Default init = Flux.kaiming_uniform(gain = 1/√3) ⇒ U(±1/√fan_in),
    zero bias  [REF src/models/GOKU.jl:204]. Inside a `Chain` the layer's weights live in the chain's single flat parameter
    `theta` (Flux.destructure order): `weight` [out, in] and `bias` [out] are views of it."""

    def __init__(self, n_in: int, n_out: int, act: str = "identity"):
        if act not in _ACT:
            raise ValueError(f"unknown activation {act!r}")
        self.n_in, self.n_out, self.act = n_in, n_out, act
        bound = 1.0 / math.sqrt(n_in)
        self._w0 = torch.empty(n_out, n_in).uniform_(-bound, bound)
        self._b0 = torch.zeros(n_out)
        self._owner, self._off = None, 0

    @property
    def weight(self) -> torch.Tensor:
        if self._owner is None:
            return self._w0
        n = self.n_in * self.n_out
        return self._owner.theta[self._off:self._off + n].view(self.n_in, self.n_out).t()     # vec(W) column-major [out×in]

    @property
    def bias(self) -> torch.Tensor:
        if self._owner is None:
            return self._b0
        o = self._off + self.n_in * self.n_out
        return self._owner.theta[o:o + self.n_out]

    def grads(self):
        """(∂L/∂W [out, in], ∂L/∂b [out]) as views of the owner's theta.grad."""
        g, n = self._owner.theta.grad, self.n_in * self.n_out
        return g[self._off:self._off + n].view(self.n_in, self.n_out).t(), g[self._off + n:self._off + n + self.n_out]


class SkipConnection:
    """SkipConnection(layer, +): y = layer(x) + x  [REF src/models/GOKU.jl:262-263]."""

    def __init__(self, layer: Dense):
        if layer.n_in != layer.n_out:
            raise ValueError("SkipConnection(+) needs in == out")
        self.layer = layer


class _ChainFn(torch.autograd.Function):
    """y = chain(x) on batch-major buffers: x (N, in) → y (N, out); backward = lde_chain_backward[_saved]. When a gradient
    will be asked for, the forward call keeps the hidden activations in a buffer of its own (lde_chain_forward_save) and the
    pullback reads them instead of recomputing the hidden layers."""

    @staticmethod
    def forward(ctx, chain: "Chain", x: torch.Tensor, W: torch.Tensor):
        if not x.is_cuda:
            raise L.LdeError("Chain needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        h = chain._native()
        lib = chain._lib
        stream = L.raw_stream(x.device.index)
        key = L.weights_key(W)
        if chain._wkey != key:          # not handed over by refresh_weights() since the parameter last changed
            Wc = W.detach().contiguous().float()
            L.check(lib.lde_chain_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h,
                    "lde_chain_set_weights_device", chain=True)
            chain._wkey = None
        N = x.shape[0]
        y = torch.empty((N, chain.sizes[-1]), device=x.device, dtype=torch.float32)
        # (inside Function.forward grad mode is always off — torch.is_grad_enabled() is False here whether or not a pullback will
        #  follow; ctx.needs_input_grad is what says so. Until round 3 this line asked is_grad_enabled() and the training variant
        #  was never taken through autograd: every pullback recomputed the hidden layers.)
        train = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        saved = None
        if train:
            saved = torch.empty((int(lib.lde_chain_saved_floats(h, N)),), device=x.device, dtype=torch.float32)
            L.check(lib.lde_chain_forward_save(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()),
                                               C.c_void_p(saved.data_ptr()), stream), h, "lde_chain_forward_save", chain=True)
        else:
            L.check(lib.lde_chain_forward(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()), stream), h,
                    "lde_chain_forward", chain=True)
        ctx.chain = chain
        ctx.need_dx = x.requires_grad
        ctx.has_saved = saved is not None
        ctx.save_for_backward(x, y, saved if saved is not None else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        chain = ctx.chain
        h = chain._native()
        lib = chain._lib
        x, y, saved = ctx.saved_tensors
        dy = dy.contiguous().float()
        stream = L.raw_stream(x.device.index)
        dx = torch.empty_like(x) if ctx.need_dx else None
        dW = torch.empty((chain.num_weights,), device=x.device, dtype=torch.float32)     # written, not accumulated (set_accumulate(0))
        if L.dw_stream is not None:
            dW.record_stream(L.dw_stream)          # written on the weight-gradient stream (set_async_weight_gradients)
        pdx = C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p()
        if ctx.has_saved:
            L.check(lib.lde_chain_backward_saved(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(dy.data_ptr()),
                                                 C.c_void_p(saved.data_ptr()), x.shape[0], pdx, C.c_void_p(dW.data_ptr()), stream),
                    h, "lde_chain_backward_saved", chain=True)
        else:
            L.check(lib.lde_chain_backward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(dy.data_ptr()),
                                           x.shape[0], pdx, C.c_void_p(dW.data_ptr()), stream), h, "lde_chain_backward", chain=True)
        return None, dx, dW


class _ChainMseFn(torch.autograd.Function):
    """(loss, y) = (base + scale·Σ (y − target)², y) with y = chain(x): the reconstructor under reconstruction_loss
    [REF examples/pendulum_friction-less/model_train.jl:225-238] as one autograd node. Forward: the calls of _ChainFn and
    loss._MseAddFn. Pullback: lde_chain_backward_saved_mse — the loss's cotangent 2·g·scale·(y − target) is formed where the chain's
    pullback reads its output gradient, instead of lde_mse_backward writing it as an (N, out) array for lde_chain_backward_saved to read
    back: one launch and three passes over the largest array of a GOKU step less; the same values (tests/test_gpu_chain.py::
    test_reconstructor_under_mse_as_one_node). A cotangent of y itself (y used elsewhere too) is added in."""

    @staticmethod
    def forward(ctx, chain: "Chain", x, W, target, scale: float, base, want_y: bool = True):
        if not x.is_cuda:
            raise L.LdeError("Chain needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        h = chain._native()
        lib = chain._lib
        stream = L.raw_stream(x.device.index)
        if chain._wkey != L.weights_key(W):
            Wc = W.detach().contiguous().float()
            L.check(lib.lde_chain_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h, "lde_chain_set_weights_device", chain=True)
            chain._wkey = None
        N = x.shape[0]
        train = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        saved = torch.empty((int(lib.lde_chain_saved_floats(h, N)),), device=x.device, dtype=torch.float32) if train else None
        ctx.delta = False
        if _RECON_MSE_FWD and train and chain.dtype == "bf16" and chain.sizes[-1] % 8 == 0:
            # bf16 chains: the forward launch leaves δ_L′ for the pullback (lde_chain_forward_save_mse_delta) — the pullback reads neither x̂
            # nor the frames, and x̂ itself is stored only when the caller wants it
            y = torch.empty((N, chain.sizes[-1]), device=x.device, dtype=torch.float32) if want_y else None
            ws = torch.empty(int(lib.lde_chain_mse_scratch_floats(h, N)) + 1, device=x.device, dtype=torch.float32)
            L.check(lib.lde_chain_forward_save_mse_delta(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()) if want_y else C.c_void_p(),
                                                         C.c_void_p(saved.data_ptr()), C.c_void_p(target.data_ptr()), scale,
                                                         C.c_void_p(base.data_ptr()) if base is not None else C.c_void_p(), C.c_void_p(ws.data_ptr()),
                                                         C.c_void_p(ws.data_ptr() + 4), stream), h, "lde_chain_forward_save_mse_delta", chain=True)
            ctx.chain, ctx.scale, ctx.has_base, ctx.need_dx, ctx.has_saved, ctx.delta = chain, scale, base is not None, x.requires_grad, True, True
            ctx.save_for_backward(x, y if want_y else x.new_empty(0), saved, target)
            ctx.set_materialize_grads(False)
            return ws[0], (y if want_y else x.new_empty(0))
        y = torch.empty((N, chain.sizes[-1]), device=x.device, dtype=torch.float32)
        if _RECON_MSE_FWD:      # the loss value from the forward launch itself: squares summed in the last layer's epilogue
            ws = torch.empty(int(lib.lde_chain_mse_scratch_floats(h, N)) + 1, device=x.device, dtype=torch.float32)     # [0]: the result, then tile sums
            L.check(lib.lde_chain_forward_save_mse(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()),
                                                   C.c_void_p(saved.data_ptr()) if train else C.c_void_p(), C.c_void_p(target.data_ptr()), scale,
                                                   C.c_void_p(base.data_ptr()) if base is not None else C.c_void_p(), C.c_void_p(ws.data_ptr()),
                                                   C.c_void_p(ws.data_ptr() + 4), stream), h, "lde_chain_forward_save_mse", chain=True)
            ctx.chain, ctx.scale, ctx.has_base, ctx.need_dx, ctx.has_saved = chain, scale, base is not None, x.requires_grad, train
            ctx.save_for_backward(x, y, saved if train else x.new_empty(0), target)
            ctx.set_materialize_grads(False)
            return ws[0], y
        if train:
            L.check(lib.lde_chain_forward_save(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()), C.c_void_p(saved.data_ptr()), stream),
                    h, "lde_chain_forward_save", chain=True)
        else:
            L.check(lib.lde_chain_forward(h, C.c_void_p(x.data_ptr()), N, C.c_void_p(y.data_ptr()), stream), h, "lde_chain_forward", chain=True)
        ws = torch.empty(L.LOSS_SCRATCH_FLOATS + 1, device=x.device, dtype=torch.float32)
        if base is not None:
            L.check(lib.lde_mse_forward_add(C.c_void_p(target.data_ptr()), C.c_void_p(y.data_ptr()), y.numel(), scale, C.c_void_p(base.data_ptr()),
                                            C.c_void_p(ws.data_ptr()), C.c_void_p(ws.data_ptr() + 4), stream), None, "lde_mse_forward_add")
        else:
            L.check(lib.lde_mse_forward(C.c_void_p(target.data_ptr()), C.c_void_p(y.data_ptr()), y.numel(), scale, C.c_void_p(ws.data_ptr()),
                                        C.c_void_p(ws.data_ptr() + 4), stream), None, "lde_mse_forward")
        ctx.chain, ctx.scale, ctx.has_base, ctx.need_dx, ctx.has_saved = chain, scale, base is not None, x.requires_grad, train
        ctx.save_for_backward(x, y, saved if train else x.new_empty(0), target)
        ctx.set_materialize_grads(False)
        return ws[0], y

    @staticmethod
    def backward(ctx, g, dy):
        chain = ctx.chain
        h = chain._native()
        lib = chain._lib
        x, y, saved, target = ctx.saved_tensors
        stream = L.raw_stream(x.device.index)
        N = x.shape[0]
        dx = torch.empty_like(x) if ctx.need_dx else None
        dW = torch.empty((chain.num_weights,), device=x.device, dtype=torch.float32)
        if L.dw_stream is not None:
            dW.record_stream(L.dw_stream)
        pdx = C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p()
        psv = C.c_void_p(saved.data_ptr()) if ctx.has_saved else C.c_void_p()
        if dy is not None:
            dy = dy.contiguous().float()
        if g is None:          # only y was used downstream: the plain pullback
            if y.numel() == 0:
                raise L.LdeError("decode_loss(want_x_hat=False): x̂ was not kept, so it cannot carry a cotangent of its own")
            if dy is None:
                dy = torch.zeros_like(y)
            fn = lib.lde_chain_backward_saved if ctx.has_saved else None
            if fn is not None:
                L.check(fn(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(dy.data_ptr()), psv, N, pdx, C.c_void_p(dW.data_ptr()), stream),
                        h, "lde_chain_backward_saved", chain=True)
            else:
                L.check(lib.lde_chain_backward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(dy.data_ptr()), N, pdx,
                                               C.c_void_p(dW.data_ptr()), stream), h, "lde_chain_backward", chain=True)
            return None, dx, dW, None, None, None, None
        g = g.contiguous().float()
        # δ_L′ staged in the chain's workspace by THIS node's forward (lde_chain_delta_is_staged: another forward of the same decoder before this
        # pullback re-stages it — then the two-pass pullback below runs from x̂ and the frames instead): no pass over x̂ / the frames; g
        # multiplies dx and dW at the end
        if ctx.delta and dy is None and lib.lde_chain_delta_is_staged(h, psv, N):
            L.check(lib.lde_chain_backward_saved_delta(h, C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), psv, N, pdx, C.c_void_p(dW.data_ptr()), stream),
                    h, "lde_chain_backward_saved_delta", chain=True)
            return None, dx, dW, None, None, (g if ctx.has_base else None), None
        if y.numel() == 0:
            raise L.LdeError("decode_loss(want_x_hat=False): x̂ was not kept — it cannot carry a cotangent of its own, and another forward of this "
                             "decoder has re-staged the loss's δ since (keep x̂, or run each forward's backward before the next forward)")
        L.check(lib.lde_chain_backward_saved_mse(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(target.data_ptr()),
                                                 C.c_void_p(g.data_ptr()), ctx.scale, C.c_void_p(dy.data_ptr()) if dy is not None else C.c_void_p(),
                                                 psv, N, pdx, C.c_void_p(dW.data_ptr()), stream), h, "lde_chain_backward_saved_mse", chain=True)
        return None, dx, dW, None, None, (g if ctx.has_base else None), None


class _ChainGroupFn(torch.autograd.Function):
    """(y_1, …, y_n) = (chain_1(x_1), …, chain_n(x_n)) for INDEPENDENT chains as one autograd node and — when the library can merge
    them (n ≤ 4, same dtype mode, small tiles) — one launch per stage: lde_chain_group_forward_save / _backward_saved. The heads of
    apply_latent_in and the chains of apply_latent_out act on B columns only; as separate nodes they are 6 forward and 18 pullback
    launches of a few microseconds each. Same kernels on the same arguments per chain: the results are those of _ChainFn bit for bit
    (tests/test_gpu_chain.py::test_grouped_chains_equal_separate_calls). Inputs: chains, then x_1 … x_n (N_i, in_i), then W_1 … W_n."""

    @staticmethod
    def forward(ctx, chains, *xw):
        n = len(chains)
        xs, Ws = xw[:n], xw[n:]
        if not all(x.is_cuda for x in xs):
            raise L.LdeError("Chain needs CUDA/HIP tensors: it runs on the GPU only (no CPU fallback)")
        lib = chains[0]._lib if chains[0]._handle is not None else None
        stream = L.raw_stream(xs[0].device.index)
        hs = []
        for c, W in zip(chains, Ws):
            h = c._native()
            lib = c._lib
            key = L.weights_key(W)
            if c._wkey != key:
                Wc = W.detach().contiguous().float()
                L.check(lib.lde_chain_set_weights_device(h, C.c_void_p(Wc.data_ptr()), Wc.numel(), stream), h,
                        "lde_chain_set_weights_device", chain=True)
                c._wkey = None
            hs.append(h)
        ys = [torch.empty((x.shape[0], c.sizes[-1]), device=x.device, dtype=torch.float32) for c, x in zip(chains, xs)]
        train = any(ctx.needs_input_grad[1:])
        saveds = [torch.empty((int(lib.lde_chain_saved_floats(h, x.shape[0])),), device=x.device, dtype=torch.float32) for h, x in zip(hs, xs)] \
            if train else None
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
        ctx.c_handles = arr([h.value for h in hs])
        ctx.c_Ns = (C.c_int64 * n)(*[x.shape[0] for x in xs])
        rc = lib.lde_chain_group_forward_save(n, ctx.c_handles, arr([x.data_ptr() for x in xs]), ctx.c_Ns, arr([y.data_ptr() for y in ys]),
                                              arr([t.data_ptr() for t in saveds]) if train else None, stream)
        L.check(rc, hs[0], "lde_chain_group_forward_save", chain=True)
        ctx.chains, ctx.n, ctx.train = chains, n, train
        ctx.need_dx = [bool(g) for g in ctx.needs_input_grad[1:1 + n]]
        ctx.save_for_backward(*xs, *ys, *(saveds if train else []))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        chains, n = ctx.chains, ctx.n
        t = ctx.saved_tensors
        xs, ys, saveds = t[:n], t[n:2 * n], (t[2 * n:] if ctx.train else None)
        lib = chains[0]._lib
        stream = L.raw_stream(xs[0].device.index)
        dys = [dy.contiguous().float() for dy in dys]
        dxs = [torch.empty_like(x) if need else None for x, need in zip(xs, ctx.need_dx)]
        dWs = [torch.empty((c.num_weights,), device=x.device, dtype=torch.float32) for c, x in zip(chains, xs)]   # written, not accumulated
        if L.dw_stream is not None:
            for dW in dWs:
                dW.record_stream(L.dw_stream)
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
        rc = lib.lde_chain_group_backward_saved(n, ctx.c_handles, arr([x.data_ptr() for x in xs]), arr([y.data_ptr() for y in ys]),
                                                arr([d.data_ptr() for d in dys]), arr([s_.data_ptr() for s_ in saveds]) if saveds is not None else None,
                                                ctx.c_Ns, arr([d.data_ptr() if d is not None else None for d in dxs]),
                                                arr([d.data_ptr() for d in dWs]), stream)
        L.check(rc, chains[0]._native(), "lde_chain_group_backward_saved", chain=True)
        return (None, *dxs, *dWs)


_CHAIN_GROUP = True   # apply_latent_in / apply_latent_out: independent chains as one node (diagnostic switch)


def apply_chains_grouped(pairs):
    """[(chain, x [in, B]), …] → [chain(x), …] for independent chains, as ONE autograd node (one launch per stage where the library can
    merge them). Falls back to separate calls for anything that is not a plain Chain on a 2-D CUDA tensor."""
    ok = _CHAIN_GROUP and 2 <= len(pairs) <= 4 and all(isinstance(m, Chain) and x.dim() == 2 and x.is_cuda for m, x in pairs)
    if not ok:
        return [m(x) for m, x in pairs]
    chains = tuple(m for m, _ in pairs)
    xs = [x.t().contiguous().float() for _, x in pairs]
    ys = _ChainGroupFn.apply(chains, *xs, *[m.flat_weights() for m in chains])
    return [y.t() for y in ys]


class Chain(torch.nn.Module):
    """Chain(layers...) of Dense / SkipConnection(Dense) applied to the first dimension of x ([in, B] or [in, B, T])."""

    def __init__(self, *layers):
        super().__init__()
        self.layers = list(layers)
        dense = [l.layer if isinstance(l, SkipConnection) else l for l in layers]
        if not dense or not all(isinstance(d, Dense) for d in dense):
            raise TypeError("Chain takes Dense and SkipConnection(Dense) layers")
        for a, b in zip(dense[:-1], dense[1:]):
            if a.n_out != b.n_in:
                raise ValueError("layer sizes do not chain")
        self._dense = dense
        self.sizes = [dense[0].n_in] + [d.n_out for d in dense]
        self.acts = [_ACT[d.act] for d in dense]
        self.skips = [int(isinstance(l, SkipConnection)) for l in layers]
        self.num_weights = sum(d.n_in * d.n_out + d.n_out for d in dense)
        # ONE flat parameter in Flux.destructure order (per Dense vec(W) column-major [out×in], then b): it is what
        # lde_chain_* consumes and what the gradient comes back as — no per-step gathering or scattering of pieces
        parts, off = [], 0
        for d in dense:
            parts += [d.weight.detach().t().reshape(-1), d.bias.detach()]
            d._off = off
            off += d.n_in * d.n_out + d.n_out
        self.theta = torch.nn.Parameter(torch.cat(parts).float())
        for d in dense:
            d._owner, d._w0, d._b0 = self, None, None
        self._handle = None
        self._lib = None
        self._wkey = None          # set by _lib.refresh_weights: the parameter value the handle already holds
        self.dtype = "f32"

    def set_dtype(self, dtype: str) -> "Chain":
        """'f32' (default) or 'bf16': operands of every matrix product rounded to bfloat16, f32 accumulation, f32 master weights
        and activations (lde_chain_set_dtype; BASELINE.json configs[4] "mixed fp32 solve / bf16 encoder-decoder")."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype: 'f32' or 'bf16'")
        self.dtype = dtype
        if self._handle is not None:
            L.check(self._lib.lde_chain_set_dtype(self._handle, L.DTYPE_BF16 if dtype == "bf16" else L.DTYPE_F32), self._handle,
                    "lde_chain_set_dtype", chain=True)
        return self

    def _native(self):
        if self._handle is None:
            self._lib = L.load()
            d = L.ChainDesc()
            d.abi_version = L.LDE_ABI_VERSION
            d.n_layers = len(self._dense)
            for i, s in enumerate(self.sizes):
                d.sizes[i] = s
            for i, (a, k) in enumerate(zip(self.acts, self.skips)):
                d.activation[i], d.skip[i] = a, k
            h = C.c_void_p()
            rc = self._lib.lde_chain_create(C.byref(d), C.byref(h))
            if rc != 0:
                try:
                    L.check(rc, h if h else None, "lde_chain_create", chain=True)
                finally:
                    if h:
                        self._lib.lde_chain_destroy(h)
            L.check(self._lib.lde_chain_set_accumulate(h, 0), h, "lde_chain_set_accumulate", chain=True)   # the pullback hands autograd a fresh gradient
            if self.dtype == "bf16":
                L.check(self._lib.lde_chain_set_dtype(h, L.DTYPE_BF16), h, "lde_chain_set_dtype", chain=True)
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None and self._lib is not None:
                self._lib.lde_chain_destroy(self._handle)
        except Exception:
            pass

    def flat_weights(self) -> torch.Tensor:
        """Flux.destructure order: per Dense vec(W) column-major [out×in], then b — the parameter itself."""
        return self.theta

    def apply_batch_major(self, x_N_in: torch.Tensor) -> torch.Tensor:
        """x (N, in) contiguous → y (N, out). Differentiable wrt x and the layers' parameters."""
        return _ChainFn.apply(self, x_N_in.contiguous().float(), self.flat_weights())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [in, B] → [out, B];  x [in, B, T] → [out, B, T] (a Flux Dense acts on the first dimension)."""
        if x.dim() == 2:
            return self.apply_batch_major(x.t()).t()
        if x.dim() == 3:
            n_in, B, T = x.shape
            buf = x.permute(2, 1, 0)                          # (T, B, in): contiguous when x came from diffeq_layer
            y = self.apply_batch_major(buf.reshape(T * B, n_in))
            return y.reshape(T, B, -1).permute(2, 1, 0)
        raise ValueError("Chain takes [in, B] or [in, B, T] arrays")


def apply_latent_out(decoder: Decoder, l_tilde):
    """l̂ = apply_latent_out(decoder, l̃)  [REF src/models/GOKU.jl:83-91], [REF src/models/LatentODE.jl:53]."""
    if isinstance(decoder.model_type, GOKU):
        z0_t, th_t = l_tilde
        lo_z0, lo_th = decoder.latent_out
        z0_hat, th_hat = apply_chains_grouped([(lo_z0, z0_t), (lo_th, th_t)])   # the two chains as one autograd node, one launch per stage
        return z0_hat, th_hat
    if isinstance(decoder.model_type, LatentODE):
        return decoder.latent_out(l_tilde)
    raise TypeError(f"no apply_latent_out method for model type {type(decoder.model_type).__name__}")


def apply_reconstructor(decoder: Decoder, z_hat: torch.Tensor) -> torch.Tensor:
    """x̂ = decoder.reconstructor(ẑ)  [REF src/models/GOKU.jl:148], [REF src/models/LatentODE.jl:80]."""
    return decoder.reconstructor(z_hat)


def decode(decoder: Decoder, l_tilde, t):
    """(x̂, ẑ, l̂) = decoder(l̃, t)  [REF src/models/LatentDiffEqModel.jl:101-113]."""
    l_hat = apply_latent_out(decoder, l_tilde)
    z_hat = diffeq_layer(decoder, l_hat, t)
    x_hat = apply_reconstructor(decoder, z_hat)
    return x_hat, z_hat, l_hat


_RECON_MSE_FWD = True   # _ChainMseFn: the loss value from the reconstructor's forward launch (its squares summed
                                                                     # per column tile in the last layer's epilogue) instead of lde_mse_forward's pass over x and x̂
_RECON_MSE = True   # decode_loss: the reconstructor and reconstruction_loss as one autograd node (diagnostic switch)


def decode_loss(decoder: Decoder, l_tilde, t, x, batch_size=None, plus=None, want_x_hat: bool = True):
    """(reconstruction_loss(x, x̂) [+ plus], (x̂, ẑ, l̂)) with (x̂, ẑ, l̂) = decoder(l̃, t)  [REF src/models/LatentDiffEqModel.jl:101-113],
    [REF examples/pendulum_friction-less/model_train.jl:225-238] — `decode` followed by `loss.reconstruction_loss`, with the
    reconstructor and the loss as ONE autograd node when the reconstructor is a Chain on HIP frames x [pixels, B, T] (_ChainMseFn).
    `want_x_hat=False` (a training step that only needs the loss): a bf16 reconstructor then never writes x̂ [pixels × B × T] — the largest
    array of the step — to HBM, and x̂ in the returned tuple is None."""
    from .loss import reconstruction_loss
    l_hat = apply_latent_out(decoder, l_tilde)
    z_hat = diffeq_layer(decoder, l_hat, t)
    rec = decoder.reconstructor
    if not (_RECON_MSE and isinstance(rec, Chain) and z_hat.dim() == 3 and z_hat.is_cuda and x.dim() == 3 and x.is_cuda):
        x_hat = apply_reconstructor(decoder, z_hat)
        return reconstruction_loss(x, x_hat, batch_size, plus=plus), (x_hat, z_hat, l_hat)
    n_in, B, T = z_hat.shape
    if x.shape[1:] != (B, T) or x.shape[0] != rec.sizes[-1]:
        raise ValueError("decode_loss: x and the decoder's output differ in shape")
    zb = z_hat.permute(2, 1, 0).reshape(T * B, n_in).contiguous().float()          # (T·B, D): in place when ẑ came from diffeq_layer
    xt = x.permute(2, 1, 0).reshape(T * B, x.shape[0]).contiguous().float()       # the frames in the same (T·B, pixels) order
    n_mean = (batch_size or B) * T
    loss, y = _ChainMseFn.apply(rec, zb, rec.flat_weights(), xt, 1.0 / n_mean, plus.float() if plus is not None else None, want_x_hat)
    return loss, ((y.reshape(T, B, -1).permute(2, 1, 0) if y.numel() else None), z_hat, l_hat)


def default_decoder_layers(model_type, input_dim: int, diffeq, hidden_dim_resnet: int = 200, latent_dim_z0: int = 16,
                           latent_dim_theta: int = 16, latent_to_diffeq_dim: int = 200, z0_activation: str = "identity",
                           theta_activation: str = "softplus", output_activation: str = "sigmoid", device=None):
    """The decoder half of default_layers  [REF src/models/GOKU.jl:247-271], [REF src/models/LatentODE.jl:130-142]:
    (latent_out, diffeq, reconstructor)."""
    if isinstance(model_type, GOKU):
        z_dim, th_dim = len(diffeq.prob.u0), len(diffeq.prob.p)
        lo_z0 = Chain(Dense(latent_dim_z0, latent_to_diffeq_dim, "relu"), Dense(latent_to_diffeq_dim, z_dim, z0_activation))
        lo_th = Chain(Dense(latent_dim_theta, latent_to_diffeq_dim, "relu"),
                      Dense(latent_to_diffeq_dim, th_dim, theta_activation))
        latent_out = (lo_z0, lo_th)
        rec_in = z_dim
    elif isinstance(model_type, LatentODE):
        latent_out = lambda x: x                                            # [REF LatentODE.jl:142]
        rec_in = diffeq.latent_dim_out
    else:
        raise TypeError("default_decoder_layers: GOKU_basic() or LatentODE()")
    rec = Chain(Dense(rec_in, hidden_dim_resnet, "relu"),
                SkipConnection(Dense(hidden_dim_resnet, hidden_dim_resnet, "relu")),
                SkipConnection(Dense(hidden_dim_resnet, hidden_dim_resnet, "relu")),
                Dense(hidden_dim_resnet, input_dim, output_activation))
    if device is not None:
        rec = rec.to(device)
        if isinstance(latent_out, tuple):
            latent_out = tuple(m.to(device) for m in latent_out)
    return latent_out, diffeq, rec
