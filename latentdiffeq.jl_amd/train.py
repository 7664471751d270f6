"""Host-side training harness (scope row f-3): what the reference's example script wraps around the model. Pure host code —
the compute under `model(x, t)` and the loss terms is liblde.so (api.py, chain.py, recurrent.py, loss.py); this file launches
no kernel of its own.

    reference                                                                         here
    --------------------------------------------------------------------------------  -----------------------------
    LatentDiffEqModel(model_type, encoder_layers, decoder_layers); model(x, t, variational)
                                                      [REF src/models/LatentDiffEqModel.jl:6-37]      LatentDiffEqModel
    default_layers(model_type, input_dim, diffeq)     [REF src/models/GOKU.jl:201-273]               default_layers
    loss_batch(model, x, t, β, variational)           [REF examples/pendulum_friction-less/model_train.jl:225-238]   loss_batch
    kl, vector_kl                                     [REF src/utils/utils.jl:15-49]                  kl (elementwise), loss.vector_kl
    frange_cycle_linear(n_iter, start, stop, n_cycle, ratio)   [REF src/utils/utils.jl:53-66]         frange_cycle_linear
    normalize_to_unit_segment / denormalize_unit_segment       [REF src/utils/utils.jl:71-79]         same
    time_loader(x, full_seq_len, seq_len), rand_time  [REF src/utils/utils.jl:85-99]                  time_loader, rand_time
    the epoch loop: β schedule, progressive sequence length, ADAMW, per-minibatch validation loss, best weights
                                                      [REF model_train.jl:138-150, :172-218]          train
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .api import Decoder
from .chain import decode, decode_loss, default_decoder_layers
from .loss import reconstruction_loss, sample, sample_with_kl, vector_kl  # noqa: F401
from .loss import backward as _loss_backward
from .recurrent import Encoder, default_encoder_layers, encode


_FUSED_LOSS = True    # module-level switches (diagnostic: set the attribute before the call; nothing here reads the environment)
_NATIVE_ADAM = True   # FluxADAMW: the one-launch lde_adamw_flux_step   # loss_batch: sample + KL in one pass (diagnostic switch)


class LatentDiffEqModel:
    """LatentDiffEqModel(model_type, encoder_layers, decoder_layers); `model(x, t, variational)` → ((x̂, ẑ, l̂), μ, logσ²)."""

    def __init__(self, model_type, encoder_layers, decoder_layers):
        self.model_type = model_type
        self.encoder = Encoder(model_type, encoder_layers)
        self.decoder = Decoder(model_type, decoder_layers)

    def modules(self) -> List[torch.nn.Module]:
        e, d = self.encoder, self.decoder
        pe = list(e.pattern_extractor) if isinstance(e.pattern_extractor, tuple) else [e.pattern_extractor]
        lo = list(d.latent_out) if isinstance(d.latent_out, tuple) else []
        extra = [d.diffeq.dudt] if hasattr(d.diffeq, "dudt") else []   # the NODE's weights: trained here
        return [e.feature_extractor, *pe, *e.latent_in, *lo, *extra, d.reconstructor]

    def parameters(self) -> List[torch.nn.Parameter]:
        return [p for m in self.modules() for p in m.parameters()]

    def refresh_weights(self) -> int:
        """After an optimiser step: hand the new weights of every chain and recurrent stack to the library in ONE launch
        (lde_refresh_weights) instead of one per module at its next call. Optional — a module whose parameter changed without
        it re-uploads by itself."""
        from . import _lib as L
        return L.refresh_weights(self.modules())

    def __call__(self, x, t, variational: bool = False):
        mu, logvar = encode(self.encoder, x)
        l_tilde = sample(mu, logvar) if variational else mu                      # [REF LatentDiffEqModel.jl:31]
        return decode(self.decoder, l_tilde, t), mu, logvar


def default_layers(model_type, input_dim: int, diffeq, device=None, **kw):
    """(encoder_layers, decoder_layers)  [REF src/models/GOKU.jl:201-273], [REF src/models/LatentODE.jl:91-145]."""
    ekw = {k: v for k, v in kw.items() if k in ("hidden_dim_resnet", "rnn_input_dim", "rnn_output_dim", "latent_dim_z0", "latent_dim_theta")}
    dkw = {k: v for k, v in kw.items() if k in ("hidden_dim_resnet", "latent_dim_z0", "latent_dim_theta", "latent_to_diffeq_dim",
                                                "z0_activation", "theta_activation", "output_activation")}
    return (default_encoder_layers(model_type, input_dim, diffeq, device=device, **ekw),
            default_decoder_layers(model_type, input_dim, diffeq, device=device, **dkw))


def kl(mu, logvar):
    """kl(μ, logσ²) = (exp(logσ²) + μ² − logσ² − 1) / 2, elementwise  [REF src/utils/utils.jl:15]."""
    return (torch.exp(logvar) + mu ** 2 - logvar - 1) / 2


def loss_batch(model, x, t, beta: float, variational: bool, batch_size: Optional[int] = None):
    """reconstruction_loss + β·kl_loss with reconstruction_loss = sum(mean((x − x̂)², dims=(2,3)))  [REF model_train.jl:225-238].
    `batch_size`: the GLOBAL minibatch size when x is one rank's shard (the per-rank losses then add up to the reference's)."""
    if variational and _FUSED_LOSS:
        # the same expression with the sample and the KL term read in one pass and the scalar additions folded into the
        # reductions (loss.sample_with_kl): encoder → (l̃, β·kl) → decoder → reconstruction_loss + β·kl
        mu, logvar = encode(model.encoder, x)
        l_tilde, bkl = sample_with_kl(mu, logvar, beta, batch_size)
        loss, _ = decode_loss(model.decoder, l_tilde, t, x, batch_size, plus=bkl, want_x_hat=False)   # (the reconstructor and the loss as one autograd node; x̂ is not kept)
        return loss
    (x_hat, _z, _l), mu, logvar = model(x, t, variational)
    return reconstruction_loss(x, x_hat, batch_size) + beta * vector_kl(mu, logvar, batch_size)


def frange_cycle_linear(n_iter: int, start: float = 0.0, stop: float = 1.0, n_cycle: int = 4, ratio: float = 0.5) -> np.ndarray:
    """Cyclical KL-annealing schedule, the reference's loop transcribed index for index (1-based `L[Int(round(i + c·period))]`
    with the strict `< n_iter` guard — so the last entry is never rewritten; Julia's `round` is round-half-even like Python's)
    [REF src/utils/utils.jl:53-66]."""
    L = np.ones(n_iter, dtype=np.float64) * stop
    period = n_iter / n_cycle
    step = np.float32((stop - start) / (period * ratio))
    for c in range(n_cycle):
        v, i = np.float32(start), 1
        while v <= stop and int(round(i + c * period)) < n_iter:
            L[int(round(i + c * period)) - 1] = v
            v = np.float32(v + step)
            i += 1
    return L.astype(np.float32)


def normalize_to_unit_segment(X):
    """[REF src/utils/utils.jl:71-77]."""
    lo, hi = X.min(), X.max()
    return (X - lo) / (hi - lo), lo, hi


def denormalize_unit_segment(Xh, lo, hi):
    """[REF src/utils/utils.jl:79]."""
    return Xh * (hi - lo) + lo


def rand_time(full_seq_len: int, seq_len: int, rng: Optional[np.random.Generator] = None):
    """start_time = rand(1:full_seq_len − seq_len); idxs = start:start+seq_len−1 (1-based, inclusive)  [REF utils.jl:95-99].
    Returned 0-based as a slice. (As in the reference the last possible window is never drawn.)"""
    rng = rng or np.random.default_rng()
    start = int(rng.integers(1, full_seq_len - seq_len + 1))      # 1 … full − seq, inclusive
    return slice(start - 1, start - 1 + seq_len)


def time_loader(x, full_seq_len: int, seq_len: int, rng: Optional[np.random.Generator] = None):
    """One random window of seq_len frames, the same for every sample of the minibatch: x [pixels, B, full] → [pixels, B, seq]
    [REF src/utils/utils.jl:85-93]."""
    return x[:, :, rand_time(full_seq_len, seq_len, rng)]


class FluxADAMW(torch.optim.Adam):
    """`ADAMW(η, (β₁, β₂), decay)` of the pinned Flux 0.13.6 [REF Manifest.toml:452] = `Optimiser(ADAM(η, β), WeightDecay(decay))`
    [REF examples/pendulum_friction-less/model_train.jl:138]: Δ = η·m̂/(√v̂ + ε), then Δ += decay·x, then x −= Δ, i.e.
    x ← (1 − decay)·x − ADAM step — the decay is NOT multiplied by η (torch.optim.AdamW applies lr·weight_decay·x: 1000× weaker at
    the example's η = decay = 1e-3). ε = 1e-8 and the bias correction are Flux's = torch's.
    On the GPU (`native`, the default when every parameter is a contiguous f32 HIP tensor) the whole update is ONE liblde.so launch
    for all parameter arrays (lde_adamw_flux_step) — no per-step tensor grouping, step-counter kernels or separate decay pass;
    otherwise one multi-tensor scale of the parameters followed by torch's Adam update with the gradients taken at the un-decayed
    point (same arithmetic up to rounding; tests/test_gpu_optim.py compares the two). The native path bumps the parameters'
    version counters after its launch (it writes through raw pointers), so `refresh_weights()` is optional after it too. Its
    optimiser state keeps `step` as a Python int (a tensor step from a torch checkpoint is converted on entry); a native
    state_dict is therefore not resumable by torch's FUSED Adam, which expects tensor steps."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), decay: float = 0.0, eps: float = 1e-8, fused=None, native=None,
                 capturable: bool = False):
        groups = list(params)
        params = [p for g in groups for p in (g["params"] if isinstance(g, dict) else [g])]     # (arrays, or torch's group dicts)
        on_gpu = bool(params) and all(p.is_cuda for p in params)
        if fused is None:
            fused = on_gpu
        self.capturable = False             # (add_param_group runs inside the base constructor)
        super().__init__(groups, lr=lr, betas=betas, eps=eps, weight_decay=0.0, fused=fused or None)
        self.decay = float(decay)
        ok = on_gpu and all(p.dtype == torch.float32 and p.is_contiguous() for p in params)
        if native is None:
            native = ok and _NATIVE_ADAM
        if native and not ok:
            raise ValueError("FluxADAMW(native=True) needs contiguous float32 HIP parameters")
        self.native = bool(native)
        # capturable (native only): the step count lives in ONE device int64 shared by all arrays (lde_adamw_flux_step_dev), so that the
        # update can sit inside a captured hipGraph (train.GraphedStep); every array must then have a gradient at every step
        self.capturable = bool(capturable)
        if self.capturable and not self.native:
            raise ValueError("FluxADAMW(capturable=True) needs the native path")
        self._step_dev = torch.zeros((), dtype=torch.int64, device=params[0].device) if self.capturable else None
        if self.capturable:
            if len(self.param_groups) != 1:     # one device counter, bumped by every lde_adamw_flux_step_dev launch: one launch per step
                raise ValueError("FluxADAMW(capturable=True) takes ONE parameter group")
            from .loss import set_noise_epoch
            set_noise_epoch(self._step_dev)     # a captured step's ε is keyed by this counter: fresh noise at every replay (loss.randn)

    def add_param_group(self, param_group):
        if getattr(self, "capturable", False) and self.param_groups:
            raise ValueError("FluxADAMW(capturable=True) takes ONE parameter group")
        super().add_param_group(param_group)

    def state_dict(self):
        """The capturable form's step count lives on the device (`_step_dev`), not in `state`: a checkpoint carries it as every array's
        `step` (what the non-capturable and torch forms keep), so that a resumed run's bias corrections continue at t, not at 0."""
        sd = super().state_dict()
        if self.capturable:
            t = int(self._step_dev.item())
            sd["state"] = {k: dict(v, step=t) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self.capturable:
            steps = [int(st["step"]) for st in self.state.values() if "step" in st]
            if steps and min(steps) != max(steps):
                raise ValueError("FluxADAMW(capturable=True): the checkpoint's arrays have been updated unequally often")
            self._step_dev.fill_(steps[0] if steps else 0)
            for st in self.state.values():
                st.pop("step", None)

    @torch.no_grad()
    def _native_step(self):
        lib = L.load()
        if self.capturable:
            for g in self.param_groups:
                ps = g["params"]
                if any(p.grad is None for p in ps):
                    raise RuntimeError("FluxADAMW(capturable=True): every parameter needs a gradient at every step")
                keep = []
                tab = (L.AdamTensor * len(ps))()
                for i, p in enumerate(ps):
                    st, gr = self.state[p], p.grad
                    if not st:
                        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                    if gr.dtype != torch.float32 or not gr.is_contiguous():
                        gr = gr.float().contiguous()
                        keep.append(gr)
                    t = tab[i]
                    t.p, t.g, t.m, t.v, t.n = p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                L.check(lib.lde_adamw_flux_step_dev(len(ps), tab, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.decay,
                                                    C.c_void_p(self._step_dev.data_ptr()), L.raw_stream(ps[0].device.index)), None,
                        "lde_adamw_flux_step_dev")
                torch.autograd.graph.increment_version(ps)
            return
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.grad is not None]
            if not ps:
                continue
            keep, by_step = [], {}
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                # a state loaded from a torch / fused Adam checkpoint carries `step` as a tensor: normalise to a Python int (the
                # native state is not resumable by torch's fused Adam, which wants tensor steps — documented in the class docstring)
                st["step"] = int(st["step"]) + 1
                by_step.setdefault(st["step"], []).append(p)
            # Flux keeps β₁ᵗ, β₂ᵗ per array: arrays that have been updated equally often share a launch (normally all of them)
            for t_step, group in by_step.items():
                tab = (L.AdamTensor * len(group))()
                for i, p in enumerate(group):
                    st, gr = self.state[p], p.grad
                    if gr.dtype != torch.float32 or not gr.is_contiguous():
                        gr = gr.float().contiguous()
                        keep.append(gr)
                    t = tab[i]
                    t.p, t.g, t.m, t.v, t.n = p.data_ptr(), gr.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                L.check(lib.lde_adamw_flux_step(len(group), tab, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.decay, t_step,
                                                L.raw_stream(group[0].device.index)), None, "lde_adamw_flux_step")
                # the kernel wrote through raw pointers: tell torch the parameters changed in place, so that `_lib.weights_key`
                # (data_ptr, _version) — which lets a module skip its weight upload — and autograd's saved-tensor checks see it
                torch.autograd.graph.increment_version(group)

    @torch.no_grad()
    def step(self, closure=None):
        if self.native:
            loss = None
            if closure is not None:
                with torch.enable_grad():
                    loss = closure()
            self._native_step()
            return loss
        loss = None
        if closure is not None:             # gradients at the UN-decayed point, as Flux's Optimiser(ADAM, WeightDecay) and the native path
            with torch.enable_grad():
                loss = closure()
        if self.decay:
            for g in self.param_groups:
                ps = [p for p in g["params"] if p.grad is not None]
                if ps:
                    torch._foreach_mul_(ps, 1.0 - self.decay)
        super().step()
        return loss


class GraphedStep:
    """A whole training step — `fn()`: forward, loss, backward, `FluxADAMW(capturable=True).step()`, `refresh_weights` — captured ONCE
    in a hipGraph (torch.cuda.CUDAGraph) and replayed: the step is ≈ 80 small launches whose host enqueue time (≈ 1.5 ms) equals the
    device time; a replay is one submission  [REF examples/pendulum_friction-less/model_train.jl:186-204: the loop body this replaces].
    What the capture needs, and gets: every workspace sized before it (`warmup` eager steps on a side stream), the optimiser's step count
    in device memory (lde_adamw_flux_step_dev), gradients at fixed addresses (allocated from the graph's pool during capture), inputs
    copied INTO the captured tensors (`static_inputs`: tensors `fn` reads; `replay(*new)` copies into them), and a single stream — call
    with the encoder's branch streams off (`recurrent._BRANCH_STREAMS = False`: cross-stream capture of the recurrent stacks' side streams aborts
    inside the HIP runtime on ROCm 7.2). ε of `sample` comes from torch's generator, which is graph-safe (its offsets advance per replay).
    Several GPUs: the collective stays OUTSIDE the graphs — pass the step in two halves: `fn` = zero_grad, forward, loss, backward;
    `between` = the gradient all-reduce (dist.FlatGradAllReduce: eager, on the gradients' fixed addresses); `fn2` = the optimiser step and
    the weight hand-over, a second graph. A replay is then graph · all-reduce · graph (capture in `thread_local` error mode: torch's
    collective watchdog thread may query its events meanwhile)."""

    def __init__(self, fn: Callable[[], torch.Tensor], static_inputs: Sequence[torch.Tensor] = (), warmup: int = 3,
                 between: Optional[Callable[[], None]] = None, fn2: Optional[Callable[[], None]] = None):
        self.fn, self.static_inputs, self.between, self.fn2 = fn, list(static_inputs), between, fn2

        def whole():
            out = fn()
            if between is not None:
                between()
            if fn2 is not None:
                fn2()
            return out
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                whole()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        mode = dict(capture_error_mode="thread_local") if (between is not None or fn2 is not None) else {}
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **mode):
            out = fn()
            self.loss = out.detach().clone() if isinstance(out, torch.Tensor) else None
        torch.cuda.synchronize()
        self.graph2 = None
        if fn2 is not None:
            if between is not None:
                between()                      # the gradients the second half is captured on are the reduced ones (shapes only matter)
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, **mode):
                fn2()
            torch.cuda.synchronize()

    def replay(self, *new_inputs) -> Optional[torch.Tensor]:
        for dst, src in zip(self.static_inputs, new_inputs):
            dst.copy_(src)
        self.graph.replay()
        if self.between is not None:
            self.between()
        if self.graph2 is not None:
            self.graph2.replay()
        return self.loss


def train(model: LatentDiffEqModel, loader_train: Iterable, val_set, dt: float, epochs: int, seq_len: int, full_seq_len: int,
          lr: float = 1e-3, decay: float = 1e-10, start_beta: float = 0.0, end_beta: float = 1.0, n_cycle: int = 3, ratio: float = 0.9,
          progressive_training: bool = False, prog_training_duration: int = 0, start_seq_len: int = 0, variational: bool = True,
          rng: Optional[np.random.Generator] = None, on_epoch: Optional[Callable] = None, grad_sync: Optional[Callable] = None):
    """The epoch loop of the example script  [REF examples/pendulum_friction-less/model_train.jl:138-218]: cyclical β, optional
    progressive sequence length, Flux-flavour ADAMW, validation loss after every minibatch, best weights kept. `loader_train` yields
    x [pixels, B, full_seq_len] tensors on the model's device; `grad_sync` is called between backward and the optimiser step
    (dist.FlatGradAllReduce for multi-GPU). Returns (history, best_state)."""
    params = model.parameters()
    opt = FluxADAMW(params, lr=lr, betas=(0.9, 0.999), decay=decay)      # ADAMW(η, (0.9, 0.999), decay), Flux flavour
    schedule = frange_cycle_linear(epochs, start_beta, end_beta, n_cycle, ratio)
    if progressive_training:
        prog = np.rint(np.linspace(start_seq_len, seq_len, prog_training_duration)).astype(int)
    else:
        prog_training_duration = 0
    t_val = np.arange(val_set.shape[2]) * dt
    best, best_state, history = float("inf"), None, []
    val_loss = 0.0
    for epoch in range(1, epochs + 1):
        beta = float(schedule[epoch - 1])
        sl = int(prog[epoch - 1]) if epoch <= prog_training_duration else seq_len
        t = np.arange(sl) * dt
        for x in loader_train:
            xb = time_loader(x, full_seq_len, sl, rng)
            opt.zero_grad(set_to_none=True)
            loss = loss_batch(model, xb, t, beta, variational)
            _loss_backward(loss)
            L.join_weight_gradients()          # no-op unless _lib.set_async_weight_gradients(True)
            if grad_sync is not None:
                grad_sync()
            opt.step()
            model.refresh_weights()
            with torch.no_grad():
                val_loss = float(loss_batch(model, val_set, t_val, beta, False))
            history.append((epoch, float(loss.detach()), val_loss))
        if on_epoch is not None:
            on_epoch(epoch, history)
        if val_loss < best:
            best = val_loss
            best_state = [p.detach().clone() for p in params]
    return history, best_state
