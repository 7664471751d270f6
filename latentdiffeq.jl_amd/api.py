"""Host-side mirror of the reference's plug-in interface for ONE path: `diffeq_layer`.

Same names, argument meaning and error behaviour as the reference (Julia) so that its examples read
the same here; everything under the call is liblde.so (HIP, gfx950) reached through the C ABI of
include/lde.h. PyTorch is used only for device memory, streams and autograd plumbing.

    reference                                                       here
    ---------------------------------------------------------------  -----------------------------------
    Pendulum(; solver, sensalg, kwargs...)    [REF pendulum.jl:4-46]   Pendulum(solver=, sensalg=, **kwargs)
    Pendulum_friction(...)                    [REF pendulum.jl:50-91]  Pendulum_friction(...)
    NODE(latent_dim_in; hidden_dim, augment_dim, kwargs...) [REF nODE.jl:3-32]   NODE(latent_dim_in, hidden_dim=, augment_dim=, **kwargs)
    GOKU_basic(), LatentODE()                 [REF GOKU.jl:6-7], [REF LatentODE.jl:6]   same
    Decoder(model_type, (latent_out, diffeq, reconstructor)) [REF LatentDiffEqModel.jl:79-99]   same
    diffeq_layer(decoder, l̂, t)               [REF GOKU.jl:98-130], [REF LatentODE.jl:61-78]   same
    transform_after_diffeq(x, diffeq)         [REF GOKU.jl:136]        same (override per diffeq class)

Array shapes follow the reference: ẑ₀ [D, B], θ̂ [P, B], result ẑ [D', B, T]. The memory layout is the
reference's column-major one, i.e. the torch tensors are (transposed) views of batch-major buffers:
`ẑ₀ = buf_BD.T`, `ẑ = buf_TBD.permute(2, 1, 0)`. Inputs that are not already such views are copied once.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

# ------------------------------------------------------------------------------------------------
# solver / sensealg tags (stand-ins for the OrdinaryDiffEq / SciMLSensitivity singletons)


class Tsit5:
    """Tsit5()  [REF pendulum.jl:11], [REF nODE.jl:15]."""
    code = L.SOLVER_TSIT5


class RK4:
    """RK4() — fixed step only: pass adaptive=False, dt=h through the diffeq kwargs."""
    code = L.SOLVER_RK4


class BacksolveAdjoint:
    """Reverse-time continuous adjoint (hinted at in [REF nODE.jl:17]).

    checkpointing=True (default) resets z to the saved ẑ(t_j) at every save time."""

    def __init__(self, checkpointing: bool = True):
        self.code = L.SENSE_BACKSOLVE_CHECKPOINTED if checkpointing else L.SENSE_BACKSOLVE


class InterpolatingAdjoint(BacksolveAdjoint):
    """DiffEqFlux's NeuralODE default [REF src/models/LatentODE.jl:67-70]. Served by the checkpointed
    reverse-time adjoint: same continuous adjoint, z taken from the saved states instead of a dense interpolant."""

    def __init__(self):
        super().__init__(checkpointing=True)


class ParallelAdjoint(BacksolveAdjoint):
    """Checkpointed continuous adjoint, parallel in time (LDE_SENSE_PARALLEL_CHECKPOINTED): one lane per
    (trajectory, save interval) integrates the interval's transition operator, a short scan composes them.
    Analytic right-hand sides with per-trajectory batching (the GOKU path)."""

    def __init__(self):
        self.code = L.SENSE_PARALLEL_CHECKPOINTED


class DiscreteSensitivity(BacksolveAdjoint):
    """LDE_SENSE_DISCRETE: the exact derivative of the DISCRETE solve on its accepted step sequence (step sizes constant) —
    what `ForwardDiffSensitivity()` computes in the reference [REF pendulum.jl:11], [REF src/models/GOKU.jl:107, :121], here in
    reverse mode: the forward solve records (t_n, dt_n, y_n) per accepted step, the pullback sweeps the steps in reverse.
    Every right-hand side and both batchings; 5–13× fewer evaluations than the continuous adjoint on the MLP right-hand sides."""

    def __init__(self):
        self.code = L.SENSE_DISCRETE


class ForwardDiffSensitivity(DiscreteSensitivity):
    """The reference's GOKU default [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into `solve` at
    [REF src/models/GOKU.jl:107, :121]: the exact derivative of the discrete solve on its accepted steps = LDE_SENSE_DISCRETE
    (see DiscreteSensitivity; upstream pushes dual numbers through the stepper, here the same derivative is taken in reverse mode).
    `Pendulum()` carries it, as in the reference. The continuous adjoints stay selectable: `ParallelAdjoint()` (time-parallel,
    checkpointed), `BacksolveAdjoint()`, `InterpolatingAdjoint()`. (`exact=False`, rounds 1–5's meaning of this tag — the time-parallel
    continuous adjoint — is still accepted and equals `ParallelAdjoint()`.)"""

    def __init__(self, exact: bool = True):
        self.code = L.SENSE_DISCRETE if exact else L.SENSE_PARALLEL_CHECKPOINTED


# ------------------------------------------------------------------------------------------------
# model-type tags


class LatentDE:
    """abstract type LatentDE  [REF src/LatentDiffEq.jl:11]."""


class GOKU(LatentDE):
    """abstract type GOKU <: LatentDE  [REF src/models/GOKU.jl:6]."""


class GOKU_basic(GOKU):
    """struct GOKU_basic <: GOKU  [REF src/models/GOKU.jl:7]."""


class LatentODE(LatentDE):
    """struct LatentODE <: LatentDE  [REF src/models/LatentODE.jl:6]."""


# ------------------------------------------------------------------------------------------------
# native handle


class _Handle:
    """RAII wrapper of lde_handle*."""

    def __init__(self, desc: L.ProblemDesc):
        self.lib = L.load()
        self.desc = desc
        self.ptr = C.c_void_p()
        L.check(self.lib.lde_create(C.byref(desc), C.byref(self.ptr)), None, "lde_create")
        self.nW = int(self.lib.lde_num_weights(C.byref(desc)))
        self.check_record = True          # _SolveFn.backward: look at the step record's counts before the discrete pullback

    def __del__(self):
        try:
            if getattr(self, "ptr", None) and self.ptr.value:
                self.lib.lde_destroy(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass

    def stats(self, which: int = 0) -> dict:
        s = L.Stats()
        stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.lde_get_stats(self.ptr, which, C.byref(s), C.c_void_p(stream)), self.ptr, "lde_get_stats")
        return dict(nfe=s.nfe, naccept=s.naccept, nreject=s.nreject, nfailed=s.nfailed, max_steps=s.max_steps)


def _make_desc(rhs_kind: int, state_dim: int, param_dim: int, augment_dim: int, layers: Sequence[int], solver,
               sensealg, batching: int, kwargs: dict) -> L.ProblemDesc:
    lib = L.load()
    d = L.ProblemDesc()
    lib.lde_problem_desc_default(C.byref(d))
    d.rhs_kind, d.state_dim, d.param_dim, d.augment_dim = rhs_kind, state_dim, param_dim, augment_dim
    d.n_layers = max(len(layers) - 1, 0)
    if d.n_layers > L.LDE_MAX_LAYERS:
        raise ValueError(f"RHS MLP has {d.n_layers} Dense layers; at most {L.LDE_MAX_LAYERS} supported")
    for i, s in enumerate(layers):
        d.layer_sizes[i] = int(s)
    d.solver = solver.code
    d.sensealg = sensealg.code if sensealg is not None else L.SENSE_BACKSOLVE_CHECKPOINTED
    d.batching = batching
    kw = dict(kwargs)
    d.activation = {"relu": L.ACT_RELU, "tanh": L.ACT_TANH}[kw.pop("activation", "relu")]
    # the kwargs... splat into solve()  [REF GOKU.jl:121], [REF LatentODE.jl:70]
    for name in ("abstol", "reltol", "dt", "dtmin", "qmin", "qmax", "gamma", "beta1", "beta2"):
        if name in kw:
            setattr(d, name, float(kw.pop(name)))
    if "maxiters" in kw:
        d.maxiters = int(kw.pop("maxiters"))
    if "adaptive" in kw:
        d.adaptive = int(bool(kw.pop("adaptive")))
    kw.pop("saveat", None)  # always `t`, as in the reference
    if kw:
        raise TypeError(f"unsupported solve kwargs: {sorted(kw)}")
    return d


# ------------------------------------------------------------------------------------------------
# the `diffeq` plug-in structs


class _ODEProblemStub:
    """What default_layers reads from diffeq.prob: u0 and p  [REF src/models/GOKU.jl:207-208]."""

    def __init__(self, u0, p, tspan):
        self.u0, self.p, self.tspan = u0, p, tspan


class _PhysicsDiffEq:
    _rhs_kind = L.RHS_PENDULUM

    def __init__(self, solver=None, sensalg=None, sensealg=None, **kwargs):
        # the reference's constructor keyword is spelled `sensalg`, the field `sensealg` [REF pendulum.jl:8, :11]
        self.prob = _ODEProblemStub(np.array([1.0, 1.0], np.float32), np.array([1.0], np.float32), (0.0, 1.0))
        self.solver = solver if solver is not None else Tsit5()
        sa = sensealg if sensealg is not None else sensalg
        self.sensealg = sa if sa is not None else ForwardDiffSensitivity()
        self.kwargs = kwargs
        self._handle: Optional[_Handle] = None

    def _native(self) -> _Handle:
        if self._handle is None:
            d = _make_desc(self._rhs_kind, 2, 1, 0, (), self.solver, self.sensealg, L.BATCH_PER_TRAJECTORY, self.kwargs)
            self._handle = _Handle(d)
        return self._handle

    def flat_weights(self):
        return None


class Pendulum(_PhysicsDiffEq):
    """Frictionless pendulum, du = [y, -G/L sin x], G = 10, L = p[1]  [REF pendulum.jl:4-46]."""
    _rhs_kind = L.RHS_PENDULUM


class Pendulum_friction(_PhysicsDiffEq):
    """Pendulum with friction b/m = 0.7  [REF pendulum.jl:50-91]."""
    _rhs_kind = L.RHS_PENDULUM_FRICTION


class NeuralODE:
    """Tag for NODE.neural_model  [REF nODE.jl:16]."""


class NODE:
    """Neural-ODE `diffeq` struct  [REF examples/pendulum_friction-less/nODE.jl:3-32].

    dudt = Chain(Dense(D+a, H, relu), Dense(H, H, relu), Dense(H, D+a)), H = 200 by default.
    Unlike the reference (where `dudt` is invisible to Flux.params — SURVEY.md B2), `dudt` is a torch
    module whose parameters receive gradients from the adjoint.

    `batching="coupled"` (default) is the NeuralODE semantics: one solve on the [D'×B] matrix state with a
    shared step size and an RMS error norm over all D'·B entries [REF src/models/LatentODE.jl:70-72].
    `batching="coupled_global"`: the same ONE solve with its batch sharded over ranks — this rank integrates its columns and the
    sums of the step control run over all ranks' columns through `set_global_sum(fn, global_batch)` (SURVEY.md §8e option (ii):
    the parity mode; `dist.global_sum_hook()` builds `fn` from a torch.distributed HOST group). Calls are synchronous in this mode.
    """

    def __init__(self, latent_dim_in: int, hidden_dim: int = 200, augment_dim: int = 0, device=None,
                 hidden_layers: int = 2, solver=None, sensealg=None, batching: str = "coupled", **kwargs):
        Dp = latent_dim_in + augment_dim
        sizes = [Dp] + [hidden_dim] * hidden_layers + [Dp]
        mods = []
        act = kwargs.get("activation", "relu")
        for i in range(len(sizes) - 1):
            lin = torch.nn.Linear(sizes[i], sizes[i + 1])
            # Flux's default Dense init — the reference passes no `init=` here [REF nODE.jl:12-14]: glorot_uniform
            # W ~ U(±√(6/(in+out))), bias = 0 (torch's default would be U(±1/√in) for both)
            with torch.no_grad():
                bound = (6.0 / (sizes[i] + sizes[i + 1])) ** 0.5
                lin.weight.uniform_(-bound, bound)
                lin.bias.zero_()
            mods.append(lin)
            if i < len(sizes) - 2:
                mods.append(torch.nn.ReLU() if act == "relu" else torch.nn.Tanh())
        self.dudt = torch.nn.Sequential(*mods)
        if device is not None:
            self.dudt = self.dudt.to(device)
        self.solver = solver if solver is not None else Tsit5()
        self.neural_model = NeuralODE
        self.sensealg = sensealg if sensealg is not None else InterpolatingAdjoint()
        self.latent_dim_in = latent_dim_in
        self.latent_dim_out = Dp
        self.augment_dim = augment_dim
        self.kwargs = kwargs
        self.layer_sizes = sizes
        self.batching = {"coupled": L.BATCH_COUPLED, "per_trajectory": L.BATCH_PER_TRAJECTORY,
                         "coupled_global": L.BATCH_COUPLED_GLOBAL}[batching]
        self._handle: Optional[_Handle] = None
        self._sum_cb = None

    _rhs_kind = L.RHS_MLP

    def _native(self) -> _Handle:
        if self._handle is None:
            P = 1 if self._rhs_kind == L.RHS_PENDULUM_PLUS_MLP else 0
            sa = self.sensealg if self.sensealg.code != L.SENSE_PARALLEL_CHECKPOINTED else BacksolveAdjoint()   # (time-parallel: analytic RHS only)
            d = _make_desc(self._rhs_kind, self.latent_dim_in, P, self.augment_dim, self.layer_sizes, self.solver,
                           sa, self.batching, self.kwargs)
            self._handle = _Handle(d)
        return self._handle

    def set_global_sum(self, fn, global_batch: int):
        """lde_set_global_sum_hook: `fn(vals: numpy float64 array of 1 or 2 entries)` replaces the entries in place by their sums
        over all ranks (a HOST collective — the device is busy with the solve). `fn=None` clears the hook."""
        import numpy as np
        lib = L.load()
        h = self._native()
        if fn is None:
            cb = L.SUM_HOOK(0)
        else:
            def _cb(_user, vals, n):
                try:
                    fn(np.ctypeslib.as_array(vals, shape=(n,)))
                    return 0
                except Exception:      # noqa: BLE001  (an exception must not unwind through the C frame)
                    return 1
            cb = L.SUM_HOOK(_cb)
        L.check(lib.lde_set_global_sum_hook(h.ptr, cb, None, int(global_batch)), h.ptr, "lde_set_global_sum_hook")
        self._sum_cb = cb              # keep the trampoline alive as long as the handle uses it

    def set_global_sum_peers(self, rank: int, nranks: int, mailboxes, global_batch: int):
        """lde_set_global_sum_peers: the same exchange device to device — `mailboxes` = every rank's mailbox as mapped on this device, in rank
        order (`dist.GlobalSumMailboxes(group).pointers()`; integers or tensors). The calls stay asynchronous. nranks = 0 switches it off."""
        import ctypes as C
        lib = L.load()
        h = self._native()
        if nranks == 0:
            L.check(lib.lde_set_global_sum_peers(h.ptr, 0, 0, None, 0), h.ptr, "lde_set_global_sum_peers")
            self._mailboxes = None
            return
        ptrs = [m.data_ptr() if hasattr(m, "data_ptr") else int(m) for m in mailboxes]
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        L.check(lib.lde_set_global_sum_peers(h.ptr, int(rank), int(nranks), arr, int(global_batch)), h.ptr, "lde_set_global_sum_peers")
        self._mailboxes = mailboxes       # kept alive as long as the handle uses them

    def flat_weights(self) -> torch.Tensor:
        """Flux.destructure order: per Dense layer vec(W) column-major [out×in], then b."""
        parts = []
        for m in self.dudt:
            if isinstance(m, torch.nn.Linear):
                parts.append(m.weight.t().reshape(-1))  # memory o + out*i
                parts.append(m.bias)
        return torch.cat(parts).float()


class PendulumNODE(NODE):
    """Known physics + learned correction: f = pendulum(z, L) + MLP(z) (BASELINE.json configs[2]; the
    reference only hints at it through the undefined `Pendulum_NN_friction` [REF model_train_LatentODE.jl:35]).
    Per-trajectory (GOKU) semantics; l̂ = (ẑ₀, θ̂)."""
    _rhs_kind = L.RHS_PENDULUM_PLUS_MLP

    def __init__(self, hidden_dim: int = 64, hidden_layers: int = 2, **kwargs):
        kwargs.setdefault("batching", "per_trajectory")
        super().__init__(2, hidden_dim=hidden_dim, hidden_layers=hidden_layers, **kwargs)
        self.prob = _ODEProblemStub(np.array([1.0, 1.0], np.float32), np.array([1.0], np.float32), (0.0, 1.0))


def transform_after_diffeq(x, diffeq):
    """Identity by default [REF src/models/GOKU.jl:136]; a diffeq class may define its own method
    `transform_after_diffeq(self, x)` ("mainly used for Kuramoto-like systems" [REF LatentODE.jl:74])."""
    hook = getattr(diffeq, "transform_after_diffeq", None)
    return hook(x) if hook is not None else x


# ------------------------------------------------------------------------------------------------
# Decoder container (only what the hot path needs from it)


class Decoder:
    """Decoder(model_type, (latent_out, diffeq, reconstructor))  [REF src/models/LatentDiffEqModel.jl:79-99]."""

    def __init__(self, model_type, decoder_layers):
        self.model_type = model_type
        self.latent_out, self.diffeq, self.reconstructor = decoder_layers


# ------------------------------------------------------------------------------------------------
# autograd bridge


def _as_colmajor_2d(x: torch.Tensor) -> torch.Tensor:
    """[D, B] tensor → batch-major (B, D) contiguous buffer (== the reference's column-major [D×B])."""
    xt = x.t()
    return xt if xt.is_contiguous() and xt.dtype == torch.float32 else xt.contiguous().float()


def _ts_array(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    ts = np.ascontiguousarray(np.asarray(t, dtype=np.float64))
    if ts.ndim != 1 or ts.size < 1:
        raise ValueError("t must be a non-empty 1-D sequence of save times")
    return ts


def _ptr(x: Optional[torch.Tensor]):
    return C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p()


class _SolveFn(torch.autograd.Function):
    """ẑ = solve(z0, θ, W); backward = lde_adjoint. Buffers are batch-major: z0 (B,D), θ (B,P), ẑ (T,B,D')."""

    @staticmethod
    def forward(ctx, handle: _Handle, ts: np.ndarray, z0: torch.Tensor, theta: Optional[torch.Tensor],
                W: Optional[torch.Tensor]):
        if not z0.is_cuda:
            raise L.LdeError("diffeq_layer needs CUDA/HIP tensors: the solve runs on the GPU only (no CPU fallback)")
        lib = handle.lib
        B, D = z0.shape
        T = int(ts.shape[0])
        Dp = D + handle.desc.augment_dim
        stream = L.raw_stream(z0.device.index)
        if W is not None:
            Wc = W.detach().contiguous().float()
            L.check(lib.lde_set_weights_device(handle.ptr, _ptr(Wc), Wc.numel(), stream), handle.ptr,
                    "lde_set_weights_device")
        z_out = torch.empty((T, B, Dp), device=z0.device, dtype=torch.float32)
        retcode = torch.empty((B,), device=z0.device, dtype=torch.int32)
        tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
        ctx.rec = None
        if handle.desc.sensealg == L.SENSE_DISCRETE:
            # the step record travels with THIS graph node (several forwards of one diffeq may be in flight before their pullbacks)
            nbytes = int(lib.lde_step_record_bytes(handle.ptr, B, T))
            ctx.rec = torch.empty((nbytes,), device=z0.device, dtype=torch.uint8)
            L.check(lib.lde_set_step_record(handle.ptr, _ptr(ctx.rec), nbytes), handle.ptr, "lde_set_step_record")
        try:
            L.check(lib.lde_forward(handle.ptr, _ptr(z0), _ptr(theta), tsp, T, B, _ptr(z_out), _ptr(retcode), stream),
                    handle.ptr, "lde_forward")
        finally:
            if ctx.rec is not None:       # the handle must not keep a pointer into a block this graph node owns (and torch recycles)
                lib.lde_set_step_record(handle.ptr, C.c_void_p(), 0)
        ctx.handle, ctx.ts = handle, ts
        ctx.has_theta, ctx.has_W = theta is not None, W is not None
        ctx.save_for_backward(z_out, theta if theta is not None else z0.new_empty(0))
        ctx.mark_non_differentiable(retcode)
        ctx.set_materialize_grads(False)      # retcode never has a gradient: without this autograd hands the pullback a zero-filled
                                              # int array for it — one fill launch per step
        return z_out, retcode

    @staticmethod
    def backward(ctx, dz_out, _dret):
        handle, ts = ctx.handle, ctx.ts
        lib = handle.lib
        z_out, theta = ctx.saved_tensors
        theta = theta if ctx.has_theta else None
        T, B, Dp = z_out.shape
        D = handle.desc.state_dim
        if dz_out is None:                    # (nothing downstream used ẑ)
            dz_out = torch.zeros_like(z_out)
        dz_out = dz_out.contiguous().float()
        stream = L.raw_stream(z_out.device.index)
        dz0 = torch.empty((B, D), device=z_out.device, dtype=torch.float32)
        dth = torch.empty_like(theta) if theta is not None else None
        dW = torch.zeros((handle.nW,), device=z_out.device, dtype=torch.float32) if ctx.has_W else None
        tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
        rec = ctx.rec
        if rec is not None and handle.check_record and not torch.cuda.is_current_stream_capturing():
            rec = _SolveFn._record_that_holds(handle, rec, z_out, theta, ts, stream)
        if rec is not None:
            L.check(lib.lde_set_step_record(handle.ptr, _ptr(rec), rec.numel()), handle.ptr, "lde_set_step_record")
        try:
            L.check(lib.lde_adjoint(handle.ptr, _ptr(z_out), _ptr(theta), tsp, T, B, _ptr(dz_out), _ptr(dz0), _ptr(dth),
                                    _ptr(dW), stream), handle.ptr, "lde_adjoint")
        finally:
            if rec is not None:
                lib.lde_set_step_record(handle.ptr, C.c_void_p(), 0)
        return None, None, dz0, dth, dW

    @staticmethod
    def _record_that_holds(handle: _Handle, rec: torch.Tensor, z_out: torch.Tensor, theta, ts: np.ndarray, stream) -> torch.Tensor:
        """A solve that accepted more steps than its record holds would come back from lde_adjoint as NaN gradients (never a truncated
        sweep) — and NaNs in an optimiser are for ever. The reference's ForwardDiffSensitivity differentiates any solve up to maxiters
        [REF src/models/GOKU.jl:121], so: look at the record's step counts (one device synchronisation — `diffeq.check_record = False`
        switches it off; a captured step cannot do it: GraphedStep's eager warm-up steps do), and when it overflowed raise the handle's
        "record_capacity" and repeat the forward solve — deterministic: the same steps, the same ẑ — into a record that holds it."""
        lib = handle.lib
        T, B, Dp = z_out.shape
        nmax, cap = C.c_int32(0), C.c_int32(0)
        L.check(lib.lde_step_record_status(handle.ptr, _ptr(rec), B, T, C.byref(nmax), C.byref(cap), stream), handle.ptr,
                "lde_step_record_status")
        have = (rec.numel(), int(lib.lde_step_record_bytes(handle.ptr, B, T)))
        if nmax.value <= cap.value and have[0] >= have[1]:
            return rec
        if have[0] < have[1] and nmax.value <= cap.value:
            # the capacity option was RAISED since this node's forward (another node's overflow): the record is complete but laid out for
            # the old capacity — the adjoint's view would be wrong. Re-solve into a record of today's layout.
            pass
        else:
            want = min(int(handle.desc.maxiters), max(nmax.value + 8, 2 * cap.value))
            if want <= cap.value:
                raise L.LdeError(f"the solve accepted {nmax.value} steps, more than maxiters = {handle.desc.maxiters} lets a step record hold")
            L.check(lib.lde_set_option(handle.ptr, b"record_capacity", float(want)), handle.ptr, "lde_set_option")
        nbytes = int(lib.lde_step_record_bytes(handle.ptr, B, T))
        rec2 = torch.empty((nbytes,), device=z_out.device, dtype=torch.uint8)
        z0 = z_out[0, :, :handle.desc.state_dim].contiguous()    # ẑ(t₁) is ẑ₀ itself; the augmented rows start at zero
        z_tmp = torch.empty_like(z_out)
        tsp = ts.ctypes.data_as(C.POINTER(C.c_double))
        L.check(lib.lde_set_step_record(handle.ptr, _ptr(rec2), nbytes), handle.ptr, "lde_set_step_record")
        try:
            L.check(lib.lde_forward(handle.ptr, _ptr(z0), _ptr(theta), tsp, T, B, _ptr(z_tmp), C.c_void_p(), stream), handle.ptr,
                    "lde_forward")
        finally:
            lib.lde_set_step_record(handle.ptr, C.c_void_p(), 0)
        return rec2


def solve_batch(diffeq, z0_BD: torch.Tensor, theta_BP: Optional[torch.Tensor], t) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batch-major entry point: z0 (B, D), θ (B, P) → ẑ (T, B, D'), retcode (B,). Differentiable."""
    handle = diffeq._native()
    handle.check_record = bool(getattr(diffeq, "check_record", True))     # (see _SolveFn._record_that_holds)
    W = diffeq.flat_weights()
    return _SolveFn.apply(handle, _ts_array(t), z0_BD.contiguous().float(),
                          None if theta_BP is None else theta_BP.contiguous().float(), W)


def diffeq_layer(decoder: Decoder, l_hat: Any, t) -> torch.Tensor:
    """ẑ = diffeq_layer(decoder, l̂, t)  [REF src/models/LatentDiffEqModel.jl:107].

    GOKU  (decoder.model_type isa GOKU):  l̂ = (ẑ₀ [D,B], θ̂ [P,B]) → ẑ [D, B, T]   [REF src/models/GOKU.jl:98-130]
    LatentODE:                             l̂ = ẑ₀ [D,B]           → ẑ [D', B, T]  [REF src/models/LatentODE.jl:61-78]
    A trajectory whose solve fails comes back as a NaN block; nothing is raised [REF GOKU.jl:114].
    """
    diffeq = decoder.diffeq
    if isinstance(decoder.model_type, GOKU):
        z0, theta = l_hat
        z, _ = solve_batch(diffeq, _as_colmajor_2d(z0), _as_colmajor_2d(theta), t)
        # the hook sees [D, T, B] (Array(ens)) and the result is then permuted to [D, B, T]  [REF GOKU.jl:124-125]
        z = transform_after_diffeq(z.permute(2, 0, 1), diffeq)
        return z.permute(0, 2, 1)
    if isinstance(decoder.model_type, LatentODE):
        z, _ = solve_batch(diffeq, _as_colmajor_2d(l_hat), None, t)
        # (T,B,D') buffer viewed as the reference's [D', B, T]; hook applied on that  [REF LatentODE.jl:72-75]
        return transform_after_diffeq(z.permute(2, 1, 0), diffeq)
    raise TypeError(f"no diffeq_layer method for model type {type(decoder.model_type).__name__}")
