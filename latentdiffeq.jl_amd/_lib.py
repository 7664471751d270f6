"""ctypes binding of liblde.so — the only way the Python host code reaches the solver.

There is deliberately NO fallback: if the HIP library is missing or fails to load, every entry point
raises. (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# LDE_LIB_PATH: load another build of the same library (diagnostic / ablation builds made with -DLDE_ABL=n)
LIB_PATH = os.environ.get("LDE_LIB_PATH") or os.path.join(HERE, "liblde.so")

LDE_ABI_VERSION = 1
LDE_MAX_LAYERS = 6

RHS_PENDULUM, RHS_PENDULUM_FRICTION, RHS_MLP, RHS_PENDULUM_PLUS_MLP = 0, 1, 2, 3
SOLVER_TSIT5, SOLVER_RK4 = 0, 1
BATCH_PER_TRAJECTORY, BATCH_COUPLED, BATCH_COUPLED_GLOBAL = 0, 1, 2
SENSE_BACKSOLVE_CHECKPOINTED, SENSE_BACKSOLVE, SENSE_PARALLEL_CHECKPOINTED, SENSE_DISCRETE = 0, 1, 2, 3
ACT_RELU, ACT_TANH = 0, 1

STATUS = {0: "LDE_OK", -1: "LDE_ERR_INVALID_ARG", -2: "LDE_ERR_UNSUPPORTED", -3: "LDE_ERR_NO_DEVICE",
          -4: "LDE_ERR_HIP", -5: "LDE_ERR_NO_WEIGHTS", -6: "LDE_ERR_ALLOC"}

# every symbol include/lde.h declares
EXPORTS = ["lde_abi_version", "lde_problem_desc_default", "lde_num_weights", "lde_create", "lde_destroy",
           "lde_build_info", "lde_global_sum_mailbox_bytes", "lde_set_global_sum_peers", "lde_set_weights", "lde_set_weights_device", "lde_reserve", "lde_forward", "lde_adjoint",
           "lde_get_stats", "lde_last_error", "lde_last_kernel", "lde_set_global_sum_hook", "lde_set_phase_timing", "lde_get_phase_ms",
           "lde_step_record_bytes", "lde_set_step_record", "lde_get_step_record", "lde_step_record_capacity", "lde_step_record_status", "lde_set_option", "lde_get_option",
           "lde_chain_num_weights", "lde_chain_create", "lde_chain_destroy", "lde_chain_set_weights",
           "lde_chain_set_weights_device", "lde_chain_reserve", "lde_chain_forward", "lde_chain_backward",
           "lde_chain_last_error", "lde_chain_set_option", "lde_rnn_set_option", "lde_chain_set_accumulate", "lde_chain_set_dtype", "lde_rnn_set_accumulate", "lde_chain_saved_floats", "lde_chain_forward_save", "lde_chain_backward_saved", "lde_chain_group_forward_save", "lde_chain_group_backward_saved", "lde_rnn_group_forward", "lde_rnn_group_backward", "lde_rnn_forward_train", "lde_rnn_group_forward_train", "lde_chain_backward_saved_sum", "lde_chain_backward_saved_mse", "lde_chain_forward_save_mse", "lde_chain_forward_save_mse_delta", "lde_chain_backward_saved_delta", "lde_chain_delta_is_staged", "lde_chain_mse_scratch_floats", "lde_randn", "lde_sample_kl_pair_forward", "lde_sample_kl_pair_backward", "lde_rnn_group_forward_ld", "lde_rnn_group_backward_ld",
           "lde_rnn_num_weights", "lde_rnn_create", "lde_rnn_destroy", "lde_rnn_set_weights", "lde_rnn_set_weights_device",
           "lde_rnn_reserve", "lde_rnn_forward", "lde_rnn_backward", "lde_rnn_last_error", "lde_rnn_backward_dx", "lde_rnn_backward_dw", "lde_refresh_weights",
           "lde_sample_forward", "lde_sample_backward", "lde_kl_forward", "lde_kl_backward", "lde_mse_forward",
           "lde_mse_backward", "lde_sample_kl_forward", "lde_sample_kl_backward", "lde_mse_forward_add", "lde_adamw_flux_step", "lde_adamw_flux_step_dev", "lde_set_dw_stream", "lde_join_dw",
           "lde_comm_unique_id", "lde_comm_init", "lde_comm_allreduce_f32", "lde_comm_nranks", "lde_comm_rank",
           "lde_comm_destroy", "lde_comm_last_error"]
# lde_sum_hook: int hook(void* user, double* vals, int n) — vals[0..n) ← Σ over ranks, in place
SUM_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
COMM_ID_BYTES = 128
LOSS_SCRATCH_FLOATS = 1024

LDE_RNN_MAX_LAYERS = 4
CELL_RNN_RELU, CELL_RNN_TANH, CELL_LSTM = 0, 1, 2


class RnnDesc(C.Structure):
    """lde_rnn_desc (include/lde.h)."""

    _fields_ = [("abi_version", C.c_int32), ("cell", C.c_int32), ("n_layers", C.c_int32),
                ("sizes", C.c_int32 * (LDE_RNN_MAX_LAYERS + 1)), ("reverse", C.c_int32)]

LDE_CHAIN_MAX_LAYERS = 6
CACT_IDENTITY, CACT_RELU, CACT_TANH, CACT_SIGMOID, CACT_SOFTPLUS = 0, 1, 2, 3, 4
DTYPE_F32, DTYPE_BF16 = 0, 1


class ChainDesc(C.Structure):
    """lde_chain_desc (include/lde.h)."""

    _fields_ = [("abi_version", C.c_int32), ("n_layers", C.c_int32), ("sizes", C.c_int32 * (LDE_CHAIN_MAX_LAYERS + 1)),
                ("activation", C.c_int32 * LDE_CHAIN_MAX_LAYERS), ("skip", C.c_int32 * LDE_CHAIN_MAX_LAYERS)]


class ProblemDesc(C.Structure):
    """lde_problem_desc (include/lde.h)."""

    _fields_ = [
        ("abi_version", C.c_int32), ("rhs_kind", C.c_int32), ("state_dim", C.c_int32), ("param_dim", C.c_int32),
        ("augment_dim", C.c_int32), ("n_layers", C.c_int32), ("layer_sizes", C.c_int32 * (LDE_MAX_LAYERS + 1)),
        ("activation", C.c_int32), ("solver", C.c_int32), ("batching", C.c_int32), ("sensealg", C.c_int32),
        ("adaptive", C.c_int32), ("maxiters", C.c_int64), ("dt", C.c_double), ("abstol", C.c_double),
        ("reltol", C.c_double), ("dtmin", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double),
        ("gamma", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
    ]


class Stats(C.Structure):
    """lde_stats (include/lde.h)."""

    _fields_ = [("nfe", C.c_int64), ("naccept", C.c_int64), ("nreject", C.c_int64), ("nfailed", C.c_int64),
                ("max_steps", C.c_int64)]


class LdeError(RuntimeError):
    pass


_lib = None


def load():
    """Load liblde.so (once). Raises if it is missing: there is no CPU path in the product."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdeError(f"{LIB_PATH} not found — build it with `python latentdiffeq.jl_amd/build.py` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: liblde.so and torch must share ONE HIP runtime (device pointers and streams cross the boundary), and
    # the process uses whichever libamdhip64 is loaded first — torch bundles its own. Loading liblde.so before torch
    # binds it to /opt/rocm's copy and leaves two runtimes in the process (lde_create then reports LDE_ERR_NO_DEVICE).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.lde_abi_version.restype = i32
    lib.lde_build_info.restype = C.c_char_p
    lib.lde_build_info.argtypes = []
    lib.lde_problem_desc_default.argtypes = [C.POINTER(ProblemDesc)]
    lib.lde_num_weights.argtypes = [C.POINTER(ProblemDesc)]
    lib.lde_num_weights.restype = i64
    lib.lde_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(vp)]
    lib.lde_destroy.argtypes = [vp]
    lib.lde_destroy.restype = None
    lib.lde_set_weights.argtypes = [vp, vp, i64]
    lib.lde_set_weights_device.argtypes = [vp, vp, i64, vp]
    lib.lde_reserve.argtypes = [vp, i32, i32]
    lib.lde_forward.argtypes = [vp, vp, vp, C.POINTER(C.c_double), i32, i32, vp, vp, vp]
    lib.lde_adjoint.argtypes = [vp, vp, vp, C.POINTER(C.c_double), i32, i32, vp, vp, vp, vp, vp]
    lib.lde_get_stats.argtypes = [vp, i32, C.POINTER(Stats), vp]
    lib.lde_set_phase_timing.argtypes = [vp, C.c_int]
    lib.lde_get_phase_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.lde_set_global_sum_hook.argtypes = [vp, SUM_HOOK, vp, C.c_int64]
    lib.lde_set_global_sum_hook.restype = C.c_int
    lib.lde_global_sum_mailbox_bytes.argtypes = [i32]
    lib.lde_global_sum_mailbox_bytes.restype = C.c_int64
    lib.lde_set_global_sum_peers.argtypes = [vp, i32, i32, C.POINTER(vp), C.c_int64]
    lib.lde_set_global_sum_peers.restype = C.c_int
    lib.lde_last_error.argtypes = [vp]
    lib.lde_last_error.restype = C.c_char_p
    lib.lde_last_kernel.argtypes = [vp, i32]
    lib.lde_last_kernel.restype = C.c_char_p
    lib.lde_step_record_bytes.argtypes = [vp, i32, i32]
    lib.lde_step_record_bytes.restype = i64
    lib.lde_set_step_record.argtypes = [vp, vp, i64]
    lib.lde_get_step_record.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp]
    lib.lde_step_record_capacity.argtypes = [vp, i32]
    lib.lde_step_record_status.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    lib.lde_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.lde_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    lib.lde_chain_num_weights.argtypes = [C.POINTER(ChainDesc)]
    lib.lde_chain_num_weights.restype = i64
    lib.lde_chain_create.argtypes = [C.POINTER(ChainDesc), C.POINTER(vp)]
    lib.lde_chain_destroy.argtypes = [vp]
    lib.lde_chain_destroy.restype = None
    lib.lde_chain_set_weights.argtypes = [vp, vp, i64]
    lib.lde_chain_set_weights_device.argtypes = [vp, vp, i64, vp]
    lib.lde_chain_reserve.argtypes = [vp, i64]
    lib.lde_chain_forward.argtypes = [vp, vp, i64, vp, vp]
    lib.lde_chain_backward.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
    lib.lde_chain_last_error.argtypes = [vp]
    lib.lde_chain_last_error.restype = C.c_char_p
    lib.lde_chain_set_accumulate.argtypes = [vp, i32]
    lib.lde_chain_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.lde_rnn_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.lde_chain_set_dtype.argtypes = [vp, i32]
    lib.lde_rnn_set_accumulate.argtypes = [vp, i32]
    lib.lde_chain_saved_floats.argtypes = [vp, i64]
    lib.lde_chain_saved_floats.restype = i64
    lib.lde_chain_forward_save.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.lde_chain_backward_saved.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp]
    lib.lde_chain_group_forward_save.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    lib.lde_rnn_group_forward.argtypes = [i32, vp, vp, i32, i32, vp, vp]
    lib.lde_rnn_group_forward_train.argtypes = [i32, vp, vp, i32, i32, vp, vp]
    lib.lde_rnn_forward_train.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.lde_rnn_group_backward.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.lde_rnn_group_forward_ld.argtypes = [i32, vp, vp, i32, i32, vp, vp, i32, vp]
    lib.lde_rnn_group_backward_ld.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.lde_chain_backward_saved_sum.argtypes = [vp, vp, vp, i32, vp, vp, i64, vp, vp, vp]
    lib.lde_chain_backward_saved_mse.argtypes = [vp, vp, vp, vp, vp, C.c_float, vp, vp, i64, vp, vp, vp]
    lib.lde_sample_kl_pair_forward.argtypes = [vp, vp, i64, C.c_float, vp, vp, i64, C.c_float, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lde_sample_kl_pair_backward.argtypes = [vp, vp, vp, vp, i64, C.c_float, vp, vp, vp, vp, i64, C.c_float, vp, vp, vp, vp, vp, vp]
    lib.lde_chain_forward_save_mse.argtypes = [vp, vp, i64, vp, vp, vp, C.c_float, vp, vp, vp, vp]
    lib.lde_chain_forward_save_mse_delta.argtypes = [vp, vp, i64, vp, vp, vp, C.c_float, vp, vp, vp, vp]
    lib.lde_chain_backward_saved_delta.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
    lib.lde_chain_delta_is_staged.argtypes = [vp, vp, i64]
    lib.lde_chain_mse_scratch_floats.argtypes = [vp, i64]
    lib.lde_chain_mse_scratch_floats.restype = i64
    lib.lde_randn.argtypes = [vp, i64, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp]
    lib.lde_chain_group_backward_saved.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.lde_rnn_num_weights.argtypes = [C.POINTER(RnnDesc)]
    lib.lde_rnn_num_weights.restype = i64
    lib.lde_rnn_create.argtypes = [C.POINTER(RnnDesc), C.POINTER(vp)]
    lib.lde_rnn_destroy.argtypes = [vp]
    lib.lde_rnn_destroy.restype = None
    lib.lde_rnn_set_weights.argtypes = [vp, vp, i64]
    lib.lde_rnn_set_weights_device.argtypes = [vp, vp, i64, vp]
    lib.lde_rnn_reserve.argtypes = [vp, i32, i32]
    lib.lde_rnn_forward.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.lde_rnn_backward.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.lde_rnn_backward_dx.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.lde_rnn_backward_dw.argtypes = [vp, vp, vp]
    lib.lde_rnn_last_error.argtypes = [vp]
    lib.lde_rnn_last_error.restype = C.c_char_p
    lib.lde_refresh_weights.argtypes = [i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), vp]
    f32 = C.c_float
    lib.lde_sample_forward.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.lde_sample_backward.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.lde_kl_forward.argtypes = [vp, vp, i64, f32, vp, vp, vp]
    lib.lde_kl_backward.argtypes = [vp, vp, i64, f32, vp, vp, vp, vp]
    lib.lde_mse_forward.argtypes = [vp, vp, i64, f32, vp, vp, vp]
    lib.lde_mse_backward.argtypes = [vp, vp, i64, f32, vp, vp, vp]
    lib.lde_sample_kl_forward.argtypes = [vp, vp, vp, i64, f32, vp, vp, vp, vp, vp]
    lib.lde_sample_kl_backward.argtypes = [vp, vp, vp, vp, vp, f32, i64, vp, vp, vp]
    lib.lde_mse_forward_add.argtypes = [vp, vp, i64, f32, vp, vp, vp, vp]
    lib.lde_set_dw_stream.argtypes = [vp]
    lib.lde_join_dw.argtypes = [vp]
    lib.lde_adamw_flux_step.argtypes = [i32, C.POINTER(AdamTensor), f32, f32, f32, f32, f32, i64, vp]
    lib.lde_adamw_flux_step_dev.argtypes = [i32, C.POINTER(AdamTensor), f32, f32, f32, f32, f32, vp, vp]
    lib.lde_comm_unique_id.argtypes = [C.c_char_p]
    lib.lde_comm_init.argtypes = [C.POINTER(vp), i32, i32, C.c_char_p]
    lib.lde_comm_allreduce_f32.argtypes = [vp, vp, i64, vp]
    lib.lde_comm_nranks.argtypes = [vp]
    lib.lde_comm_rank.argtypes = [vp]
    lib.lde_comm_destroy.argtypes = [vp]
    lib.lde_comm_destroy.restype = None
    lib.lde_comm_last_error.argtypes = [vp]
    lib.lde_comm_last_error.restype = C.c_char_p
    if lib.lde_abi_version() != LDE_ABI_VERSION:
        raise LdeError("liblde.so ABI version mismatch — rebuild")
    _lib = lib
    return lib


def raw_stream(device_index=None) -> C.c_void_p:
    """The current HIP stream of the device as a raw pointer (torch.cuda.current_stream() builds a Stream object and
    resolves the device three times: ≈ 10 µs per call, twenty calls per training step)."""
    import torch
    if device_index is None:
        device_index = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(device_index))


MODULE_CHAIN, MODULE_RNN = 0, 1

dw_stream = None        # torch.cuda.Stream the chain / recurrent pullbacks enqueue their weight-gradient kernels on (or None)


def set_async_weight_gradients(on: bool = True, device=None):
    """lde_set_dw_stream: the pullbacks of Chain / Recurrent put their weight-gradient kernels on a stream of their own, off the
    critical path of the backward pass. The gradients of the parameters are then complete only after `join_weight_gradients()` —
    call it after `loss.backward()` and before anything reads `.grad` (optimiser step, all-reduce, clipping). Opt-in."""
    global dw_stream
    import torch
    lib = load()
    if on:
        if dw_stream is None:
            dw_stream = torch.cuda.Stream(device)
        check(lib.lde_set_dw_stream(C.c_void_p(dw_stream.cuda_stream)), None, "lde_set_dw_stream")
    else:
        if dw_stream is not None:
            join_weight_gradients()
        check(lib.lde_set_dw_stream(None), None, "lde_set_dw_stream")
        dw_stream = None


def join_weight_gradients():
    """lde_join_dw on the current stream (a device-side wait; the host does not block). No-op when the mode is off."""
    if dw_stream is not None:
        check(load().lde_join_dw(raw_stream(dw_stream.device.index)), None, "lde_join_dw")


class AdamTensor(C.Structure):
    """lde_adam_tensor (include/lde.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


def weights_key(W):
    """Identity of a parameter's current VALUE as torch tracks it: storage address + in-place version counter (an optimiser
    step, `copy_`, `fill_` … bump it; writes through `.data` do not — torch's own saved-tensor checks share that blind spot)."""
    return (W.data_ptr(), W._version)


def refresh_weights(modules, stream=None):
    """lde_refresh_weights for torch modules that own a native handle (Chain / Recurrent): ONE launch re-lays-out the weights of
    all of them; each module then skips its own per-call upload until its parameter changes again (tracked by `weights_key`).
    Call it after the optimiser step. Modules on the CPU or without CUDA are left alone."""
    import torch
    mods = [m for m in modules if hasattr(m, "_native") and hasattr(m, "theta") and m.theta.is_cuda]
    if not mods:
        return 0
    lib = load()
    n = len(mods)
    kinds, handles, ptrs, keep = (C.c_int32 * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(), []
    for i, m in enumerate(mods):
        W = m.theta.detach()
        if W.dtype != torch.float32 or not W.is_contiguous():
            W = W.contiguous().float()
        keep.append(W)
        kinds[i] = MODULE_RNN if getattr(m, "_is_recurrent", False) else MODULE_CHAIN
        handles[i] = m._native().value
        ptrs[i] = W.data_ptr()
    check(lib.lde_refresh_weights(n, kinds, handles, ptrs, stream if stream is not None else raw_stream(mods[0].theta.device.index)),
          None, "lde_refresh_weights")
    for m, W in zip(mods, keep):
        m._wkey = weights_key(m.theta) if W.data_ptr() == m.theta.data_ptr() else None
    return n


def check(rc: int, handle=None, what: str = "", chain: bool = False, rnn: bool = False):
    if rc == 0:
        return
    msg = STATUS.get(rc, str(rc))
    if handle is not None and _lib is not None:
        detail = (_lib.lde_rnn_last_error if rnn else _lib.lde_chain_last_error if chain else _lib.lde_last_error)(handle)
        if detail:
            msg += ": " + detail.decode()
    raise LdeError(f"{what} failed: {msg}")
