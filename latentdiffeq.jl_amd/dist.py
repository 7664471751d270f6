"""One process per GPU: batch sharding of the solve and the single gradient all-reduce of the path.

The reference has no multi-process code at all (its only parallelism is `EnsembleThreads()` over trajectories
[REF src/models/GOKU.jl:121]). Trajectories are independent [REF GOKU.jl:111], so the batch shards by columns with NO
collective inside the solve; the only exchange of the hot path is one sum-all-reduce per optimiser step over the
gradients of parameters that every rank shares (the RHS-MLP weights dW — plus, in an end-to-end trainer, the
encoder/decoder weights): one flat fp32 buffer, one call, in place. `backend="nccl"` is RCCL over xGMI on ROCm;
`gloo` is used by the CPU tests.

Coupled (NeuralODE) control under sharding: by default each rank adapts its step size on ITS columns (shard-local norm,
SURVEY.md §8e option (i)); results then depend on the shard at the level of the solver tolerance. Fixed-step and
per-trajectory control are shard-invariant. `NODE(batching="coupled_global")` + `global_sum_hook()` is option (ii): the
step control's sums run over ALL ranks' columns (one small host all-reduce per sum) and the sharded solve equals the
unsharded one up to summation order — the parity mode, not the fast one.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of columns owned by `rank`: sizes differ by at most one, blocks tile [0, B) exactly."""
    base, rem = divmod(B, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_columns(x: Optional[torch.Tensor], rank: int, world: int, dim: int = -1) -> Optional[torch.Tensor]:
    """Slice the batch dimension of a [·, B] / [·, B, T] tensor for this rank (a view, no copy)."""
    if x is None:
        return None
    lo, hi = shard_bounds(x.shape[dim], rank, world)
    return x.narrow(dim, lo, hi - lo)


FORCE_ALLREDUCE = False   # diagnostic: run the collective on a one-rank group too (tests set the attribute; nothing here reads the environment)

class FlatGradAllReduce:
    """Data-parallel gradient sum over the parameters of a model: ONE persistent flat f32 buffer, the parameters' `.grad`
    fields are VIEWS into it (`attach()`), so a step packs and unpacks nothing — the collective runs on the buffer the
    backward wrote. The buffer is cut into `buckets` contiguous pieces in the order the backward finishes them (the last
    parameters first): each piece's all-reduce is issued asynchronously from a post-accumulate hook as soon as its last
    gradient is written, i.e. it overlaps the rest of the backward; `__call__()` (after `backward()`) issues what is left and
    waits. Zero the gradients with `zero_()` (or `zero_grad(set_to_none=False)`): setting them to None detaches the views —
    that is detected, and the step then falls back to one multi-tensor pack / unpack (`torch._foreach_copy_`), still one
    message per bucket. [north star: RCCL all-reduce of dθ once per optimiser step; REF has no multi-device path]

    Each rank's loss must already be divided by the GLOBAL batch (the loss terms take `batch_global`), so a plain sum is exact.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, buckets: int = 2, attach: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.nbuckets = max(1, int(buckets))
        self._flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []
        self._bucket_of: List[int] = []
        self._bounds: List[tuple] = []
        self._pending: List[int] = []
        self._handles: list = []
        self._hooks: list = []
        self._issued: List[bool] = []
        self._events: list = []
        if attach:
            self.attach()

    # ---- layout -------------------------------------------------------------------------------------------------------
    def _active(self) -> bool:
        # (dist.FORCE_ALLREDUCE: also with ONE rank — a one-GPU box then exercises the pack / all-reduce / unpack path over RCCL)
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or FORCE_ALLREDUCE)

    def _build(self, device) -> None:
        n = sum(p.numel() for p in self.params)
        self._flat = torch.zeros(n, dtype=torch.float32, device=device)
        self._views, off = [], 0
        for p in self.params:
            self._views.append(self._flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        # buckets: contiguous runs of parameters with about n / nbuckets elements each
        self._bucket_of, self._bounds = [], []
        target, start, acc, b = n / self.nbuckets, 0, 0, 0
        for i, p in enumerate(self.params):
            self._bucket_of.append(b)
            acc += p.numel()
            last = i == len(self.params) - 1
            if (acc - start >= target and b < self.nbuckets - 1) or last:
                self._bounds.append((start, acc))
                start, b = acc, b + 1
        self._reset_step()

    def _reset_step(self) -> None:
        nb = len(self._bounds)
        self._pending = [0] * nb
        for b in self._bucket_of:
            self._pending[b] += 1
        self._issued = [False] * nb
        self._handles = []
        self._events = [None] * len(self.params)

    def attach(self) -> "FlatGradAllReduce":
        """Allocate the flat buffer and make every parameter's .grad a view into it (zeros); install the overlap hooks."""
        if not self.params:
            return self
        self._build(self.params[0].device)
        for p, v in zip(self.params, self._views):
            p.grad = v
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        return self

    def zero_(self) -> None:
        """Zero all gradients in one fill (keeps the views attached)."""
        if self._flat is not None:
            self._flat.zero_()
            self._reset_step()

    def _attached(self) -> bool:
        return self._flat is not None and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                              for p, v in zip(self.params, self._views))

    # ---- the collective -----------------------------------------------------------------------------------------------
    def _issue(self, b: int) -> None:
        if self._issued[b]:
            return
        self._issued[b] = True
        lo, hi = self._bounds[b]
        if self._flat.is_cuda:
            # RCCL orders the collective after the ISSUING stream only, but one bucket holds gradients written on several streams
            # (encode() runs the GOKU branches on side streams; the autograd engine joins leaf streams only at the end of backward):
            # the issuing stream waits for the event each parameter's hook recorded on the stream that wrote its gradient, and for
            # the weight-gradient stream when one is set (lde_set_dw_stream).
            cur = torch.cuda.current_stream(self._flat.device)
            for i, bi in enumerate(self._bucket_of):
                if bi == b and self._events[i] is not None:
                    cur.wait_event(self._events[i])
            from . import _lib as L
            if getattr(L, "dw_stream", None) is not None:
                L.join_weight_gradients()
        self._handles.append(dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, i: int):
        def hook(param):
            if not self._active() or self._flat is None or param.grad is None or param.grad.data_ptr() != self._views[i].data_ptr():
                return
            b = self._bucket_of[i]
            if self._issued[b]:
                # a second backward() before __call__(): this bucket's sum is already on the wire, a local gradient added on top of it
                # would be silently wrong on every rank
                raise RuntimeError("FlatGradAllReduce(attach=True): backward() ran again before the reduce of the previous one was "
                                   "collected — call the reducer after every backward(), or accumulate with attach=False")
            if param.grad.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(param.grad.device))     # the stream this gradient was written on
                self._events[i] = ev
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._issue(b)
        return hook

    def __call__(self) -> None:
        if not self._active():
            return
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        if self._attached():
            for b in range(len(self._bounds)):
                self._issue(b)
            for h in self._handles:
                h.wait()
            self._reset_step()
            return
        # detached gradients (zero_grad(set_to_none=True), or never attached): one multi-tensor pack, one message, one unpack
        if len(ps) != len(self.params):
            raise RuntimeError("FlatGradAllReduce: every parameter needs a gradient (a missing one would desynchronise the ranks)")
        if self._flat is None or self._flat.device != ps[0].grad.device:
            self._build(ps[0].grad.device)
        torch._foreach_copy_(self._views, [p.grad for p in ps])
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        torch._foreach_copy_([p.grad for p in ps], self._views)
        self._reset_step()


def allreduce_flat_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of an already-flat gradient buffer (e.g. the dW written by lde_adjoint)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class LdeComm:
    """The C-ABI communicator (lde_comm_*, include/lde.h): what a `ccall` host uses for the one collective of the path.
    Rank 0 draws the id (lde_comm_unique_id) and it travels to the other ranks over the already-initialised
    torch.distributed group (any backend) — or pass `id_bytes` obtained by other means. In-place f32 sum on a HIP stream."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, id_bytes: Optional[bytes] = None, group=None):
        import ctypes as C

        from . import _lib as L
        self._L, self._C = L, C
        lib = self._lib = L.load()
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if id_bytes is None:
            box = [None]
            if rank == 0:
                buf = C.create_string_buffer(L.COMM_ID_BYTES)
                rc = lib.lde_comm_unique_id(buf)
                box[0] = (rc, buf.raw)
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            rc, id_bytes = box[0]
            if rc:
                raise L.LdeError("lde_comm_unique_id failed: " + (lib.lde_comm_last_error(None) or b"").decode())
        self.handle = C.c_void_p()
        rc = lib.lde_comm_init(C.byref(self.handle), world, rank, id_bytes)
        if rc:
            raise L.LdeError(f"lde_comm_init failed ({L.STATUS.get(rc, rc)}): " + (lib.lde_comm_last_error(None) or b"").decode())
        self.rank, self.world = rank, world

    def allreduce_(self, buf: torch.Tensor, stream=None) -> torch.Tensor:
        if buf.dtype != torch.float32 or not buf.is_cuda or not buf.is_contiguous():
            raise self._L.LdeError("lde_comm_allreduce_f32 takes a contiguous f32 device buffer")
        sp = self._L.raw_stream(buf.device.index) if stream is None else self._C.c_void_p(stream.cuda_stream)
        self._L.check(self._lib.lde_comm_allreduce_f32(self.handle, self._C.c_void_p(buf.data_ptr()), buf.numel(), sp), None,
                      "lde_comm_allreduce_f32")
        return buf

    def close(self):
        if self.handle:
            self._lib.lde_comm_destroy(self.handle)
            self.handle = self._C.c_void_p()


def global_sum_hook(group=None):
    """The `fn` of `NODE.set_global_sum` (LDE_BATCH_COUPLED_GLOBAL, SURVEY.md §8e option (ii)): sums the step-control sums of ONE
    coupled solve over the ranks that share it. `group` must be a HOST group (gloo): the hook runs while the solve's kernel occupies the
    device and waits for the answer — create one with `dist.new_group(backend="gloo", timeout=timedelta(seconds=…))` next to the RCCL
    group of the gradients. Give it a FINITE timeout: a rank whose solve has failed stops entering the exchange (after a hook failure the
    library keeps calling the hook with NaN payloads for the requests it still receives, but a failed solve issues few or none), and a
    peer's all-reduce then ends by the group's timeout — the exception makes the hook fail and that peer's solve ends with retcode ≠ 0
    instead of blocking for ever. The sums cross the boundary as doubles but are formed and consumed in float32 on the device."""
    def fn(vals):
        t = torch.from_numpy(vals)          # shares memory with the C array: the all-reduce is in place
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return fn


class GlobalSumMailboxes:
    """The device-to-device exchange of LDE_BATCH_COUPLED_GLOBAL's step-control sums (`NODE.set_global_sum_peers`, lde_set_global_sum_peers):
    one mailbox per rank in FINE-GRAINED device memory (hipExtMallocWithFlags — a kernel on another GPU sees the words while both kernels run),
    shared between the processes of one node by hipIpcGetMemHandle / hipIpcOpenMemHandle, the handles gathered over `group` (any backend:
    64 bytes per rank, once). `.pointers()` is the list every rank passes, in rank order; keep the object alive as long as the handles use it.

    One process per GPU, one node (≤ 8 ranks). tests/test_gpu_ipc_mailboxes.py runs the cross-PROCESS half on a one-GPU box: two processes
    on GPU 0 exchange their IPC handles over a gloo group, map each other's mailbox and run the two shards of one coupled solve; across
    DEVICES (peer access over xGMI) it has not run on hardware in this repository."""

    _FINEGRAINED = 0x1   # hipDeviceMallocFinegrained
    _LAZY_PEER = 0x1     # hipIpcMemLazyEnablePeerAccess

    def __init__(self, group=None, device=None):
        import ctypes as C
        from . import _lib as L
        self._C = C
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world > 8:
            raise ValueError("GlobalSumMailboxes: at most 8 ranks (one node)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._hip = C.CDLL("libamdhip64.so")
        self.nbytes = int(L.load().lde_global_sum_mailbox_bytes(self.world))
        with torch.cuda.device(self.device):
            own = C.c_void_p()
            rc = self._hip.hipExtMallocWithFlags(C.byref(own), C.c_size_t(self.nbytes), C.c_uint(self._FINEGRAINED))
            if rc != 0 or not own.value:
                raise RuntimeError(f"hipExtMallocWithFlags(fine-grained mailbox) failed: {rc}")
            if self._hip.hipMemset(own, 0, C.c_size_t(self.nbytes)) != 0:
                raise RuntimeError("hipMemset(mailbox) failed")
            torch.cuda.synchronize()
            self._own = own
            self._ptrs = [None] * self.world
            self._ptrs[self.rank] = own.value
            self._opened = []
            if self.world > 1:
                # hipIpcMemHandle_t is a 64-byte struct and hipIpcOpenMemHandle takes it BY VALUE (ctypes would pass an array as a pointer:
                # the callee would read a garbage handle — ADVICE r5)
                class hipIpcMemHandle_t(C.Structure):
                    _fields_ = [("reserved", C.c_char * 64)]
                hip = self._hip
                hip.hipIpcGetMemHandle.argtypes = [C.POINTER(hipIpcMemHandle_t), C.c_void_p]
                hip.hipIpcGetMemHandle.restype = C.c_int
                hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), hipIpcMemHandle_t, C.c_uint]
                hip.hipIpcOpenMemHandle.restype = C.c_int
                hip.hipIpcCloseMemHandle.argtypes = [C.c_void_p]
                hip.hipIpcCloseMemHandle.restype = C.c_int
                handle = hipIpcMemHandle_t()
                rc = hip.hipIpcGetMemHandle(C.byref(handle), own)
                if rc != 0:
                    raise RuntimeError(f"hipIpcGetMemHandle(mailbox) failed: {rc}")
                mine = torch.frombuffer(bytearray(bytes(handle)), dtype=torch.uint8).clone()
                backend = dist.get_backend(group)
                if backend == "nccl":
                    mine = mine.to(self.device)
                gathered = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(gathered, mine, group=group)
                for r, hb in enumerate(gathered):
                    if r == self.rank:
                        continue
                    hr = hipIpcMemHandle_t.from_buffer_copy(bytes(hb.cpu().tolist()))
                    p = C.c_void_p()
                    rc = hip.hipIpcOpenMemHandle(C.byref(p), hr, C.c_uint(self._LAZY_PEER))
                    if rc != 0 or not p.value:
                        raise RuntimeError(f"hipIpcOpenMemHandle(rank {r}'s mailbox) failed: {rc}")
                    self._ptrs[r] = p.value
                    self._opened.append(p)
                dist.barrier(group=group)      # every rank has mapped every mailbox before anyone's kernel writes

    def pointers(self):
        return list(self._ptrs)

    def close(self):
        C = self._C
        for p in getattr(self, "_opened", []):
            self._hip.hipIpcCloseMemHandle(p)
        self._opened = []
        if getattr(self, "_own", None) is not None:
            self._hip.hipFree(self._own)
            self._own = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


def diffeq_layer_sharded(decoder, l_hat, t, rank: Optional[int] = None, world: Optional[int] = None):
    """diffeq_layer on this rank's block of columns: returns the LOCAL shard ẑ[:, lo:hi, :] (it feeds the local shard of
    the reconstructor). No communication."""
    from .api import GOKU, diffeq_layer
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    if isinstance(decoder.model_type, GOKU):
        z0, theta = l_hat
        local = (shard_columns(z0, rank, world, 1), shard_columns(theta, rank, world, 1))
    else:
        local = shard_columns(l_hat, rank, world, 1)
    return diffeq_layer(decoder, local, t)
