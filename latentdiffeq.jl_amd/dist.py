"""One process per GPU: batch sharding of the solve and the single gradient all-reduce of the path.

The reference has no multi-process code at all (its only parallelism is `EnsembleThreads()` over trajectories
[REF src/models/GOKU.jl:121]). Trajectories are independent [REF GOKU.jl:111], so the batch shards by columns with NO
collective inside the solve; the only exchange of the hot path is one sum-all-reduce per optimiser step over the
gradients of parameters that every rank shares (the RHS-MLP weights dW — plus, in an end-to-end trainer, the
encoder/decoder weights): one flat fp32 buffer, one call, in place. `backend="nccl"` is RCCL over xGMI on ROCm;
`gloo` is used by the CPU tests.

Coupled (NeuralODE) control under sharding: each rank adapts its step size on ITS columns (shard-local norm,
SURVEY.md §8e option (i)); results then depend on the shard at the level of the solver tolerance. Fixed-step and
per-trajectory control are shard-invariant.
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


class FlatGradAllReduce:
    """Sum-all-reduce the .grad of a set of shared parameters as ONE flat fp32 message.

    The loss is a mean over the global batch [REF examples/pendulum_friction-less/model_train.jl:232]: fold the 1/N
    into the cotangent (each rank's loss divides by the GLOBAL batch) and a plain sum is exact.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self._flat: Optional[torch.Tensor] = None

    def __call__(self) -> None:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        n = sum(p.grad.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n or self._flat.device != ps[0].grad.device:
            self._flat = torch.empty(n, dtype=torch.float32, device=ps[0].grad.device)
        off = 0
        for p in ps:
            k = p.grad.numel()
            self._flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in ps:
            k = p.grad.numel()
            p.grad.copy_(self._flat[off:off + k].view_as(p.grad))
            off += k


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
