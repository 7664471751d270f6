"""GPU: the C-ABI collective (lde_comm_*). One GPU per box here, so the communicator has ONE rank: RCCL binds, the
communicator initialises, and the in-place sum over one rank is the identity; the N-rank arithmetic is covered by the
gloo tests (tests/test_dist_gloo.py) and the launch path by tests/test_bench_launch.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_allreduce_is_identity():
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd.dist import LdeComm
    c = LdeComm(rank=0, world=1)
    lib = L.load()
    assert lib.lde_comm_nranks(c.handle) == 1 and lib.lde_comm_rank(c.handle) == 0
    x = torch.randn(24864, device="cuda")
    ref = x.clone()
    c.allreduce_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    assert lib.lde_comm_allreduce_f32(c.handle, C.c_void_p(), 0, C.c_void_p()) == 0          # empty message
    assert lib.lde_comm_allreduce_f32(c.handle, C.c_void_p(), 5, C.c_void_p()) == -1         # NULL buffer → INVALID_ARG
    c.close()


def test_comm_init_validates_arguments():
    from latentdiffeq_amd import _lib as L
    lib = L.load()
    h = C.c_void_p()
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    assert lib.lde_comm_unique_id(buf) == 0
    assert lib.lde_comm_init(C.byref(h), 2, 2, buf.raw) == -1 and not h      # rank out of range
    assert lib.lde_comm_init(C.byref(h), 0, 0, buf.raw) == -1 and not h
    assert b"rank" in lib.lde_comm_last_error(None)
