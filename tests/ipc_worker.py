"""One rank of tests/test_gpu_ipc_mailboxes.py: a PROCESS of its own on GPU 0 that maps the other ranks' mailboxes through HIP IPC
(dist.GlobalSumMailboxes over a gloo group) and solves its shard of ONE coupled solve under LDE_BATCH_COUPLED_GLOBAL with the sums
exchanged device to device (lde_set_global_sum_peers). Not a test module: started by the test, one per rank.

    python tests/ipc_worker.py RANK WORLD PORT OUTDIR MODE SENSE LO HI B
MODE = solve | leave (rank > 0 maps the mailboxes, then exits without solving: the remaining rank must be poisoned, not hang)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir, mode, sense = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    lo, hi, B = int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    from latentdiffeq_amd import _lib as L
    from latentdiffeq_amd import dist as D
    from oracle import oracle as O
    from tests.gpu_util import Native, make_desc
    from tests.test_gpu_coupled_global import KW, LAYERS, _inputs

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    W = O.mlp_weights(LAYERS, seed=3)
    z0, ts, dz = _inputs(B)
    z0[80:] *= 3.0
    more = dict(sensealg=O.SENSE_DISCRETE) if sense == "discrete" else {}
    # warm-up on a handle that exchanges nothing: the module is loaded and the same kernel instantiations have run before the first
    # exchanging call (a rank that enters seconds late would be taken for gone — include/lde.h)
    warm = Native(make_desc(batching=O.BATCH_COUPLED, **KW, **more))
    warm.set_weights(W)
    zw, _, _ = warm.forward(z0[lo:hi], None, ts)
    warm.adjoint(zw, None, ts, dz[:, lo:hi])
    warm.close()

    nat = Native(make_desc(batching=O.BATCH_COUPLED_GLOBAL, **KW, **more))
    nat.set_weights(W)
    boxes = D.GlobalSumMailboxes()                      # hipExtMallocWithFlags + hipIpcGetMemHandle / hipIpcOpenMemHandle over the gloo group
    assert boxes.world == world and boxes.rank == rank and all(p for p in boxes.pointers())
    assert len(set(boxes.pointers())) == world         # the peers' mailboxes are mapped at addresses of their own
    ptrs = (C.c_void_p * world)(*boxes.pointers())
    L.check(nat.lib.lde_set_global_sum_peers(nat.h, rank, world, ptrs, B), nat.h, "lde_set_global_sum_peers")
    assert nat.lib.lde_reserve(nat.h, hi - lo, len(ts)) == 0
    if mode == "leave":
        nat.set_option("peer_spin_k", 256)              # ≈ 0.3 s instead of ≈ 10 s
    torch.cuda.synchronize()
    dist.barrier()
    if mode == "leave" and rank > 0:
        dist.destroy_process_group()
        return
    z, ret, st = nat.forward(z0[lo:hi], None, ts)
    out = dict(z=z, ret=ret, naccept=st["naccept"], nfailed=st["nfailed"])
    if mode == "solve":
        g0, _, gW, sb = nat.adjoint(z, None, ts, dz[:, lo:hi])
        out.update(g0=g0, gW=gW, badj=sb["naccept"], adj_failed=sb["nfailed"])
        # a second pair of calls on the same handles: the word sets alternate by the exchange's launch parity
        z2, ret2, _ = nat.forward(z0[lo:hi], None, ts)
        out.update(z2=z2, ret2=ret2)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    if mode == "solve":
        dist.barrier()
    nat.close()
    boxes.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
