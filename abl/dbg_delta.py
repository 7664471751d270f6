import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, ".")
from latentdiffeq_amd import _lib as L
from tests.gpu_util import NativeChain
from latentdiffeq_amd import synthetic as S
N=int(sys.argv[1]) if len(sys.argv)>1 else 64
sizes, acts, skips = (2, 200, 200, 200, 784), (L.CACT_RELU, L.CACT_RELU, L.CACT_RELU, L.CACT_SIGMOID), (0, 1, 1, 0)
rng = np.random.default_rng(3)
W = S.mlp_weights(sizes, seed=11)
x = torch.from_numpy((0.7 * rng.standard_normal((N, 2))).astype(np.float32)).cuda()
tgt = torch.from_numpy(rng.uniform(0, 1, (N, 784)).astype(np.float32)).cuda()
ch = NativeChain(sizes, acts, skips); ch.set_weights(W); ch.set_dtype("bf16")
lib, h = ch.lib, ch.h
lib.lde_chain_saved_floats.restype = C.c_int64; lib.lde_chain_mse_scratch_floats.restype = C.c_int64
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
scale = 1.0 / N
nsv, nws = int(lib.lde_chain_saved_floats(h, N)), int(lib.lde_chain_mse_scratch_floats(h, N)) + 1
def run(delta):
    y, saved, ws = torch.empty((N, 784), device="cuda"), torch.empty((nsv,), device="cuda"), torch.empty((nws,), device="cuda")
    dx, dW, gd = torch.empty_like(x), torch.zeros((ch.nW,), device="cuda"), torch.tensor([1.0], device="cuda")
    if delta:
        L.check(lib.lde_chain_forward_save_mse_delta(h, p(x), N, p(y), p(saved), p(tgt), scale, C.c_void_p(), p(ws), C.c_void_p(ws.data_ptr() + 4), s), h, "f", chain=True)
        L.check(lib.lde_chain_backward_saved_delta(h, p(x), p(gd), p(saved), N, p(dx), p(dW), s), h, "b", chain=True)
    else:
        L.check(lib.lde_chain_forward_save_mse(h, p(x), N, p(y), p(saved), p(tgt), scale, C.c_void_p(), p(ws), C.c_void_p(ws.data_ptr() + 4), s), h, "f", chain=True)
        L.check(lib.lde_chain_backward_saved_mse(h, p(x), p(y), p(tgt), p(gd), scale, C.c_void_p(), p(saved), N, p(dx), p(dW), s), h, "b", chain=True)
    torch.cuda.synchronize()
    return dx.cpu().numpy(), dW.cpu().numpy()
dx0, dW0 = run(False); dx1, dW1 = run(True)
bL0, bL1 = dW0[-784:], dW1[-784:]
print("N", N, "bias grad last layer ref[:8]", bL0[:8], "delta[:8]", bL1[:8])
print("ratio", (bL1/bL0)[:8], "corr", np.corrcoef(bL0, bL1)[0,1], "norms", np.linalg.norm(bL0), np.linalg.norm(bL1))
print("dx rel", np.abs(dx1-dx0).max()/np.abs(dx0).max(), "dW rel", np.abs(dW1-dW0).max()/np.abs(dW0).max())
