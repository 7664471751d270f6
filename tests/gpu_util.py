"""Helpers for the `-m gpu` parity tests: call liblde.so through its C ABI with torch-owned device buffers."""
import ctypes as C

import numpy as np
import torch

from latentdiffeq_amd import _lib as L


def make_desc(**kw):
    lib = L.load()
    d = L.ProblemDesc()
    lib.lde_problem_desc_default(C.byref(d))
    # the library's default is LDE_SENSE_DISCRETE (the reference's ForwardDiffSensitivity; tests/test_gpu_discrete.py, test_gpu_default_sensealg.py);
    # a parity test that says nothing about the sensealg checks the CONTINUOUS adjoint against the oracle's reverse-time solve
    # (time-parallel on the GOKU path, the sequential checkpointed one elsewhere: lde_create's mapping)
    kw.setdefault("sensealg", L.SENSE_PARALLEL_CHECKPOINTED)
    layers = kw.pop("layers", ())
    d.n_layers = max(len(layers) - 1, 0)
    for i, s in enumerate(layers):
        d.layer_sizes[i] = s
    for k, v in kw.items():
        assert hasattr(d, k), k
        setattr(d, k, v)
    return d


def copy_desc_to_oracle(d):
    """Same bytes, oracle's own ctypes type."""
    from oracle import oracle as O
    od = O.Desc()
    C.memmove(C.byref(od), C.byref(d), C.sizeof(od))
    assert C.sizeof(od) == C.sizeof(d)
    return od


class Native:
    """Thin RAII over the C ABI, numpy in / numpy out (device staging through torch)."""

    def __init__(self, desc):
        self.lib = L.load()
        self.d = desc
        self.h = C.c_void_p()
        L.check(self.lib.lde_create(C.byref(desc), C.byref(self.h)), None, "lde_create")
        self.nW = int(self.lib.lde_num_weights(C.byref(desc)))

    def close(self):
        if self.h:
            self.lib.lde_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()

    def set_weights(self, W):
        W = np.ascontiguousarray(W, np.float32)
        L.check(self.lib.lde_set_weights(self.h, W.ctypes.data_as(C.c_void_p), W.size), self.h, "lde_set_weights")

    def forward(self, z0, theta, ts):
        dev = "cuda"
        z0d = torch.from_numpy(np.ascontiguousarray(z0, np.float32)).to(dev)
        thd = None if theta is None else torch.from_numpy(np.ascontiguousarray(theta, np.float32)).to(dev)
        B, D = z0d.shape
        ts = np.ascontiguousarray(ts, np.float64)
        T = ts.shape[0]
        Dp = D + self.d.augment_dim
        out = torch.full((T, B, Dp), 7.0, device=dev, dtype=torch.float32)
        ret = torch.full((B,), -1, device=dev, dtype=torch.int32)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_forward(self.h, self._p(z0d), self._p(thd), ts.ctypes.data_as(C.POINTER(C.c_double)), T, B,
                                     self._p(out), self._p(ret), s), self.h, "lde_forward")
        torch.cuda.synchronize()
        return out.cpu().numpy(), ret.cpu().numpy(), self.stats(0)

    def adjoint(self, z_out, theta, ts, dz_out):
        dev = "cuda"
        zo = torch.from_numpy(np.ascontiguousarray(z_out, np.float32)).to(dev)
        dzo = torch.from_numpy(np.ascontiguousarray(dz_out, np.float32)).to(dev)
        thd = None if theta is None else torch.from_numpy(np.ascontiguousarray(theta, np.float32)).to(dev)
        T, B, Dp = zo.shape
        D, P = self.d.state_dim, self.d.param_dim
        ts = np.ascontiguousarray(ts, np.float64)
        dz0 = torch.full((B, D), 7.0, device=dev)
        dth = torch.full((B, P), 7.0, device=dev) if P else None
        dW = torch.zeros((self.nW,), device=dev) if self.nW else None
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_adjoint(self.h, self._p(zo), self._p(thd), ts.ctypes.data_as(C.POINTER(C.c_double)), T, B,
                                     self._p(dzo), self._p(dz0), self._p(dth), self._p(dW), s), self.h, "lde_adjoint")
        torch.cuda.synchronize()
        return (dz0.cpu().numpy(), None if dth is None else dth.cpu().numpy(),
                None if dW is None else dW.cpu().numpy(), self.stats(1))

    def set_option(self, key, value):
        L.check(self.lib.lde_set_option(self.h, key.encode(), float(value)), self.h, "lde_set_option")

    def step_record(self, which, B, cap=None):
        """Host copy of the last call's step sequences as the oracle's `rec` dict: t, dt [nseq, cap], n [nseq]."""
        nseq = B if self.d.batching == L.BATCH_PER_TRAJECTORY else 1
        cap = cap or 4096
        t, dt, n = np.zeros((nseq, cap)), np.zeros((nseq, cap)), np.zeros(nseq, np.int32)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_get_step_record(self.h, which, t.ctypes.data_as(C.c_void_p) if which == 0 else None,
                                             dt.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), nseq, cap, s), self.h,
                "lde_get_step_record")
        m = max(int(n.max()), 1)
        assert m <= cap, "record longer than the host copy"
        return dict(t=t[:, :m].copy(), dt=dt[:, :m].copy(), n=n)

    def stats(self, which):
        st = L.Stats()
        L.check(self.lib.lde_get_stats(self.h, which, C.byref(st), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                self.h, "lde_get_stats")
        return dict(nfe=st.nfe, naccept=st.naccept, nreject=st.nreject, nfailed=st.nfailed, max_steps=st.max_steps)


class NativeChain:
    """The dense-chain entry points of the C ABI (lde_chain_*), numpy in / numpy out."""

    def set_option(self, key, value):
        L.check(self.lib.lde_chain_set_option(self.h, key.encode(), float(value)), self.h, "lde_chain_set_option", chain=True)

    def __init__(self, sizes, acts, skips=None):
        self.lib = L.load()
        d = L.ChainDesc()
        d.abi_version = L.LDE_ABI_VERSION
        d.n_layers = len(sizes) - 1
        for i, s in enumerate(sizes):
            d.sizes[i] = s
        for i, a in enumerate(acts):
            d.activation[i] = a
            d.skip[i] = int(bool(skips[i])) if skips is not None else 0
        self.d, self.sizes = d, tuple(sizes)
        self.h = C.c_void_p()
        rc = self.lib.lde_chain_create(C.byref(d), C.byref(self.h))
        if rc != 0:
            try:
                L.check(rc, self.h if self.h else None, "lde_chain_create", chain=True)
            finally:
                self.close()
        self.nW = int(self.lib.lde_chain_num_weights(C.byref(d)))

    def close(self):
        if self.h:
            self.lib.lde_chain_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, W):
        W = np.ascontiguousarray(W, np.float32)
        L.check(self.lib.lde_chain_set_weights(self.h, W.ctypes.data_as(C.c_void_p), W.size), self.h,
                "lde_chain_set_weights", chain=True)

    def set_dtype(self, dtype):
        L.check(self.lib.lde_chain_set_dtype(self.h, {"f32": L.DTYPE_F32, "bf16": L.DTYPE_BF16}[dtype]), self.h,
                "lde_chain_set_dtype", chain=True)

    def forward(self, x):
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda")
        N = xd.shape[0]
        y = torch.full((N, self.sizes[-1]), 7.0, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_chain_forward(self.h, C.c_void_p(xd.data_ptr()), N, C.c_void_p(y.data_ptr()), s), self.h,
                "lde_chain_forward", chain=True)
        torch.cuda.synchronize()
        return y.cpu().numpy()

    def forward_save(self, x):
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda")
        N = xd.shape[0]
        y = torch.full((N, self.sizes[-1]), 7.0, device="cuda")
        self.lib.lde_chain_saved_floats.restype = C.c_int64
        saved = torch.full((int(self.lib.lde_chain_saved_floats(self.h, N)),), 7.0, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_chain_forward_save(self.h, C.c_void_p(xd.data_ptr()), N, C.c_void_p(y.data_ptr()),
                                                C.c_void_p(saved.data_ptr()), s), self.h, "lde_chain_forward_save", chain=True)
        torch.cuda.synchronize()
        return y.cpu().numpy(), saved

    def backward_saved(self, x, y, dy, saved, need_dx=True):
        dev = "cuda"
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        yd = torch.from_numpy(np.ascontiguousarray(y, np.float32)).to(dev)
        dyd = torch.from_numpy(np.ascontiguousarray(dy, np.float32)).to(dev)
        N = xd.shape[0]
        dx = torch.full_like(xd, 7.0) if need_dx else None
        dW = torch.zeros((self.nW,), device=dev)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_chain_backward_saved(self.h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), C.c_void_p(dyd.data_ptr()),
                                                  C.c_void_p(saved.data_ptr()), N,
                                                  C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(),
                                                  C.c_void_p(dW.data_ptr()), s), self.h, "lde_chain_backward_saved", chain=True)
        torch.cuda.synchronize()
        return (None if dx is None else dx.cpu().numpy()), dW.cpu().numpy()

    def backward(self, x, y, dy, need_dx=True, dW0=None):
        dev = "cuda"
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        yd = torch.from_numpy(np.ascontiguousarray(y, np.float32)).to(dev)
        dyd = torch.from_numpy(np.ascontiguousarray(dy, np.float32)).to(dev)
        N = xd.shape[0]
        dx = torch.full_like(xd, 7.0) if need_dx else None
        dW = torch.zeros((self.nW,), device=dev) if dW0 is None else torch.from_numpy(np.ascontiguousarray(dW0, np.float32)).to(dev)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_chain_backward(self.h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()),
                                            C.c_void_p(dyd.data_ptr()), N,
                                            C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(),
                                            C.c_void_p(dW.data_ptr()), s), self.h, "lde_chain_backward", chain=True)
        torch.cuda.synchronize()
        return (None if dx is None else dx.cpu().numpy()), dW.cpu().numpy()


class NativeRnn:
    """The recurrent-stack entry points of the C ABI (lde_rnn_*), numpy in / numpy out."""

    def set_option(self, key, value):
        L.check(self.lib.lde_rnn_set_option(self.h, key.encode(), float(value)), self.h, "lde_rnn_set_option", rnn=True)

    def __init__(self, cell, sizes, reverse=False):
        self.lib = L.load()
        d = L.RnnDesc()
        d.abi_version, d.cell, d.n_layers, d.reverse = L.LDE_ABI_VERSION, cell, len(sizes) - 1, int(bool(reverse))
        for i, s in enumerate(sizes):
            d.sizes[i] = s
        self.d, self.sizes = d, tuple(sizes)
        self.h = C.c_void_p()
        rc = self.lib.lde_rnn_create(C.byref(d), C.byref(self.h))
        if rc != 0:
            try:
                L.check(rc, self.h if self.h else None, "lde_rnn_create", rnn=True)
            finally:
                self.close()
        self.nW = int(self.lib.lde_rnn_num_weights(C.byref(d)))

    def close(self):
        if self.h:
            self.lib.lde_rnn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, W):
        W = np.ascontiguousarray(W, np.float32)
        L.check(self.lib.lde_rnn_set_weights(self.h, W.ctypes.data_as(C.c_void_p), W.size), self.h, "lde_rnn_set_weights", rnn=True)

    def forward(self, x):
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda")
        T, B, _ = xd.shape
        y = torch.full((B, self.sizes[-1]), 7.0, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_rnn_forward(self.h, C.c_void_p(xd.data_ptr()), T, B, C.c_void_p(y.data_ptr()), s), self.h,
                "lde_rnn_forward", rnn=True)
        torch.cuda.synchronize()
        return y.cpu().numpy()

    def backward(self, x, dy, need_dx=True, dW0=None):
        xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda")
        dyd = torch.from_numpy(np.ascontiguousarray(dy, np.float32)).to("cuda")
        T, B, _ = xd.shape
        dx = torch.full_like(xd, 7.0) if need_dx else None
        dW = torch.zeros((self.nW,), device="cuda") if dW0 is None else torch.from_numpy(np.ascontiguousarray(dW0, np.float32)).to("cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.lde_rnn_backward(self.h, C.c_void_p(xd.data_ptr()), C.c_void_p(dyd.data_ptr()), T, B,
                                          C.c_void_p(dx.data_ptr()) if dx is not None else C.c_void_p(), C.c_void_p(dW.data_ptr()), s),
                self.h, "lde_rnn_backward", rnn=True)
        torch.cuda.synchronize()
        return (None if dx is None else dx.cpu().numpy()), dW.cpu().numpy()


class NativeLoss:
    """The stateless loss entry points of the C ABI (lde_sample_* / lde_kl_* / lde_mse_*), numpy in / numpy out."""

    def __init__(self):
        self.lib = L.load()

    @staticmethod
    def _d(a, offset=0):
        """numpy → device tensor; `offset` floats into a larger buffer (to exercise unaligned operands)."""
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        buf = torch.empty(a.size + offset, device="cuda", dtype=torch.float32)
        t = buf[offset:]
        t.copy_(torch.from_numpy(a))
        return t

    @staticmethod
    def _s():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def sample_forward(self, mu, logvar, eps, offset=0):
        m, s, e = self._d(mu, offset), self._d(logvar, offset), self._d(eps, offset)
        out = torch.full_like(m, 7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_sample_forward(p(m), p(s), p(e), m.numel(), p(out), self._s()), None, "lde_sample_forward")
        return out.cpu().numpy()

    def sample_backward(self, logvar, eps, dl, offset=0):
        s, e, g = self._d(logvar, offset), self._d(eps, offset), self._d(dl, offset)
        out = torch.full_like(s, 7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_sample_backward(p(s), p(e), p(g), s.numel(), p(out), self._s()), None, "lde_sample_backward")
        return out.cpu().numpy()

    def _reduce(self, fn, a, b, scale, offset):
        a, b = self._d(a, offset), self._d(b, offset)
        out = torch.full((1,), 7.0, device="cuda")
        scratch = torch.full((L.LOSS_SCRATCH_FLOATS,), float("nan"), device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(fn(p(a), p(b), a.numel(), scale, p(out), p(scratch), self._s()), None, "loss reduction")
        return float(out.cpu()[0])

    def kl_forward(self, mu, logvar, scale, offset=0):
        return self._reduce(self.lib.lde_kl_forward, mu, logvar, scale, offset)

    def mse_forward(self, x, xhat, scale, offset=0):
        return self._reduce(self.lib.lde_mse_forward, x, xhat, scale, offset)

    def kl_backward(self, mu, logvar, scale, g, offset=0):
        m, s = self._d(mu, offset), self._d(logvar, offset)
        gd = torch.tensor([g], device="cuda", dtype=torch.float32)
        dm, ds = torch.full_like(m, 7.0), torch.full_like(m, 7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_kl_backward(p(m), p(s), m.numel(), scale, p(gd), p(dm), p(ds), self._s()), None, "lde_kl_backward")
        return dm.cpu().numpy(), ds.cpu().numpy()

    def sample_kl_forward(self, mu, logvar, eps, scale, base=None, offset=0):
        m, s, e = self._d(mu, offset), self._d(logvar, offset), self._d(eps, offset)
        l = torch.full_like(m, 7.0)
        out = torch.full((1,), 7.0, device="cuda")
        scratch = torch.full((L.LOSS_SCRATCH_FLOATS,), float("nan"), device="cuda")
        bd = torch.tensor([base], device="cuda", dtype=torch.float32) if base is not None else None
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_sample_kl_forward(p(m), p(s), p(e), m.numel(), scale, p(bd) if bd is not None else C.c_void_p(), p(l), p(out),
                                               p(scratch), self._s()), None, "lde_sample_kl_forward")
        return l.cpu().numpy(), float(out.cpu()[0])

    def sample_kl_backward(self, mu, logvar, eps, dl, g, scale, offset=0):
        m, s, e, d = self._d(mu, offset), self._d(logvar, offset), self._d(eps, offset), self._d(dl, offset)
        gd = torch.tensor([g], device="cuda", dtype=torch.float32)
        dm, ds = torch.full_like(m, 7.0), torch.full_like(m, 7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_sample_kl_backward(p(m), p(s), p(e), p(d), p(gd), scale, m.numel(), p(dm), p(ds), self._s()), None,
                "lde_sample_kl_backward")
        return dm.cpu().numpy(), ds.cpu().numpy()

    def mse_forward_add(self, x, xhat, scale, base, offset=0):
        a, b = self._d(x, offset), self._d(xhat, offset)
        out = torch.full((1,), 7.0, device="cuda")
        scratch = torch.full((L.LOSS_SCRATCH_FLOATS,), float("nan"), device="cuda")
        bd = torch.tensor([base], device="cuda", dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_mse_forward_add(p(a), p(b), a.numel(), scale, p(bd), p(out), p(scratch), self._s()), None, "lde_mse_forward_add")
        return float(out.cpu()[0])

    def mse_backward(self, x, xhat, scale, g, offset=0):
        a, b = self._d(x, offset), self._d(xhat, offset)
        gd = torch.tensor([g], device="cuda", dtype=torch.float32)
        out = torch.full_like(a, 7.0)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(self.lib.lde_mse_backward(p(a), p(b), a.numel(), scale, p(gd), p(out), self._s()), None, "lde_mse_backward")
        return out.cpu().numpy()
