"""ctypes front-end of the CPU oracle (oracle/lde_oracle.c).

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product package never does.

PARITY UNPINNED against the reference's own tests (it has none: [REF test/runtests.jl:4-6]); pinned by the
independent known-answer tests in tests/test_oracle_kat.py instead. See the header of lde_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

LDE_MAX_LAYERS = 6
RHS_PENDULUM, RHS_PENDULUM_FRICTION, RHS_MLP, RHS_PENDULUM_PLUS_MLP = 0, 1, 2, 3
SOLVER_TSIT5, SOLVER_RK4 = 0, 1
BATCH_PER_TRAJECTORY, BATCH_COUPLED, BATCH_COUPLED_GLOBAL = 0, 1, 2
SENSE_BACKSOLVE_CHECKPOINTED, SENSE_BACKSOLVE, SENSE_PARALLEL_CHECKPOINTED, SENSE_DISCRETE = 0, 1, 2, 3
ACT_RELU, ACT_TANH = 0, 1


class Desc(C.Structure):
    """Mirror of lde_problem_desc (include/lde.h)."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("rhs_kind", C.c_int32),
        ("state_dim", C.c_int32),
        ("param_dim", C.c_int32),
        ("augment_dim", C.c_int32),
        ("n_layers", C.c_int32),
        ("layer_sizes", C.c_int32 * (LDE_MAX_LAYERS + 1)),
        ("activation", C.c_int32),
        ("solver", C.c_int32),
        ("batching", C.c_int32),
        ("sensealg", C.c_int32),
        ("adaptive", C.c_int32),
        ("maxiters", C.c_int64),
        ("dt", C.c_double),
        ("abstol", C.c_double),
        ("reltol", C.c_double),
        ("dtmin", C.c_double),
        ("qmin", C.c_double),
        ("qmax", C.c_double),
        ("gamma", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
    ]


def make_desc(rhs_kind=RHS_PENDULUM, state_dim=2, param_dim=1, augment_dim=0, layers=(), activation=ACT_RELU,
              solver=SOLVER_TSIT5, batching=BATCH_PER_TRAJECTORY, sensealg=SENSE_BACKSOLVE_CHECKPOINTED,
              adaptive=True, dt=0.0, abstol=1e-6, reltol=1e-3, maxiters=100000, dtmin=0.0,
              qmin=0.2, qmax=10.0, gamma=0.9, beta1=7.0 / 50.0, beta2=2.0 / 25.0) -> Desc:
    d = Desc()
    d.abi_version = 1
    d.rhs_kind, d.state_dim, d.param_dim, d.augment_dim = rhs_kind, state_dim, param_dim, augment_dim
    layers = tuple(layers)
    d.n_layers = max(len(layers) - 1, 0)
    for i, s in enumerate(layers):
        d.layer_sizes[i] = s
    d.activation, d.solver, d.batching, d.sensealg = activation, solver, batching, sensealg
    d.adaptive, d.maxiters, d.dt = int(adaptive), maxiters, dt
    d.abstol, d.reltol, d.dtmin = abstol, reltol, dtmin
    d.qmin, d.qmax, d.gamma, d.beta1, d.beta2 = qmin, qmax, gamma, beta1, beta2
    return d


def build(native: bool = False) -> None:
    """Compile the oracle (gcc). Building the checker is not using it."""
    targets = ["all"] + (["native"] if native else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


class Oracle:
    """One precision of the oracle: Oracle('f32') or Oracle('f64')."""

    def __init__(self, prec: str = "f32", native: bool = False):
        assert prec in ("f32", "f64")
        name = f"liblde_oracle_{prec}{'_native' if native else ''}.so"
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build(native=native)
        self.lib = C.CDLL(path)
        self.dtype = np.float32 if prec == "f32" else np.float64
        assert self.lib.oracle_real_size() == np.dtype(self.dtype).itemsize
        self.lib.oracle_num_weights.restype = C.c_int64

    def _p(self, a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def tableau(self):
        out = np.zeros(77)
        self.lib.oracle_tsit5_tableau(self._p(out))
        c, a, bt, r1, r = out[:7], out[7:49].reshape(7, 6), out[49:56], out[56:59], out[59:].reshape(6, 3)
        return c, a, bt, r1, r

    def num_weights(self, d: Desc) -> int:
        return int(self.lib.oracle_num_weights(C.byref(d)))

    def set_sum_hook(self, fn, nscale: float = 1.0):
        """LDE_BATCH_COUPLED_GLOBAL stand-in: `fn(vals: float64 array)` replaces the entries by their sums over all ranks (in place);
        nscale = global batch / this rank's batch. `fn=None` clears it. (oracle_set_sum_hook, lde_oracle.c)"""
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
        if fn is None:
            self._hook = HOOK(0)
        else:
            def _cb(_user, vals, n):
                fn(np.ctypeslib.as_array(vals, shape=(n,)))
                return 0
            self._hook = HOOK(_cb)
        self.lib.oracle_set_sum_hook.argtypes = [HOOK, C.c_void_p, C.c_double]
        self.lib.oracle_set_sum_hook(self._hook, None, float(nscale))

    def forward(self, d: Desc, z0, theta, ts, W=None, nthreads=0, max_trace=4096):
        """z0 [D,B] (Fortran order semantic: pass arrays shaped (B,D) C-order == [D×B] column-major).

        To keep the layout explicit, all arrays here are numpy arrays whose *memory* is the reference's
        column-major layout: z0.shape == (B, D), theta.shape == (B, P), z_out.shape == (T, B, D').
        """
        dt = self.dtype
        z0 = np.ascontiguousarray(z0, dtype=dt)
        B, D = z0.shape
        assert D == d.state_dim
        Dp = D + d.augment_dim
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        T = ts.shape[0]
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        z_out = np.zeros((T, B, Dp), dtype=dt)
        ret = np.zeros(B, dtype=np.int32)
        stats = np.zeros(5, dtype=np.int64)
        trace = np.zeros(max_trace, dtype=np.float64)
        ntr = C.c_int(0)
        rc = self.lib.oracle_forward(C.byref(d), self._p(W), self._p(z0), self._p(theta), self._p(ts), T, B,
                                     self._p(z_out), self._p(ret), self._p(stats), self._p(trace), C.byref(ntr),
                                     max_trace, nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_forward failed: {rc}")
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=int(stats[2]), nfailed=int(stats[3]),
                    max_steps=int(stats[4]), dt_trace=trace[: min(ntr.value, max_trace)].copy())
        return z_out, ret, info

    def adjoint(self, d: Desc, z_out, theta, ts, dz_out, W=None, nthreads=0):
        dt = self.dtype
        z_out = np.ascontiguousarray(z_out, dtype=dt)
        dz_out = np.ascontiguousarray(dz_out, dtype=dt)
        T, B, Dp = z_out.shape
        D, P = d.state_dim, d.param_dim
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        dz0 = np.zeros((B, D), dtype=dt)
        dth = np.zeros((B, max(P, 1)), dtype=dt)
        nW = self.num_weights(d) if d.rhs_kind in (RHS_MLP, RHS_PENDULUM_PLUS_MLP) else 0
        dW = np.zeros(max(nW, 1), dtype=dt)
        stats = np.zeros(5, dtype=np.int64)
        rc = self.lib.oracle_adjoint(C.byref(d), self._p(W), self._p(z_out), self._p(theta), self._p(ts), T, B,
                                     self._p(dz_out), self._p(dz0), self._p(dth), self._p(dW), self._p(stats),
                                     nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_adjoint failed: {rc}")
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=int(stats[2]), nfailed=int(stats[3]),
                    max_steps=int(stats[4]))
        return dz0, (dth[:, :P] if P else None), (dW[:nW] if nW else None), info

    # ---- the same solves with their accepted steps recorded or prescribed (lde_oracle.c: steprec) -----------------------------------
    @staticmethod
    def _nseq(d: Desc, B: int) -> int:
        return B if d.batching == BATCH_PER_TRAJECTORY else 1

    def forward_steps(self, d: Desc, z0, theta, ts, W=None, rec=None, cap=4096, nthreads=0):
        """rec=None: an adaptive solve, its accepted steps returned as rec = dict(t [nseq,cap], dt [nseq,cap], n [nseq]).
        rec given: exactly these steps are taken (no error control)."""
        dt = self.dtype
        z0 = np.ascontiguousarray(z0, dtype=dt)
        B, D = z0.shape
        Dp = D + d.augment_dim
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        T = ts.shape[0]
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        nseq = self._nseq(d, B)
        presc = rec is not None
        if presc:
            rt = np.ascontiguousarray(rec["t"], np.float64).reshape(nseq, -1)
            rdt = np.ascontiguousarray(rec["dt"], np.float64).reshape(nseq, -1)
            rn = np.ascontiguousarray(rec["n"], np.int32).reshape(nseq)
            cap = rt.shape[1]
        else:
            rt, rdt, rn = np.zeros((nseq, cap)), np.zeros((nseq, cap)), np.zeros(nseq, np.int32)
        z_out = np.zeros((T, B, Dp), dtype=dt)
        ret = np.zeros(B, dtype=np.int32)
        stats = np.zeros(5, dtype=np.int64)
        rc = self.lib.oracle_forward_steps(C.byref(d), self._p(W), self._p(z0), self._p(theta), self._p(ts), T, B, self._p(z_out),
                                           self._p(ret), self._p(stats), self._p(rt), self._p(rdt), self._p(rn), cap, int(presc), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_forward_steps failed: {rc}")
        assert presc or int(rn.max()) <= cap, "step record overflow: raise cap"
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=int(stats[2]), nfailed=int(stats[3]), max_steps=int(stats[4]))
        return z_out, ret, dict(t=rt, dt=rdt, n=rn), info

    def forward_dual(self, d: Desc, z0, theta, ts, dz_out=None, dual_norm=True, rec=None, cap=4096, nthreads=0, keep_jac=True):
        """ForwardDiffSensitivity as the reference executes it (oracle_forward_dual): the solve on dual numbers, partials with respect to
        (ẑ₀, θ̂). dual_norm=True: the step control's norms see the partials (what the reference's TRAINING solve does); False: the
        primal step sequence. rec given: exactly these steps. Returns ẑ [T,B,2], J [T,B,2,3] (or None), (dz0 [B,2], dθ [B,1]) for dz_out, ret, rec, info."""
        dt = self.dtype
        z0 = np.ascontiguousarray(z0, dtype=dt)
        B, D = z0.shape
        assert D == 2 and d.param_dim == 1
        theta = np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        T = ts.shape[0]
        presc = rec is not None
        if presc:
            rt = np.ascontiguousarray(rec["t"], np.float64).reshape(B, -1)
            rdt = np.ascontiguousarray(rec["dt"], np.float64).reshape(B, -1)
            rn = np.ascontiguousarray(rec["n"], np.int32).reshape(B)
            cap = rt.shape[1]
        else:
            rt, rdt, rn = np.zeros((B, cap)), np.zeros((B, cap)), np.zeros(B, np.int32)
        z_out = np.zeros((T, B, 2), dtype=dt)
        J = np.zeros((T, B, 2, 3), dtype=dt) if keep_jac else None
        dz = None if dz_out is None else np.ascontiguousarray(dz_out, dtype=dt)
        dz0 = np.zeros((B, 2), dtype=dt)
        dth = np.zeros((B, 1), dtype=dt)
        ret = np.zeros(B, dtype=np.int32)
        stats = np.zeros(5, dtype=np.int64)
        rc = self.lib.oracle_forward_dual(C.byref(d), self._p(z0), self._p(theta), self._p(ts), T, B, int(bool(dual_norm)), self._p(z_out),
                                          self._p(J), self._p(dz), self._p(dz0), self._p(dth), self._p(ret), self._p(stats), self._p(rt),
                                          self._p(rdt), self._p(rn), cap, int(presc), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_forward_dual failed: {rc}")
        assert presc or int(rn.max()) <= cap, "step record overflow: raise cap"
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=int(stats[2]), nfailed=int(stats[3]), max_steps=int(stats[4]))
        return z_out, J, (dz0, dth), ret, dict(t=rt, dt=rdt, n=rn), info

    def adjoint_steps(self, d: Desc, z_out, theta, ts, dz_out, W=None, rec=None, cap=8192, nthreads=0, margins=False):
        """The continuous adjoint with the reverse-time solve's accepted step magnitudes recorded (rec=None) or prescribed.
        margins=True: info["margins"][b] = how close trajectory b's reverse-time solve came to a relu kink."""
        dt = self.dtype
        z_out = np.ascontiguousarray(z_out, dtype=dt)
        dz_out = np.ascontiguousarray(dz_out, dtype=dt)
        T, B, Dp = z_out.shape
        D, P = d.state_dim, d.param_dim
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        nseq = self._nseq(d, B)
        presc = rec is not None
        if presc:
            rdt = np.ascontiguousarray(rec["dt"], np.float64).reshape(nseq, -1)
            rn = np.ascontiguousarray(rec["n"], np.int32).reshape(nseq)
            cap = rdt.shape[1]
        else:
            rdt, rn = np.zeros((nseq, cap)), np.zeros(nseq, np.int32)
        dz0 = np.zeros((B, D), dtype=dt)
        dth = np.zeros((B, max(P, 1)), dtype=dt)
        nW = self.num_weights(d) if d.rhs_kind in (RHS_MLP, RHS_PENDULUM_PLUS_MLP) else 0
        dW = np.zeros(max(nW, 1), dtype=dt)
        stats = np.zeros(5, dtype=np.int64)
        mg = np.ones(B) if margins else None
        rc = self.lib.oracle_adjoint_steps_margins(C.byref(d), self._p(W), self._p(z_out), self._p(theta), self._p(ts), T, B,
                                                   self._p(dz_out), self._p(dz0), self._p(dth), self._p(dW), self._p(stats), self._p(rdt),
                                                   self._p(rn), cap, int(presc), self._p(mg), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_adjoint_steps failed: {rc}")
        assert presc or int(rn.max()) <= cap, "step record overflow: raise cap"
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=int(stats[2]), nfailed=int(stats[3]), max_steps=int(stats[4]),
                    margins=mg)
        return dz0, (dth[:, :P] if P else None), (dW[:nW] if nW else None), dict(dt=rdt, n=rn), info

    def adjoint_discrete(self, d: Desc, z_out, theta, ts, dz_out, rec, W=None, nthreads=0, margins=False):
        """LDE_SENSE_DISCRETE: reverse-mode derivative of the discrete solve on the recorded steps `rec` (of forward_steps, or of a kernel).
        margins=True: info["margins"][b] = how close trajectory b's solve came to a relu kink (min |pre-activation| / Σ|terms|)."""
        dt = self.dtype
        z_out = np.ascontiguousarray(z_out, dtype=dt)
        dz_out = np.ascontiguousarray(dz_out, dtype=dt)
        T, B, Dp = z_out.shape
        D, P = d.state_dim, d.param_dim
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        nseq = self._nseq(d, B)
        rt = np.ascontiguousarray(rec["t"], np.float64).reshape(nseq, -1)
        rdt = np.ascontiguousarray(rec["dt"], np.float64).reshape(nseq, -1)
        rn = np.ascontiguousarray(rec["n"], np.int32).reshape(nseq)
        dz0 = np.zeros((B, D), dtype=dt)
        dth = np.zeros((B, max(P, 1)), dtype=dt)
        nW = self.num_weights(d) if d.rhs_kind in (RHS_MLP, RHS_PENDULUM_PLUS_MLP) else 0
        dW = np.zeros(max(nW, 1), dtype=dt)
        stats = np.zeros(5, dtype=np.int64)
        mg = np.ones(B) if margins else None
        rc = self.lib.oracle_adjoint_discrete_margins(C.byref(d), self._p(W), self._p(z_out), self._p(theta), self._p(ts), T, B,
                                                      self._p(dz_out), self._p(rt), self._p(rdt), self._p(rn), rt.shape[1], self._p(dz0),
                                                      self._p(dth), self._p(dW), self._p(stats), self._p(mg), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_adjoint_discrete failed: {rc}")
        info = dict(nfe=int(stats[0]), naccept=int(stats[1]), nreject=0, nfailed=int(stats[3]), max_steps=int(stats[4]), margins=mg)
        return dz0, (dth[:, :P] if P else None), (dW[:nW] if nW else None), info

    def rhs(self, d: Desc, z, theta, W=None):
        dt = self.dtype
        z = np.ascontiguousarray(z, dtype=dt)
        out = np.zeros(d.state_dim + d.augment_dim, dtype=dt)
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        self.lib.oracle_rhs(C.byref(d), self._p(W), self._p(z), self._p(theta), self._p(out))
        return out

    def rhs_vjp(self, d: Desc, z, theta, lam, W=None):
        dt = self.dtype
        Dp = d.state_dim + d.augment_dim
        z = np.ascontiguousarray(z, dtype=dt)
        lam = np.ascontiguousarray(lam, dtype=dt)
        theta = None if theta is None else np.ascontiguousarray(theta, dtype=dt)
        W = None if W is None else np.ascontiguousarray(W, dtype=dt)
        nW = self.num_weights(d) if W is not None else 0
        f, vz, vth, dW = np.zeros(Dp, dt), np.zeros(Dp, dt), np.zeros(max(d.param_dim, 1), dt), np.zeros(max(nW, 1), dt)
        self.lib.oracle_rhs_vjp(C.byref(d), self._p(W), self._p(z), self._p(theta), self._p(lam), self._p(f),
                                self._p(vz), self._p(vth), self._p(dW))
        return f, vz, vth[: d.param_dim], dW[:nW]


# synthetic-input generators live in the product package (pure numpy); re-exported for the tests
import sys as _sys
_sys.path.insert(0, os.path.dirname(_HERE))
from latentdiffeq_amd.synthetic import cotangent, mlp_weights, pendulum_inputs, time_grid  # noqa: E402,F401


# ---- dense chains either side of the solve (oracle/lde_chain_oracle.c) ------------------------------------------
LDE_CHAIN_MAX_LAYERS = 6
CACT_IDENTITY, CACT_RELU, CACT_TANH, CACT_SIGMOID, CACT_SOFTPLUS = 0, 1, 2, 3, 4


class ChainDesc(C.Structure):
    """Mirror of lde_chain_desc (include/lde.h)."""

    _fields_ = [("abi_version", C.c_int32), ("n_layers", C.c_int32), ("sizes", C.c_int32 * (LDE_CHAIN_MAX_LAYERS + 1)),
                ("activation", C.c_int32 * LDE_CHAIN_MAX_LAYERS), ("skip", C.c_int32 * LDE_CHAIN_MAX_LAYERS)]


def make_chain_desc(sizes, activations, skips=None) -> ChainDesc:
    d = ChainDesc()
    d.abi_version = 1
    d.n_layers = len(sizes) - 1
    assert len(activations) == d.n_layers
    for i, s in enumerate(sizes):
        d.sizes[i] = s
    for i, a in enumerate(activations):
        d.activation[i] = a
        d.skip[i] = int(bool(skips[i])) if skips is not None else 0
    return d


def _chain_forward(self, d: ChainDesc, W, x, nthreads=0):
    """x.shape == (N, in) C-order == [in × N] column-major; returns y of shape (N, out)."""
    dt = self.dtype
    x = np.ascontiguousarray(x, dtype=dt)
    W = np.ascontiguousarray(W, dtype=dt)
    N = x.shape[0]
    assert x.shape[1] == d.sizes[0]
    y = np.zeros((N, d.sizes[d.n_layers]), dtype=dt)
    rc = self.lib.oracle_chain_forward(C.byref(d), self._p(W), self._p(x), C.c_int64(N), self._p(y), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle_chain_forward failed: {rc}")
    return y


def _chain_backward(self, d: ChainDesc, W, x, dy, nthreads=0, need_dx=True):
    dt = self.dtype
    x = np.ascontiguousarray(x, dtype=dt)
    dy = np.ascontiguousarray(dy, dtype=dt)
    W = np.ascontiguousarray(W, dtype=dt)
    N = x.shape[0]
    dx = np.zeros_like(x) if need_dx else None
    dW = np.zeros_like(W)
    rc = self.lib.oracle_chain_backward(C.byref(d), self._p(W), self._p(x), self._p(dy), C.c_int64(N), self._p(dx),
                                        self._p(dW), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle_chain_backward failed: {rc}")
    return dx, dW


def _chain_num_weights(self, d: ChainDesc) -> int:
    self.lib.oracle_chain_num_weights.restype = C.c_int64
    return int(self.lib.oracle_chain_num_weights(C.byref(d)))


Oracle.chain_forward = _chain_forward
Oracle.chain_backward = _chain_backward
Oracle.chain_num_weights = _chain_num_weights


# ---- recurrent pattern extractor (oracle/lde_rnn_oracle.c) -----------------------------------------------------
LDE_RNN_MAX_LAYERS = 4
CELL_RNN_RELU, CELL_RNN_TANH, CELL_LSTM = 0, 1, 2


class RnnDesc(C.Structure):
    """Mirror of lde_rnn_desc (include/lde.h)."""

    _fields_ = [("abi_version", C.c_int32), ("cell", C.c_int32), ("n_layers", C.c_int32),
                ("sizes", C.c_int32 * (LDE_RNN_MAX_LAYERS + 1)), ("reverse", C.c_int32)]


def make_rnn_desc(cell, sizes, reverse=False) -> RnnDesc:
    d = RnnDesc()
    d.abi_version, d.cell, d.n_layers, d.reverse = 1, cell, len(sizes) - 1, int(bool(reverse))
    for i, s in enumerate(sizes):
        d.sizes[i] = s
    return d


def rnn_weights(cell, sizes, seed=5, dtype=np.float32):
    """Flat Flux.destructure-order weights of a stack: per cell vec(Wi), vec(Wh), b, state0 — U(±1/√fan_in), non-zero
    bias and initial state so that every term is exercised."""
    rng = np.random.default_rng(seed)
    G = 4 if cell == CELL_LSTM else 1
    parts = []
    for i in range(len(sizes) - 1):
        n_in, h = sizes[i], sizes[i + 1]
        parts.append(rng.uniform(-1, 1, (G * h, n_in)).flatten(order="F") / np.sqrt(n_in))
        parts.append(rng.uniform(-1, 1, (G * h, h)).flatten(order="F") / np.sqrt(h))
        parts.append(rng.uniform(-0.3, 0.3, G * h))
        parts.append(rng.uniform(-0.5, 0.5, h * (2 if cell == CELL_LSTM else 1)))
    return np.concatenate(parts).astype(dtype)


def _rnn_num_weights(self, d: RnnDesc) -> int:
    self.lib.oracle_rnn_num_weights.restype = C.c_int64
    return int(self.lib.oracle_rnn_num_weights(C.byref(d)))


def _rnn_forward(self, d: RnnDesc, W, x, nthreads=0):
    """x.shape == (T, B, in) C-order == [in × B × T] column-major; returns y (B, h_last)."""
    dt = self.dtype
    x = np.ascontiguousarray(x, dtype=dt)
    W = np.ascontiguousarray(W, dtype=dt)
    T, B, n_in = x.shape
    assert n_in == d.sizes[0]
    y = np.zeros((B, d.sizes[d.n_layers]), dtype=dt)
    rc = self.lib.oracle_rnn_forward(C.byref(d), self._p(W), self._p(x), T, B, self._p(y), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle_rnn_forward failed: {rc}")
    return y


def _rnn_backward(self, d: RnnDesc, W, x, dy, nthreads=0, need_dx=True):
    dt = self.dtype
    x = np.ascontiguousarray(x, dtype=dt)
    dy = np.ascontiguousarray(dy, dtype=dt)
    W = np.ascontiguousarray(W, dtype=dt)
    T, B, _ = x.shape
    dx = np.zeros_like(x) if need_dx else None
    dW = np.zeros_like(W)
    rc = self.lib.oracle_rnn_backward(C.byref(d), self._p(W), self._p(x), self._p(dy), T, B, self._p(dx), self._p(dW), nthreads)
    if rc != 0:
        raise RuntimeError(f"oracle_rnn_backward failed: {rc}")
    return dx, dW


Oracle.rnn_num_weights = _rnn_num_weights
Oracle.rnn_forward = _rnn_forward
Oracle.rnn_backward = _rnn_backward


# ---- the variational sample and the loss terms (lde_loss_oracle.c) ---------------------------------------------------
def _sample_forward(self, mu, logvar, eps):
    dt = self.dtype
    mu, logvar, eps = (np.ascontiguousarray(a, dtype=dt) for a in (mu, logvar, eps))
    out = np.empty_like(mu)
    self.lib.oracle_sample_forward(self._p(mu), self._p(logvar), self._p(eps), C.c_int64(mu.size), self._p(out))
    return out

def _sample_backward(self, logvar, eps, dl):
    dt = self.dtype
    logvar, eps, dl = (np.ascontiguousarray(a, dtype=dt) for a in (logvar, eps, dl))
    dmu, dlv = np.empty_like(dl), np.empty_like(dl)
    self.lib.oracle_sample_backward(self._p(logvar), self._p(eps), self._p(dl), C.c_int64(dl.size), self._p(dmu), self._p(dlv))
    return dmu, dlv

def _kl_forward(self, mu, logvar, scale):
    dt = self.dtype
    mu, logvar = np.ascontiguousarray(mu, dtype=dt), np.ascontiguousarray(logvar, dtype=dt)
    self.lib.oracle_kl_forward.restype = C.c_double
    return float(self.lib.oracle_kl_forward(self._p(mu), self._p(logvar), C.c_int64(mu.size), C.c_double(scale)))

def _kl_backward(self, mu, logvar, scale, g=1.0):
    dt = self.dtype
    mu, logvar = np.ascontiguousarray(mu, dtype=dt), np.ascontiguousarray(logvar, dtype=dt)
    dmu, dlv = np.empty_like(mu), np.empty_like(mu)
    self.lib.oracle_kl_backward(self._p(mu), self._p(logvar), C.c_int64(mu.size), C.c_double(scale), C.c_double(g),
                                self._p(dmu), self._p(dlv))
    return dmu, dlv

def _mse_forward(self, x, xhat, scale):
    dt = self.dtype
    x, xhat = np.ascontiguousarray(x, dtype=dt), np.ascontiguousarray(xhat, dtype=dt)
    self.lib.oracle_mse_forward.restype = C.c_double
    return float(self.lib.oracle_mse_forward(self._p(x), self._p(xhat), C.c_int64(x.size), C.c_double(scale)))

def _mse_backward(self, x, xhat, scale, g=1.0):
    dt = self.dtype
    x, xhat = np.ascontiguousarray(x, dtype=dt), np.ascontiguousarray(xhat, dtype=dt)
    dxh = np.empty_like(xhat)
    self.lib.oracle_mse_backward(self._p(x), self._p(xhat), C.c_int64(x.size), C.c_double(scale), C.c_double(g), self._p(dxh))
    return dxh


Oracle.sample_forward = _sample_forward
Oracle.sample_backward = _sample_backward
Oracle.kl_forward = _kl_forward
Oracle.kl_backward = _kl_backward
Oracle.mse_forward = _mse_forward
Oracle.mse_backward = _mse_backward
