"""CPU-side checks of the drop-in boundary: liblde.so loads without a GPU and exports every symbol that
include/lde.h declares; struct layouts agree between the header, the product binding and the oracle binding;
host-side validation behaves like the reference interface (bad arguments → error code, never an abort)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "lde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lde_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import latentdiffeq_amd as la
    from latentdiffeq_amd import _lib
    la.build_lib()
    lib = _lib.load()
    decl = _declared_functions()
    assert len(decl) >= 12
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/lde.h but not exported by liblde.so"
    assert sorted(_lib.EXPORTS) == decl, "the ctypes binding must bind exactly the declared entry points"
    assert lib.lde_abi_version() == 1
    info = lib.lde_build_info().decode()          # the compiler that built it and the register check's verdict travel in the binary
    assert "hidden AGPR ranges respected" in info and ("clang" in info.lower() or "hip" in info.lower()), info


def test_desc_defaults_match_ordinarydiffeq_defaults_and_bindings_agree():
    from latentdiffeq_amd import _lib
    from oracle import oracle as O
    lib = _lib.load()
    d = _lib.ProblemDesc()
    assert lib.lde_problem_desc_default(C.byref(d)) == 0
    # Pendulum() carries Tsit5 and no kwargs ⇒ OrdinaryDiffEq defaults  [REF pendulum.jl:11]
    assert (d.abstol, d.reltol, d.maxiters, d.adaptive) == (1e-6, 1e-3, 100000, 1)
    assert (d.qmin, d.qmax, d.gamma) == (0.2, 10.0, 0.9) and abs(d.beta1 - 0.14) < 1e-15 and abs(d.beta2 - 0.08) < 1e-15
    assert (d.rhs_kind, d.state_dim, d.param_dim, d.solver, d.batching) == (0, 2, 1, 0, 0)
    # Pendulum() carries ForwardDiffSensitivity() [REF pendulum.jl:8-11], splatted into solve() [REF src/models/GOKU.jl:107, :121]:
    # the exact derivative of the discrete solve = LDE_SENSE_DISCRETE
    assert d.sensealg == _lib.SENSE_DISCRETE == 3
    od = O.make_desc(sensealg=O.SENSE_DISCRETE)
    assert C.sizeof(od) == C.sizeof(d)
    assert bytes(od) == bytes(d), "oracle and product describe the default problem with identical bytes"
    # header struct size: 18 int32 (incl. layer_sizes[7]) + pad, 1 int64, 9 doubles
    assert C.sizeof(d) == 4 * 18 + 8 + 9 * 8


def test_num_weights_follows_destructure_order():
    from latentdiffeq_amd import _lib
    lib = _lib.load()
    d = _lib.ProblemDesc()
    lib.lde_problem_desc_default(C.byref(d))
    assert lib.lde_num_weights(C.byref(d)) == 0
    d.rhs_kind, d.state_dim, d.param_dim, d.n_layers = _lib.RHS_MLP, 16, 0, 3
    for i, s in enumerate((16, 200, 200, 16)):
        d.layer_sizes[i] = s
    assert lib.lde_num_weights(C.byref(d)) == 46816      # the reference default NODE  [REF nODE.jl:11-14]
    for i, s in enumerate((32, 128, 128, 32)):
        d.layer_sizes[i] = s
    assert lib.lde_num_weights(C.byref(d)) == 24864      # BASELINE config 4


def test_null_and_invalid_arguments_return_error_codes():
    from latentdiffeq_amd import _lib
    lib = _lib.load()
    assert lib.lde_problem_desc_default(None) == -1
    h = C.c_void_p()
    assert lib.lde_create(None, C.byref(h)) == -1
    d = _lib.ProblemDesc()
    lib.lde_problem_desc_default(C.byref(d))
    d.abi_version = 99
    assert lib.lde_create(C.byref(d), C.byref(h)) == -1 and not h.value
    lib.lde_problem_desc_default(C.byref(d))
    d.solver, d.adaptive = _lib.SOLVER_RK4, 1
    assert lib.lde_create(C.byref(d), C.byref(h)) == -2            # adaptive RK4: LDE_ERR_UNSUPPORTED
    lib.lde_problem_desc_default(C.byref(d))
    d.state_dim = 3
    assert lib.lde_create(C.byref(d), C.byref(h)) == -1            # pendulum is 2-state
    assert lib.lde_forward(None, None, None, None, 0, 0, None, None, None) == -1
    assert lib.lde_last_error(None) == b"NULL handle"
    lib.lde_destroy(None)                                          # no-op


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    """No CPU fallback anywhere in the product path."""
    import torch
    from latentdiffeq_amd import _lib
    import latentdiffeq_amd as la
    if not torch.cuda.is_available():
        lib = _lib.load()
        d = _lib.ProblemDesc()
        lib.lde_problem_desc_default(C.byref(d))
        h = C.c_void_p()
        assert lib.lde_create(C.byref(d), C.byref(h)) == -3       # LDE_ERR_NO_DEVICE
        dec = la.Decoder(la.GOKU_basic(), (None, la.Pendulum(), None))
        with pytest.raises(_lib.LdeError):
            la.diffeq_layer(dec, (torch.zeros(2, 4), torch.ones(1, 4)), np.arange(5) * 0.05)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblde.so")
    with pytest.raises(_lib.LdeError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "latentdiffeq.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "lde_oracle" not in txt, f


def test_reference_shaped_host_api_surface():
    import latentdiffeq_amd as la
    p = la.Pendulum()
    # the struct fields the reference reads  [REF src/models/GOKU.jl:105-108, :207-208]
    assert hasattr(p, "prob") and hasattr(p, "solver") and hasattr(p, "sensealg") and hasattr(p, "kwargs")
    assert len(p.prob.u0) == 2 and len(p.prob.p) == 1
    assert isinstance(la.Pendulum(sensalg=la.BacksolveAdjoint()).sensealg, la.BacksolveAdjoint)  # `sensalg` spelling [REF pendulum.jl:11]
    # the default IS the reference's: ForwardDiffSensitivity() [REF pendulum.jl:8-11] = the exact derivative of the discrete solve
    from latentdiffeq_amd import _lib as _l
    assert isinstance(p.sensealg, la.ForwardDiffSensitivity) and p.sensealg.code == _l.SENSE_DISCRETE
    assert la.Pendulum_friction().sensealg.code == _l.SENSE_DISCRETE and la.DiscreteSensitivity().code == _l.SENSE_DISCRETE
    assert la.ParallelAdjoint().code == _l.SENSE_PARALLEL_CHECKPOINTED == la.ForwardDiffSensitivity(exact=False).code   # still selectable
    assert la.NODE(4, hidden_dim=8).sensealg.code == _l.SENSE_BACKSOLVE_CHECKPOINTED      # InterpolatingAdjoint [REF LatentODE.jl:67-70]
    n = la.NODE(4, hidden_dim=8, augment_dim=2)
    # [REF nODE.jl:3-32], [REF src/models/LatentODE.jl:62-66, :105-106]
    for f in ("dudt", "solver", "neural_model", "latent_dim_in", "latent_dim_out", "augment_dim", "kwargs"):
        assert hasattr(n, f)
    assert n.latent_dim_out == 6 and n.layer_sizes == [6, 8, 8, 6]
    w = n.flat_weights()
    assert w.numel() == 6 * 8 + 8 + 8 * 8 + 8 + 8 * 6 + 6
    lin = n.dudt[0]
    assert np.allclose(w[:48].detach().numpy().reshape(6, 8).T, lin.weight.detach().numpy())  # vec(W) column-major
    # Flux's default Dense init (no `init=` in the reference's NODE [REF nODE.jl:12-14]): glorot_uniform weights, zero biases
    ref = la.NODE(16)                                  # the reference default: 16 → 200 → 200 → 16
    assert ref.layer_sizes == [16, 200, 200, 16] and ref.flat_weights().numel() == 46816
    for m in ref.dudt:
        if hasattr(m, "weight"):
            out_f, in_f = m.weight.shape
            bound = (6.0 / (in_f + out_f)) ** 0.5
            wv = m.weight.detach().numpy()
            assert np.abs(wv).max() <= bound and np.abs(wv).max() >= 0.97 * bound        # U(±bound): the extremes are reached
            assert abs(wv.std() - bound / 3 ** 0.5) <= 0.05 * bound                     # std of U(±a) = a/√3
            assert not m.bias.detach().numpy().any()
    assert issubclass(la.GOKU_basic, la.GOKU) and issubclass(la.GOKU, la.LatentDE) and issubclass(la.LatentODE, la.LatentDE)
    assert la.transform_after_diffeq("x", p) == "x"   # identity by default [REF GOKU.jl:136]

    class Kuramoto(la.Pendulum):
        def transform_after_diffeq(self, x):
            return ("sin", x)
    assert la.transform_after_diffeq("x", Kuramoto()) == ("sin", "x")
    with pytest.raises(TypeError):
        la.diffeq_layer(la.Decoder(object(), (None, p, None)), None, [0.0])


def test_dense_chain_and_recurrent_mirrors_fail_loudly_on_cpu():
    """Rows f-1 / f-2: same rule — CPU tensors (or no GPU) raise, nothing is computed on the host."""
    import torch
    from latentdiffeq_amd import _lib
    from latentdiffeq_amd.chain import Chain, Dense, SkipConnection
    from latentdiffeq_amd.recurrent import LSTM, RNN, Recurrent
    ch = Chain(Dense(4, 8, "relu"), SkipConnection(Dense(8, 8, "tanh")), Dense(8, 3, "sigmoid"))
    assert ch.sizes == [4, 8, 8, 3] and ch.skips == [0, 1, 0] and ch.num_weights == 4 * 8 + 8 + 8 * 8 + 8 + 8 * 3 + 3
    assert ch.flat_weights().numel() == ch.num_weights
    rec = Recurrent(LSTM(6, 5), LSTM(5, 5), reverse=True)
    assert rec.num_weights == 2 * 0 + (20 * 6 + 20 * 5 + 20 + 10) + (20 * 5 + 20 * 5 + 20 + 10)
    with pytest.raises(TypeError):
        Recurrent(RNN(4, 4), LSTM(4, 4))
    with pytest.raises(ValueError):
        SkipConnection(Dense(4, 5))
    if not torch.cuda.is_available():
        with pytest.raises(_lib.LdeError):
            ch(torch.zeros(4, 2))
        with pytest.raises(_lib.LdeError):
            rec(torch.zeros(6, 2, 3))
        lib = _lib.load()
        d = _lib.ChainDesc()
        d.abi_version, d.n_layers = 1, 1
        d.sizes[0], d.sizes[1] = 3, 2
        h = C.c_void_p()
        assert lib.lde_chain_create(C.byref(d), C.byref(h)) == -3 and not h.value      # LDE_ERR_NO_DEVICE
        r = _lib.RnnDesc()
        r.abi_version, r.cell, r.n_layers = 1, _lib.CELL_LSTM, 1
        r.sizes[0], r.sizes[1] = 3, 2
        assert lib.lde_rnn_create(C.byref(r), C.byref(h)) == -3 and not h.value
        assert lib.lde_chain_num_weights(C.byref(d)) == 8 and lib.lde_rnn_num_weights(C.byref(r)) == 8 * 3 + 8 * 2 + 8 + 4


def _julia_struct_fields(name):
    """(field, julia type) list of `mutable struct <name>` in INTEGRATION.md, in declaration order."""
    src = open(os.path.join(ROOT, "julia", "LdeNative.jl")).read()      # the loadable stub (INTEGRATION.md §2 walks through it)
    m = re.search(r"mutable struct " + name + r"\b(.*?)\nend\b", src, flags=re.S)
    assert m, name
    body = re.sub(r"#.*", "", m.group(1))
    return re.findall(r"(\w+)::(Int32|Int64|Float64|NTuple\{\d+,Int32\})", body)


def test_documented_julia_struct_layouts():
    """The Julia stub (julia/LdeNative.jl) cannot run here; its struct mirrors must at least be field-for-field right:
    same order, sizes and C offsets (natural alignment — what Julia uses for an isbits-field struct passed by Ref)."""
    from latentdiffeq_amd import _lib
    for jname, cstruct in (("LdeDesc", _lib.ProblemDesc), ("LdeChainDesc", _lib.ChainDesc), ("LdeRnnDesc", _lib.RnnDesc)):
        fields = _julia_struct_fields(jname)
        assert [f for f, _ in fields] == [f for f, _ in cstruct._fields_], jname
        off = 0
        for fname, jt in fields:
            if jt.startswith("NTuple"):
                size, align = 4 * int(re.search(r"\{(\d+),", jt).group(1)), 4
            else:
                size = align = {"Int32": 4, "Int64": 8, "Float64": 8}[jt]
            off = (off + align - 1) // align * align
            cf = getattr(cstruct, fname)
            assert (cf.offset, cf.size) == (off, size), (jname, fname, cf.offset, off)
            off += size
        assert (off + 7) // 8 * 8 == C.sizeof(cstruct) or off == C.sizeof(cstruct), jname
    # the stub must start from the library's defaults (sensealg = LDE_SENSE_DISCRETE, the reference's ForwardDiffSensitivity), not from zeros
    src = open(os.path.join(ROOT, "julia", "LdeNative.jl")).read()
    assert "lde_problem_desc_default" in src.split("mutable struct LdeHandle")[0]
    # every entry point the stub ccalls is an export of the library
    from latentdiffeq_amd import _lib as _l
    called = set(re.findall(r"ccall\(\(:(\w+), liblde\)", src))
    assert called and called <= set(_l.EXPORTS), called - set(_l.EXPORTS)


def test_hidden_accumulator_registers_are_not_touched_by_the_compiler():
    """k_mlpb / k_mlpc keep their weight-gradient tiles in AGPRs a[A0 : 256) that only inline asm names (csrc/lde_mlpb.h); LLVM has no way
    to reserve them, so the built object is checked: outside the asm's own three instruction forms nothing in those kernels writes or
    reads that range (build.py runs the same check and fails the build). The checker itself is checked with a bound it must trip over."""
    from latentdiffeq_amd import build, check_agprs
    build.build_lib()
    obj = os.path.join(build.OBJ, "lde_mlp.o")
    assert os.path.exists(obj)
    assert check_agprs.check_object(obj) == []
    saved = dict(check_agprs.A0)
    try:
        check_agprs.A0.update({k: 8 for k in saved})          # the compiler's own copies live above a8: the check must see them
        assert len(check_agprs.check_object(obj)) > 0
    finally:
        check_agprs.A0.update(saved)
