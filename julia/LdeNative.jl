# LdeNative.jl — the `ccall` binding of liblde.so (include/lde.h) for gabrevaya/LatentDiffEq.jl: what a maintainer adds as
# `src/native/LdeNative.jl` and `include`s from `src/LatentDiffEq.jl` behind the model files [REF src/LatentDiffEq.jl:17-22].
#
# It replaces the BODIES of
#     diffeq_layer(::Decoder{<:GOKU}, (ẑ₀, θ̂), t)        [REF src/models/GOKU.jl:98-130]
#     diffeq_layer(::Decoder{LatentODE}, ẑ₀, t)            [REF src/models/LatentODE.jl:61-78]
# and supplies their `ChainRulesCore.rrule`s; nothing else of the package changes (constructors, `default_layers`, the training
# scripts stay as they are). Julia is not in the build image of this repository: the file is source to be read and adopted, its
# struct mirrors are checked field by field against the C structs (tests/test_abi.py::test_documented_julia_struct_layouts), and the
# same entry points are exercised through ctypes by every `-m gpu` test. INTEGRATION.md §2 walks through it; §2b–§2d show the
# bindings of the rows either side of the solve (dense chains, recurrent pattern extractor, sample / loss / optimiser).
#
# Requires: AMDGPU.jl (ROCArray, AMDGPU.stream()), ChainRulesCore, Flux (destructure) — all already in the package's Manifest
# except AMDGPU.jl, which takes the place of CUDA.jl's `gpu` on this path.
module LdeNative

using AMDGPU, ChainRulesCore, Flux
using OrdinaryDiffEq: Tsit5, RK4                       # the solver tags the `diffeq` structs carry [REF src/LatentDiffEq.jl:3]
using SciMLSensitivity: ForwardDiffSensitivity, InterpolatingAdjoint, BacksolveAdjoint   # … and the sensealg tags [REF src/LatentDiffEq.jl:5]
import ..LatentDiffEq: diffeq_layer, transform_after_diffeq, Decoder, GOKU, LatentODE

const liblde = "liblde.so"

mutable struct LdeDesc                          # mirror of lde_problem_desc (include/lde.h), field for field
    abi_version::Int32;  rhs_kind::Int32
    state_dim::Int32;    param_dim::Int32;  augment_dim::Int32
    n_layers::Int32;     layer_sizes::NTuple{7,Int32}
    activation::Int32;   solver::Int32;     batching::Int32;  sensealg::Int32
    adaptive::Int32;     maxiters::Int64
    dt::Float64;         abstol::Float64;   reltol::Float64;  dtmin::Float64
    qmin::Float64;       qmax::Float64;     gamma::Float64
    beta1::Float64;      beta2::Float64
    LdeDesc() = new()
end
# Always start from the LIBRARY's defaults (Tsit5, abstol 1e-6, reltol 1e-3, PI constants, and sensealg = LDE_SENSE_DISCRETE — what
# `Pendulum()`'s ForwardDiffSensitivity() means [REF examples/pendulum_friction-less/pendulum.jl:8-11]), then override.
function LdeDesc(; kw...)
    d = LdeDesc()
    ccall((:lde_problem_desc_default, liblde), Cint, (Ref{LdeDesc},), d) == 0 || error("lde_problem_desc_default")
    for (k, v) in kw
        setfield!(d, k, convert(fieldtype(LdeDesc, k), v))
    end
    return d
end

mutable struct LdeHandle
    ptr::Ptr{Cvoid}
    sensealg::Int32
    function LdeHandle(desc::LdeDesc)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:lde_create, liblde), Cint, (Ref{LdeDesc}, Ref{Ptr{Cvoid}}), desc, out)
        rc == 0 || error("lde_create failed: $rc")
        h = new(out[], desc.sensealg); finalizer(h -> ccall((:lde_destroy, liblde), Cvoid, (Ptr{Cvoid},), h.ptr), h); h
    end
end

# ---- sensealg tags → lde_sensealg (include/lde.h) --------------------------------------------------------------------
const LDE_SENSE_BACKSOLVE_CHECKPOINTED = Int32(0)   # InterpolatingAdjoint / BacksolveAdjoint(checkpointing = true)
const LDE_SENSE_BACKSOLVE              = Int32(1)   # BacksolveAdjoint(checkpointing = false)
const LDE_SENSE_PARALLEL_CHECKPOINTED  = Int32(2)   # the time-parallel continuous adjoint (analytic right-hand sides): selectable, not a default
const LDE_SENSE_DISCRETE               = Int32(3)   # ForwardDiffSensitivity(): the exact derivative of the discrete solve
# `Pendulum()` carries ForwardDiffSensitivity() [REF examples/pendulum_friction-less/pendulum.jl:8-11], splatted into solve() at
# [REF src/models/GOKU.jl:107, :121]: LDE_SENSE_DISCRETE reproduces that gradient definition (to f32 round-off on the same steps; the one
# stated deviation: the primal step sequence is differentiated, upstream's norm also sees the dual partials — measured at 1.5e-4 in ẑ and
# 4e-5 in the gradient at the example's tolerances, tests/test_oracle_dual.py).
sensealg_code(::ForwardDiffSensitivity) = LDE_SENSE_DISCRETE
sensealg_code(::InterpolatingAdjoint)   = LDE_SENSE_BACKSOLVE_CHECKPOINTED
sensealg_code(s::BacksolveAdjoint)      = s.checkpointing ? LDE_SENSE_BACKSOLVE_CHECKPOINTED : LDE_SENSE_BACKSOLVE
sensealg_code(::Any)                    = LDE_SENSE_DISCRETE            # (any other discrete-exact tag, e.g. ReverseDiffAdjoint)
solver_code(::Tsit5) = Int32(0)
solver_code(::RK4)   = Int32(1)                                         # fixed step only: adaptive = false, dt = h through kwargs

# one handle per `diffeq` struct, built lazily from its fields (solver, sensealg, kwargs...)  [REF GOKU.jl:105-108]. The analytic right-hand
# sides are a closed menu (include/lde.h: lde_rhs_kind) and the example structs live outside the package
# [REF examples/pendulum_friction-less/pendulum.jl], so the example names its kind with one line next to its struct:
#     LatentDiffEq.LdeNative.rhs_kind(::Pendulum) = 0            # LDE_RHS_PENDULUM;  Pendulum_friction: 1
rhs_kind(diffeq) = error("LdeNative: no native right-hand side for $(typeof(diffeq)); define LdeNative.rhs_kind(::$(nameof(typeof(diffeq))))")
const _handles = IdDict{Any,LdeHandle}()
desc_kwargs(kw) = (k => (k === :adaptive ? Int32(v) : v) for (k, v) in pairs(kw) if k !== :saveat)
function native(diffeq)                                                   # GOKU path: an analytic right-hand side
    get!(_handles, diffeq) do
        LdeHandle(LdeDesc(; rhs_kind = rhs_kind(diffeq), solver = solver_code(diffeq.solver), sensealg = sensealg_code(diffeq.sensealg),
                          desc_kwargs(diffeq.kwargs)...))
    end
end
function native_node(diffeq)                                              # LatentODE path: `NODE` [REF examples/pendulum_friction-less/nODE.jl:3-32]
    get!(_handles, diffeq) do
        sizes = Int32[size(l.weight, 2) for l in diffeq.dudt.layers]; push!(sizes, size(diffeq.dudt.layers[end].weight, 1))
        LdeHandle(LdeDesc(; rhs_kind = 2, state_dim = diffeq.latent_dim_in, param_dim = 0, augment_dim = diffeq.augment_dim,
                          n_layers = length(sizes) - 1, layer_sizes = ntuple(i -> i <= length(sizes) ? sizes[i] : Int32(0), 7),
                          batching = 1, solver = solver_code(diffeq.solver),
                          sensealg = LDE_SENSE_BACKSOLVE_CHECKPOINTED,       # DiffEqFlux's NeuralODE default, InterpolatingAdjoint [REF src/models/LatentODE.jl:67-70]
                          desc_kwargs(diffeq.kwargs)...))
    end
end
sensealg_of(h::LdeHandle) = h.sensealg

function lde_forward!(h, ẑ₀::ROCMatrix{Float32}, θ̂, t::AbstractVector, D′)
    B, T = size(ẑ₀, 2), length(t)
    ẑ = ROCArray{Float32}(undef, D′, B, T); ret = ROCArray{Int32}(undef, B)
    ts = collect(Float64, t)                                   # host array, as the range in the scripts [REF model_train.jl:44]
    rc = ccall((:lde_forward, liblde), Cint,
               (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}, Cint, Cint, Ptr{Float32}, Ptr{Int32}, Ptr{Cvoid}),
               h.ptr, pointer(ẑ₀), θ̂ === nothing ? C_NULL : pointer(θ̂), ts, T, B, pointer(ẑ), pointer(ret),
               AMDGPU.stream().stream)
    rc == 0 || error(unsafe_string(ccall((:lde_last_error, liblde), Cstring, (Ptr{Cvoid},), h.ptr)))
    return ẑ, ts
end

# --- the two methods that replace the bodies of [REF GOKU.jl:98-130] and [REF LatentODE.jl:61-78] ---------------
function diffeq_layer(decoder::Decoder{T}, l̂, t) where {T<:GOKU}
    ẑ₀, θ̂ = l̂
    ẑ, _ = lde_forward!(native(decoder.diffeq), ẑ₀, θ̂, t, size(ẑ₀, 1))
    ẑ = transform_after_diffeq(permutedims(ẑ, [1, 3, 2]), decoder.diffeq)   # hook sees [D×T×B] as today [REF GOKU.jl:124]
    return permutedims(ẑ, [1, 3, 2])
end
function diffeq_layer(decoder::Decoder{LatentODE}, ẑ₀, t)
    d = decoder.diffeq; h = native_node(d)
    p, _ = Flux.destructure(d.dudt)                               # vec(W) column-major then b, per layer — the ABI's order
    ccall((:lde_set_weights_device, liblde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Cvoid}),
          h.ptr, pointer(p), length(p), AMDGPU.stream().stream)
    ẑ, _ = lde_forward!(h, ẑ₀, nothing, t, d.latent_dim_out)
    return transform_after_diffeq(ẑ, d)
end

# --- AD boundary: Zygote differentiates through the call via this rule ------------------------------------------------
function ChainRulesCore.rrule(::typeof(diffeq_layer), decoder::Decoder{T}, l̂, t) where {T<:GOKU}
    ẑ₀, θ̂ = l̂; h = native(decoder.diffeq)
    B, Tn = size(ẑ₀, 2), length(t)
    # LDE_SENSE_DISCRETE (the default, as in the reference): this solve's accepted steps travel with THIS pullback (Zygote may hold several
    # forwards of one `diffeq` before their pullbacks): a record per rrule, handed over before lde_forward and before lde_adjoint
    rec = new_step_record(h, h.sensealg, B, Tn)
    ẑ, ts = with_step_record(h, rec) do
        lde_forward!(h, ẑ₀, θ̂, t, size(ẑ₀, 1))
    end
    function pullback(Δ)
        Δẑ = ROCArray{Float32}(unthunk(Δ))
        dẑ₀ = similar(ẑ₀); dθ̂ = similar(θ̂)
        rec2 = record_that_holds(h, rec, ẑ, θ̂, ts)                # (a solve with more accepted steps than the record holds: grow it, repeat the forward solve)
        with_step_record(h, rec2) do
            rc = ccall((:lde_adjoint, liblde), Cint,
                       (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32},
                        Ptr{Float32}, Ptr{Cvoid}),
                       h.ptr, pointer(ẑ), pointer(θ̂), ts, Tn, B, pointer(Δẑ), pointer(dẑ₀), pointer(dθ̂), C_NULL, AMDGPU.stream().stream)
            rc == 0 || error(unsafe_string(ccall((:lde_last_error, liblde), Cstring, (Ptr{Cvoid},), h.ptr)))
        end
        return NoTangent(), NoTangent(), (dẑ₀, dθ̂), NoTangent()
    end
    # (a non-identity transform_after_diffeq hook is differentiated by Zygote around this rule, as for the LatentODE rule below)
    return ẑ, pullback
end

# LatentODE / NODE: the pullback returns the tangent of `dudt` as well — the NODE weights are NOT trained in the reference
# (SURVEY.md B2: `NODE` is no @functor, `Flux.params(model)` never reaches `diffeq.dudt` [REF nODE.jl:3-32], [REF LatentODE.jl:70]);
# with this rule (and `Flux.@functor NODE (dudt,)` so that the optimiser sees them) they are.
function ChainRulesCore.rrule(::typeof(diffeq_layer), decoder::Decoder{LatentODE}, ẑ₀, t)
    d = decoder.diffeq; h = native_node(d)
    p, re = Flux.destructure(d.dudt)                              # flat θ in the ABI's order; `re` rebuilds a Chain from a flat vector
    ccall((:lde_set_weights_device, liblde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Cvoid}),
          h.ptr, pointer(p), length(p), AMDGPU.stream().stream)
    ẑ, ts = lde_forward!(h, ẑ₀, nothing, t, d.latent_dim_out)     # [D′ × B × T], D′ = latent_dim_in + augment_dim
    y = transform_after_diffeq(ẑ, d)
    function pullback(Δ)
        # (a non-identity transform_after_diffeq hook is differentiated by Zygote around this rule: apply the rule to the
        #  un-hooked layer and compose; with the default identity hook Δ is ∂L/∂ẑ itself)
        Δẑ = ROCArray{Float32}(unthunk(Δ)); B, T = size(ẑ₀, 2), length(ts)
        dẑ₀ = similar(ẑ₀)                                        # [latent_dim_in × B]: the augmented rows' cotangent is dropped by the library
        dW = AMDGPU.zeros(Float32, length(p))                     # lde_adjoint ACCUMULATES (dW +=): start from zero
        rc = ccall((:lde_adjoint, liblde), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32},
                    Ptr{Float32}, Ptr{Cvoid}),
                   h.ptr, pointer(ẑ), C_NULL, ts, T, B, pointer(Δẑ), pointer(dẑ₀), C_NULL, pointer(dW), AMDGPU.stream().stream)
        rc == 0 || error(unsafe_string(ccall((:lde_last_error, liblde), Cstring, (Ptr{Cvoid},), h.ptr)))
        # data parallel (one process per GPU, batch sharded by trajectory): the ONE collective of the path, before the optimiser step —
        #   ccall((:lde_comm_allreduce_f32, liblde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Cvoid}), comm, pointer(dW), length(dW), stream)
        d_dudt = re(dW)                                           # the flat gradient in the shape of the Chain: a structural tangent
        ∂decoder = Tangent{typeof(decoder)}(; diffeq = Tangent{typeof(d)}(; dudt = d_dudt))
        return NoTangent(), ∂decoder, dẑ₀, NoTangent()
    end
    return y, pullback
end

# ---- struct mirrors of the rows either side of the solve (INTEGRATION.md §2b, §2c use them) --------------------------
Base.@kwdef mutable struct LdeChainDesc       # mirror of lde_chain_desc (include/lde.h)
    abi_version::Int32 = 1;  n_layers::Int32 = 0
    sizes::NTuple{7,Int32} = ntuple(_ -> Int32(0), 7)
    activation::NTuple{6,Int32} = ntuple(_ -> Int32(0), 6)     # 0 identity, 1 relu, 2 tanh, 3 σ, 4 softplus
    skip::NTuple{6,Int32} = ntuple(_ -> Int32(0), 6)           # 1: SkipConnection(Dense, +)
end

Base.@kwdef mutable struct LdeRnnDesc          # mirror of lde_rnn_desc (include/lde.h)
    abi_version::Int32 = 1;  cell::Int32 = 0   # 0 RNN relu, 1 RNN tanh, 2 LSTM
    n_layers::Int32 = 0;     sizes::NTuple{5,Int32} = ntuple(_ -> Int32(0), 5);  reverse::Int32 = 0
end


# LDE_SENSE_DISCRETE: lde_forward writes a step record that the matching lde_adjoint reads. With several forwards of one `diffeq` in
# flight before their pullbacks (Zygote's tape) each rrule owns its record: lde_step_record_bytes(h, B, T) device bytes, handed over
# before lde_forward AND before lde_adjoint, and taken back afterwards (the handle must not keep a pointer into a buffer the GC may free).
function with_step_record(f, h::LdeHandle, rec)
    rec === nothing && return f()
    ccall((:lde_set_step_record, liblde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h.ptr, pointer(rec), length(rec)) == 0 || error("lde_set_step_record")
    try
        return f()
    finally
        ccall((:lde_set_step_record, liblde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h.ptr, C_NULL, 0)
    end
end
new_step_record(h::LdeHandle, sensealg, B, T) = sensealg == LDE_SENSE_DISCRETE ?
    ROCArray{UInt8}(undef, ccall((:lde_step_record_bytes, liblde), Int64, (Ptr{Cvoid}, Cint, Cint), h.ptr, B, T)) : nothing
# The reference's ForwardDiffSensitivity differentiates any solve up to maxiters [REF src/models/GOKU.jl:121]; a record holds
# "record_capacity" steps per trajectory (default max(64, 4T)). Before the pullback: did it hold the solve? If not, raise the capacity and
# repeat the (deterministic) forward solve into a larger record — never NaN gradients into an optimiser (include/lde.h: lde_step_record_status).
function record_that_holds(h::LdeHandle, rec, ẑ, θ̂, ts)
    rec === nothing && return rec
    Dp, B, T = size(ẑ)
    nmax = Ref{Int32}(0); cap = Ref{Int32}(0)
    ccall((:lde_step_record_status, liblde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ref{Int32}, Ref{Int32}, Ptr{Cvoid}),
          h.ptr, pointer(rec), B, T, nmax, cap, AMDGPU.stream().stream) == 0 || error("lde_step_record_status")
    need = ccall((:lde_step_record_bytes, liblde), Int64, (Ptr{Cvoid}, Cint, Cint), h.ptr, B, T)
    nmax[] <= cap[] && length(rec) >= need && return rec
    nmax[] > cap[] && set_option!(h, "record_capacity", max(nmax[] + 8, 2 * cap[]))
    rec2 = new_step_record(h, LDE_SENSE_DISCRETE, B, T)
    ẑ₀ = ẑ[:, :, 1]                                                # ẑ(t₁) is ẑ₀ itself (saveat includes t₁)
    with_step_record(h, rec2) do
        lde_forward!(h, ẑ₀, θ̂, ts, Dp)
    end
    return rec2
end

# options that are not part of the `diffeq` struct (a library is not steered by environment variables):
set_option!(h::LdeHandle, key::AbstractString, value::Real) =
    ccall((:lde_set_option, liblde), Cint, (Ptr{Cvoid}, Cstring, Cdouble), h.ptr, key, value) == 0 || error("lde_set_option($key)")
# set_option!(h, "adjoint_overwrite", 1)   # lde_adjoint WRITES dW: the `AMDGPU.zeros` in the LatentODE pullback becomes `similar`


# What the loaded binary was built and validated with (compiler version, the register check's verdict)
build_info() = unsafe_string(ccall((:lde_build_info, liblde), Cstring, ()))

# LDE_BATCH_COUPLED_GLOBAL without the host in the loop (include/lde.h: lde_set_global_sum_peers): one mailbox per rank in fine-grained device
# memory, mapped on every peer (HIP IPC handles exchanged once, e.g. over MPI); `mailboxes[r]` = rank r's as mapped on THIS device.
global_sum_mailbox_bytes(nranks::Integer) = Int(ccall((:lde_global_sum_mailbox_bytes, liblde), Int64, (Cint,), nranks))
function set_global_sum_peers!(h::LdeHandle, rank::Integer, mailboxes::Vector{Ptr{Cvoid}}, global_batch::Integer)
    rc = ccall((:lde_set_global_sum_peers, liblde), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Int64),
               h.ptr, rank, length(mailboxes), mailboxes, global_batch)
    rc == 0 || error("lde_set_global_sum_peers: " * unsafe_string(ccall((:lde_last_error, liblde), Cstring, (Ptr{Cvoid},), h.ptr)))
    return h
end

end # module LdeNative
