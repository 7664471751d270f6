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
# Always start from the LIBRARY's defaults (Tsit5, abstol 1e-6, reltol 1e-3, PI constants, and
# sensealg = LDE_SENSE_PARALLEL_CHECKPOINTED — the 5 µs time-parallel adjoint, not the 68 µs sequential one), then override.
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
    function LdeHandle(desc::LdeDesc)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:lde_create, liblde), Cint, (Ref{LdeDesc}, Ref{Ptr{Cvoid}}), desc, out)
        rc == 0 || error("lde_create failed: $rc")
        h = new(out[]); finalizer(h -> ccall((:lde_destroy, liblde), Cvoid, (Ptr{Cvoid},), h.ptr), h); h
    end
end

# one handle per `diffeq` struct, built lazily from its fields (solver, kwargs...)  [REF GOKU.jl:105-108]
const _handles = IdDict{Any,LdeHandle}()
native(diffeq::Pendulum) = get!(_handles, diffeq) do
    LdeHandle(LdeDesc(; rhs_kind = 0, solver = diffeq.solver isa RK4 ? 1 : 0, pairs(diffeq.kwargs)...))
end
native(diffeq::NODE) = get!(_handles, diffeq) do
    sizes = Int32[size(l.weight, 2) for l in diffeq.dudt.layers]; push!(sizes, size(diffeq.dudt.layers[end].weight, 1))
    LdeHandle(LdeDesc(; rhs_kind = 2, state_dim = diffeq.latent_dim_in, param_dim = 0, augment_dim = diffeq.augment_dim,
                      n_layers = length(sizes) - 1, layer_sizes = ntuple(i -> i <= length(sizes) ? sizes[i] : Int32(0), 7),
                      batching = 1, pairs(diffeq.kwargs)...))
end

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
    d = decoder.diffeq; h = native(d)
    p, _ = Flux.destructure(d.dudt)                               # vec(W) column-major then b, per layer — the ABI's order
    ccall((:lde_set_weights_device, liblde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Ptr{Cvoid}),
          h.ptr, pointer(p), length(p), AMDGPU.stream().stream)
    ẑ, _ = lde_forward!(h, ẑ₀, nothing, t, d.latent_dim_out)
    return transform_after_diffeq(ẑ, d)
end

# --- AD boundary: Zygote differentiates through the call via this rule ------------------------------------------------
function ChainRulesCore.rrule(::typeof(diffeq_layer), decoder::Decoder{T}, l̂, t) where {T<:GOKU}
    ẑ₀, θ̂ = l̂; h = native(decoder.diffeq)
    ẑ, ts = lde_forward!(h, ẑ₀, θ̂, t, size(ẑ₀, 1))
    function pullback(Δ)
        Δẑ = ROCArray{Float32}(unthunk(Δ)); B, T = size(ẑ₀, 2), length(ts)
        dẑ₀ = similar(ẑ₀); dθ̂ = similar(θ̂)
        ccall((:lde_adjoint, liblde), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32},
               Ptr{Float32}, Ptr{Cvoid}),
              h.ptr, pointer(ẑ), pointer(θ̂), ts, T, B, pointer(Δẑ), pointer(dẑ₀), pointer(dθ̂), C_NULL, AMDGPU.stream().stream)
        return NoTangent(), NoTangent(), (dẑ₀, dθ̂), NoTangent()
    end
    return ẑ, pullback
end

# LatentODE / NODE: the pullback returns the tangent of `dudt` as well — the NODE weights are NOT trained in the reference
# (SURVEY.md B2: `NODE` is no @functor, `Flux.params(model)` never reaches `diffeq.dudt` [REF nODE.jl:3-32], [REF LatentODE.jl:70]);
# with this rule (and `Flux.@functor NODE (dudt,)` so that the optimiser sees them) they are.
function ChainRulesCore.rrule(::typeof(diffeq_layer), decoder::Decoder{LatentODE}, ẑ₀, t)
    d = decoder.diffeq; h = native(d)
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
mutable struct LdeChainDesc       # mirror of lde_chain_desc (include/lde.h)
    abi_version::Int32 = 1;  n_layers::Int32 = 0
    sizes::NTuple{7,Int32} = ntuple(_ -> Int32(0), 7)
    activation::NTuple{6,Int32} = ntuple(_ -> Int32(0), 6)     # 0 identity, 1 relu, 2 tanh, 3 σ, 4 softplus
    skip::NTuple{6,Int32} = ntuple(_ -> Int32(0), 6)           # 1: SkipConnection(Dense, +)
end

mutable struct LdeRnnDesc          # mirror of lde_rnn_desc (include/lde.h)
    abi_version::Int32 = 1;  cell::Int32 = 0   # 0 RNN relu, 1 RNN tanh, 2 LSTM
    n_layers::Int32 = 0;     sizes::NTuple{5,Int32} = ntuple(_ -> Int32(0), 5);  reverse::Int32 = 0
end


# ---- sensealg tags → lde_sensealg (include/lde.h) --------------------------------------------------------------------
const LDE_SENSE_BACKSOLVE_CHECKPOINTED = Int32(0)   # InterpolatingAdjoint / BacksolveAdjoint(checkpointing = true)
const LDE_SENSE_BACKSOLVE              = Int32(1)   # BacksolveAdjoint(checkpointing = false)
const LDE_SENSE_PARALLEL_CHECKPOINTED  = Int32(2)   # the library's default for the GOKU path: the time-parallel continuous adjoint
const LDE_SENSE_DISCRETE               = Int32(3)   # ForwardDiffSensitivity()'s own meaning: the exact derivative of the discrete solve
# `Pendulum()` carries ForwardDiffSensitivity() [REF examples/pendulum_friction-less/pendulum.jl:11]. Mapping it to LDE_SENSE_DISCRETE
# reproduces the reference's gradient DEFINITION (to f32 round-off on the same steps); the library's default (2) is the continuous
# adjoint, which agrees with it to solver tolerance and is the faster pullback at the metric's batch size. The choice is the host's:
sensealg_code(::Any) = LDE_SENSE_PARALLEL_CHECKPOINTED
# sensealg_code(::ForwardDiffSensitivity) = LDE_SENSE_DISCRETE        # (uncomment for the reference's exact definition)

# LDE_SENSE_DISCRETE: lde_forward writes a step record that the matching lde_adjoint reads. With several forwards of one `diffeq` in
# flight before their pullbacks (Zygote's tape) each rrule owns its record: allocate lde_step_record_bytes(h, B, T) device bytes, hand
# them over before lde_forward AND before lde_adjoint.
function with_step_record(f, h::LdeHandle, rec)
    rec === nothing && return f()
    ccall((:lde_set_step_record, liblde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h.ptr, pointer(rec), length(rec)) == 0 || error("lde_set_step_record")
    try
        return f()
    finally
        ccall((:lde_set_step_record, liblde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h.ptr, C_NULL, 0)
    end
end
new_step_record(h::LdeHandle, desc_sensealg, B, T) = desc_sensealg == LDE_SENSE_DISCRETE ?
    ROCArray{UInt8}(undef, ccall((:lde_step_record_bytes, liblde), Int64, (Ptr{Cvoid}, Cint, Cint), h.ptr, B, T)) : nothing
# (in the rrules above: `rec = new_step_record(h, sensealg, B, T)`; wrap the lde_forward! call and the lde_adjoint ccall in
#  `with_step_record(h, rec) do … end`; keep `rec` alive in the pullback closure.)

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
