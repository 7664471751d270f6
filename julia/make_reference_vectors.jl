# make_reference_vectors.jl — run the REFERENCE (gabrevaya/LatentDiffEq.jl at its pinned Manifest) on the inputs of
# tests/golden/ref_inputs.bson and write tests/golden/ref_outputs.bson: the reference-made vectors this repository cannot make itself
# (no Julia in its build image — SURVEY.md §8c, DESIGN.md §7: parity is pinned by independent known answers, NOT by the reference, until
# somebody runs this script). Once the file exists, tests/test_reference_vectors.py (CPU: the oracle; `-m gpu`: the kernels through the C ABI)
# replays the reference's own accepted step sequences and gates ẑ ≤ 2e-5, gradients ≤ 1e-4 of their largest entry.
#
#     cd /path/to/LatentDiffEq.jl                      # the checkout whose Manifest.toml pins OrdinaryDiffEq 6.27.1, SciMLSensitivity 7.10.0,
#     julia --project=. -e 'using Pkg; Pkg.instantiate()'     #   DiffEqFlux 1.52.0, Flux 0.13.6, Zygote 0.6.47 (Julia 1.8.1) [REF Manifest.toml:3, :979, :1200, :304, :452, :1468]
#     julia --project=. /path/to/this/repo/julia/make_reference_vectors.jl /path/to/this/repo/tests/golden
#
# What is recorded per case (arrays in the reference's layouts, Float32 unless noted):
#   zhat        [D' × B × T]  diffeq_layer(decoder, l̂, t) as an inference call (no AD)              [REF src/models/GOKU.jl:98-130], [REF src/models/LatentODE.jl:61-78]
#   zhat_train  [D' × B × T]  the value Zygote.pullback returns (GOKU: the solve on dual numbers — its step control sees the partials)
#   dz0 [D × B], dtheta [P × B] (GOKU), dW [n] (NODE: the gradient the reference never applies — SURVEY.md B2)   from back(dz)
#   steps_t, steps_dt          Vector{Vector{Float64}}: start time and size of every accepted step of the PRIMAL solve, per trajectory (NODE: one sequence)
#   steps_train_t, steps_train_dt   the same for the solve on dual numbers (GOKU), what ForwardDiffSensitivity differentiates
#   versions                   the package versions that produced the file
using LatentDiffEq, OrdinaryDiffEq, SciMLSensitivity, DiffEqFlux, Flux, Zygote, ForwardDiff, BSON, Pkg

const dir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
const inputs = BSON.load(joinpath(dir, "ref_inputs.bson"))

# the right-hand sides of the example, verbatim in meaning [REF examples/pendulum_friction-less/pendulum.jl:19-26, :65-74] (without the
# ModelingToolkit pass the example puts in front: same function, no extra dependency)
function pendulum!(du, u, p, t)
    x, y = u
    G = 10.0f0
    L = p[1]
    du[1] = y
    du[2] = -G / L * sin(x)
end
function pendulum_friction!(du, u, p, t)
    x, y = u
    G = 10.0f0
    L = p[1]
    b, m = 0.7f0, 1.0f0
    du[1] = y
    du[2] = -G / L * sin(x) - b / m * y
end

steps_of(ts) = (collect(Float64, ts[1:end-1]), collect(Float64, diff(ts)))

function goku_case(c)
    ẑ₀, θ̂, t, Δ = c[:z0], c[:theta], c[:ts], c[:dz]
    f! = c[:kind] == "pendulum" ? pendulum! : pendulum_friction!
    prob = ODEProblem(f!, Float32[1.0, 1.0], (0.0f0, 1.0f0), Float32[1.0])
    # the fields diffeq_layer reads off `decoder.diffeq` [REF src/models/GOKU.jl:105-108]; defaults as Pendulum() carries them [REF pendulum.jl:8-11]
    diffeq = (prob = prob, solver = Tsit5(), sensealg = ForwardDiffSensitivity(), kwargs = (abstol = c[:abstol], reltol = c[:reltol]))
    decoder = LatentDiffEq.Decoder(GOKU_basic(), (identity, diffeq, identity))
    ẑ = LatentDiffEq.diffeq_layer(decoder, (ẑ₀, θ̂), t)
    ẑtr, back = Zygote.pullback((a, b) -> LatentDiffEq.diffeq_layer(decoder, (a, b), t), ẑ₀, θ̂)
    dẑ₀, dθ̂ = back(Δ)
    B = size(ẑ₀, 2)
    st, sdt, tt, tdt = Vector{Float64}[], Vector{Float64}[], Vector{Float64}[], Vector{Float64}[]
    n = size(ẑ₀, 1) + size(θ̂, 1)
    tag = typeof(ForwardDiff.Tag(:lde_reference_vectors, Float32))
    seeds = ForwardDiff.construct_seeds(ForwardDiff.Partials{n,Float32})
    for i in 1:B
        p_i = remake(prob; u0 = ẑ₀[:, i], p = θ̂[:, i], tspan = (t[1], t[end]))
        sol = solve(p_i, Tsit5(); diffeq.kwargs...)                       # no saveat: sol.t = every accepted step
        a, b = steps_of(sol.t); push!(st, a); push!(sdt, b)
        # the solve SciMLSensitivity runs under ForwardDiffSensitivity: (u0, p) seeded with D + P partials, the same solver, the same kwargs
        u0d = [ForwardDiff.Dual{tag}(ẑ₀[k, i], seeds[k]) for k in 1:size(ẑ₀, 1)]
        pd = [ForwardDiff.Dual{tag}(θ̂[k, i], seeds[size(ẑ₀, 1) + k]) for k in 1:size(θ̂, 1)]
        sold = solve(remake(prob; u0 = u0d, p = pd, tspan = (t[1], t[end])), Tsit5(); diffeq.kwargs...)
        a, b = steps_of(sold.t); push!(tt, a); push!(tdt, b)
    end
    return Dict{Symbol,Any}(:zhat => Float32.(ẑ), :zhat_train => Float32.(ẑtr), :dz0 => Float32.(dẑ₀), :dtheta => Float32.(dθ̂),
                            :steps_t => st, :steps_dt => sdt, :steps_train_t => tt, :steps_train_dt => tdt)
end

function node_case(c)
    ẑ₀, t, Δ, W, sizes = c[:z0], c[:ts], c[:dz], c[:W], Int.(c[:sizes])
    dudt0 = Chain(Dense(sizes[1], sizes[2], relu), Dense(sizes[2], sizes[3], relu), Dense(sizes[3], sizes[4]))     # [REF examples/pendulum_friction-less/nODE.jl:12-14]
    _, re = Flux.destructure(dudt0)
    kwargs = (abstol = c[:abstol], reltol = c[:reltol])
    mk(w) = (dudt = re(w), solver = Tsit5(), neural_model = NeuralODE, latent_dim_in = sizes[1], latent_dim_out = sizes[1], augment_dim = 0, kwargs = kwargs)
    layer(w, a) = LatentDiffEq.diffeq_layer(LatentDiffEq.Decoder(LatentODE(), (identity, mk(w), identity)), a, t)
    ẑ = layer(W, ẑ₀)
    ẑtr, back = Zygote.pullback(layer, W, ẑ₀)
    dW, dẑ₀ = back(Δ)
    sol = NeuralODE(re(W), (t[1], t[end]), Tsit5(); kwargs...)(ẑ₀)        # no saveat: every accepted step of the ONE coupled solve
    a, b = steps_of(sol.t)
    return Dict{Symbol,Any}(:zhat => Float32.(ẑ), :zhat_train => Float32.(ẑtr), :dz0 => Float32.(dẑ₀), :dW => Float32.(dW),
                            :steps_t => [a], :steps_dt => [b])
end

out = Dict{Symbol,Any}()
for (name, c) in inputs
    @info "case" name
    out[name] = c[:kind] == "node" ? node_case(c) : goku_case(c)
end
deps = Pkg.dependencies()
ver(n) = string(first(v.version for (_, v) in deps if v.name == n))
out[:versions] = Dict{Symbol,Any}(:julia => string(VERSION), (Symbol(n) => ver(n) for n in ("OrdinaryDiffEq", "SciMLSensitivity", "DiffEqFlux", "Flux", "Zygote", "ForwardDiff"))...)
bson(joinpath(dir, "ref_outputs.bson"), out)
@info "wrote" joinpath(dir, "ref_outputs.bson")
