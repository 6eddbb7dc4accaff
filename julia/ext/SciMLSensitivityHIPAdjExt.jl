"""
SciMLSensitivityHIPAdjExt — package extension OF SciMLSensitivity.jl, triggered by `HIPAdj` (julia/HIPAdj).  Drop this file into the
reference's `ext/` and add to its Project.toml

    [weakdeps]
    HIPAdj = "6b1e3f0a-52c7-4c8e-9d4b-1f7a2c9e8d31"
    [extensions]
    SciMLSensitivityHIPAdjExt = "HIPAdj"

exactly as `SciMLSensitivityMooncakeExt = "Mooncake"` is wired (Project.toml:50-56, ext/SciMLSensitivityMooncakeExt.jl:123).

What it adds — methods for `sensealg::HIPAdj.HIPBatchedAdjoint`, nothing else:

  B1  `SciMLBase._concrete_solve_adjoint(prob::AbstractODEProblem, alg, sensealg::HIPBatchedAdjoint, u0::AbstractMatrix, p, originator, args...; kw...)`
      the method SciMLBase's `solve` AD rule reaches for a MATRIX-STATE problem whose columns are independent trajectories — the
      reference's documented batching pattern (docs/src/tutorials/data_parallel.md:11-75, test/Core5/size_handling_adjoint.jl:37-70).
      Same contract as src/concrete_solve.jl:523-1042: returns `(out, pullback)`; honours `saveat` (number / list / empty), `save_start`,
      `save_end`, `save_idxs`, `only_end`; the pullback accepts every cotangent form the reference accepts (dense array, vector of
      arrays / VectorOfArray, `Tangent` with `.u`, thunks, NoTangent entries; :776-869) and returns the originator-dependent tuple
      (:1027-1039).
  B2  `SciMLSensitivity._adjoint_sensitivities(sol::HIPAdjSolution, sensealg::HIPBatchedAdjoint, alg; t, dgdu_discrete, ...)`
      the direct API (src/sensitivity_interface.jl:426-526): `(du0, dp')`.

An `EnsembleProblem` itself never reaches `_concrete_solve_adjoint` (ensembles are N separate solves, test/Core4/ensembles.jl:13-31),
so the ensemble entry is `HIPAdj`-side: `HIPAdj.ensemble_u0_p(ensprob, trajectories)` stacks the `prob_func` outputs into the matrix state.

Written in an image without Julia: NOT executed.  The `ccall` sequence it performs (through HIPAdj.Handle / forward! / adjoint!) is
replayed from C with the same struct layout and argument order by tests/c/julia_seam.c.
"""
module SciMLSensitivityHIPAdjExt

using SciMLSensitivity
using SciMLSensitivity: InterpolatingAdjoint, BacksolveAdjoint, GaussAdjoint, GaussKronrodAdjoint, QuadratureAdjoint, ischeckpointing
using SciMLBase: SciMLBase, ReturnCode
using ChainRulesCore: ChainRulesCore, NoTangent, ZeroTangent, AbstractThunk, AbstractZero, Tangent, unthunk
using RecursiveArrayTools: AbstractVectorOfArray
import LinearAlgebra
import HIPAdj
using HIPAdj: HIPBatchedAdjoint, HIPAdjSolution, Handle, forward!, adjoint!

algid(::InterpolatingAdjoint) = HIPAdj.ALG_INTERPOLATING
algid(::BacksolveAdjoint) = HIPAdj.ALG_BACKSOLVE
algid(::GaussAdjoint) = HIPAdj.ALG_GAUSS
algid(::QuadratureAdjoint) = HIPAdj.ALG_QUADRATURE
algid(::GaussKronrodAdjoint) = HIPAdj.ALG_GAUSS_KRONROD

# the steppers the device implements for lane and wide models; anything else is an error, as unsupported combinations are in the reference
function stepper_of(alg)
    nm = nameof(typeof(alg))
    nm === :RK4 && return HIPAdj.STEPPER_RK4_FIXED
    nm === :Tsit5 && return HIPAdj.STEPPER_TSIT5_ADAPTIVE
    nm === :Rosenbrock23 && return HIPAdj.STEPPER_ROSENBROCK23_ADAPTIVE      # the stiff stepper of the lane-per-trajectory models (test/Core2/stiff_adjoints.jl:66-80); also the one
                                                                             # stepper for a model with a singular mass matrix (set_mass_matrix!, test/Core3/adjoint.jl:1434-1530)
    error("HIPBatchedAdjoint: the device steppers are RK4() (fixed dt), Tsit5() and Rosenbrock23() (adaptive); got $(nm)")
end

"""
The save / loss times `ts` as the forward pass of `_concrete_solve_adjoint` forms them (src/concrete_solve.jl:713-770).
Returns `(ts, only_end)`.
"""
function save_times(tspan, saveat, dt, save_start, save_end, stepper)
    t0, t1 = Float64(tspan[1]), Float64(tspan[2])
    if saveat isa Number
        ts = collect(t0:abs(Float64(saveat)):t1)
        ts[end] == t1 || push!(ts, t1)                       # fix_endpoints (:725)
    elseif isempty(saveat)
        stepper == HIPAdj.STEPPER_RK4_FIXED || error("saveat = [] (every step) needs the fixed-step RK4(): an adaptive solve's step times are not known up front")
        S = round(Int, (t1 - t0) / dt)
        ts = [t0 + k * dt for k in 0:S]; ts[end] = t1
        save_start || popfirst!(ts)                          # :740-750
        save_end || pop!(ts)
    else
        ts = sort(collect(Float64, saveat))                  # :752
    end
    return ts, (length(ts) == 1 && ts[1] == t1)
end

"Cotangent of the saved solution at time index i as a plain array (or `nothing` for a structural zero) — src/concrete_solve.jl:778-869."
function cotangent_at(Δ, i::Int, M::Int, only_end::Bool)
    Δu = Δ isa Tangent ? unthunk.(Δ.u) : Δ
    if Δu isa AbstractVectorOfArray
        x = Δu.u[i]
    elseif Δu isa AbstractArray{<:AbstractArray} || Δu isa AbstractVector{<:Any} && !(eltype(Δu) <: Number)
        x = Δu[i]
    elseif only_end && (ndims(Δu) < 3 || size(Δu)[end] != M)
        x = Δu                                               # user did sol[end] on only_end: the bare state cotangent
    else
        x = reshape(Δu, :, size(Δu)[end])[:, i]              # dense array, time is the LAST axis (:842-851)
    end
    x = x isa AbstractThunk ? unthunk(x) : x
    return x isa AbstractZero ? nothing : x
end

"""
Pack the cotangent of `out` into the ABI's `Δ[N][M][n]` == Julia `(n, M, N)`; `idxs`: linear indices of `save_idxs` into the `(n, N)`
state (zeros elsewhere, `_out[_save_idxs] .= ...` :790-824).
"""
function pack_cotangent(Δ, n::Int, N::Int, M::Int, only_end::Bool, idxs)
    buf = zeros(Float64, n, M, N)
    (Δ isa AbstractZero) && return buf
    state = zeros(Float64, n, N)
    for i in 1:M
        x = cotangent_at(Δ, i, M, only_end)
        x === nothing && continue
        if idxs === nothing
            copyto!(state, reshape(collect(Float64, x), n, N))
        else
            fill!(state, 0.0)
            state[idxs] .= vec(collect(Float64, x))
        end
        @views buf[:, i, :] .= state
    end
    return buf
end

"""
The mass matrix of the matrix-state problem would be `(n N) x (n N)`; the device wants the per-trajectory `n x n` block, attached to the
device model (`HIPAdj.register_model(...; mass_matrix = M)` / `HIPAdj.set_mass_matrix!`).  A problem that carries its own is refused
instead of being silently integrated with M = I (src/adjoint_common.jl:110-135).
"""
function check_mass_matrix(prob)
    mm = prob.f.mass_matrix
    (mm === LinearAlgebra.I || mm isa LinearAlgebra.UniformScaling && isone(mm.λ)) ||
        error("HIPBatchedAdjoint: attach the n x n mass matrix to the device model (HIPAdj.register_model(...; mass_matrix = M)); " *
              "prob.f.mass_matrix of the matrix-state problem is not read")
    return nothing
end

function SciMLBase._concrete_solve_adjoint(
        prob::SciMLBase.AbstractODEProblem, alg, sensealg::HIPBatchedAdjoint,
        u0::AbstractMatrix, p, originator::SciMLBase.ADOriginator, args...;
        save_start = true, save_end = true, saveat = eltype(prob.tspan)[], save_idxs = nothing,
        dt = nothing, abstol = 1.0e-6, reltol = 1.0e-3, checkpoints = nothing, kwargs...
    )
    inner = sensealg.inner
    model = sensealg.model
    check_mass_matrix(prob)
    n, N = size(u0)
    n == model.n || error("HIPBatchedAdjoint: the device model has $(model.n) states, u0 has $n rows (columns are trajectories)")
    p_shared = p isa AbstractVector
    (p_shared ? length(p) == model.np : size(p) == (model.np, N)) ||
        error("HIPBatchedAdjoint: p must be a vector of $(model.np) shared parameters or a $(model.np) x $N matrix (one column per trajectory)")
    stepper = stepper_of(alg)
    stepper == HIPAdj.STEPPER_RK4_FIXED && dt === nothing && error("RK4() needs dt")
    ts, only_end = save_times(prob.tspan, saveat, dt, save_start, save_end, stepper)
    M = length(ts)
    no_start = !save_start && (Float64(prob.tspan[1]) in ts)          # :962
    u0d = convert(Matrix{Float64}, u0)
    pd = p_shared ? convert(Vector{Float64}, p) : convert(Matrix{Float64}, p)
    h = Handle(model; alg = algid(inner), stepper, N, tspan = prob.tspan, dt = something(dt, 0.0), ts,
        checkpointing = ischeckpointing(inner), checkpoints,
        quad_abstol = inner isa QuadratureAdjoint ? inner.abstol : 1.0e-6,
        quad_reltol = inner isa QuadratureAdjoint ? inner.reltol : 1.0e-3,
        no_start, p_shared, device = sensealg.device, devices = sensealg.devices, time_segments = sensealg.time_segments, max_steps = sensealg.max_steps,
        abstol, reltol)
    raw = forward!(h, u0d, pd)                                          # (n, M, N)
    idxs = save_idxs === nothing ? nothing : (save_idxs isa Number ? [save_idxs] : collect(save_idxs))
    # the saved solution as the reference hands it out: one state per save time; for the matrix-state problem a state is (n, N)
    us = map(1:M) do i
        full = Matrix{Float64}(raw[:, i, :])
        idxs === nothing ? full : (save_idxs isa Number ? full[save_idxs] : full[idxs])
    end
    sol = SciMLBase.build_solution(prob, alg, ts, us; retcode = ReturnCode.Success)
    out = SciMLBase.sensitivity_solution(sol, us, ts)

    function adjoint_sensitivity_backpass(Δ)
        Δ = Δ isa AbstractThunk ? unthunk(Δ) : Δ
        buf = pack_cotangent(Δ, n, N, M, only_end, idxs)
        du0, dp = adjoint!(h, buf)
        du0 = reshape(du0, size(u0))                                    # :978
        dp_tangent = p === nothing || p isa SciMLBase.NullParameters ? nothing : reshape(dp, size(p))   # :980-986
        return if originator isa SciMLBase.TrackerOriginator || originator isa SciMLBase.ReverseDiffOriginator
            (NoTangent(), NoTangent(), du0, dp_tangent, NoTangent(), ntuple(_ -> NoTangent(), length(args))...)
        else
            (NoTangent(), NoTangent(), NoTangent(), du0, dp_tangent, NoTangent(), ntuple(_ -> NoTangent(), length(args))...)
        end
    end
    return out, adjoint_sensitivity_backpass
end

# ---- B2: the direct API -------------------------------------------------------------------------------------------------
"""
    sol = HIPAdj.hip_solve(prob, alg, sensealg; u0, p, saveat, dt, ...)      (the forward solve of the direct API)

`adjoint_sensitivities(sol, alg; sensealg, t = sol.t, dgdu_discrete)` then returns `(du0 (n, N), dp')` like the reference
(src/sensitivity_interface.jl:500-508: `dp` as a row).  `dgdu_discrete(out, u, p, t, i)` is the reference's in-place loss gradient,
called once per trajectory and loss time on the host to fill the cotangent block the device consumes.
"""
function HIPAdj.hip_solve(prob::SciMLBase.AbstractODEProblem, alg, sensealg::HIPBatchedAdjoint; u0 = prob.u0, p = prob.p, saveat,
        dt = nothing, abstol = 1.0e-6, reltol = 1.0e-3, checkpoints = nothing)
    check_mass_matrix(prob)
    n, N = size(u0)
    stepper = stepper_of(alg)
    ts, _ = save_times(prob.tspan, saveat, dt, true, true, stepper)
    inner = sensealg.inner
    h = Handle(sensealg.model; alg = algid(inner), stepper, N, tspan = prob.tspan, dt = something(dt, 0.0), ts,
        checkpointing = ischeckpointing(inner), checkpoints,
        quad_abstol = inner isa QuadratureAdjoint ? inner.abstol : 1.0e-6, quad_reltol = inner isa QuadratureAdjoint ? inner.reltol : 1.0e-3,
        p_shared = p isa AbstractVector, device = sensealg.device, devices = sensealg.devices, time_segments = sensealg.time_segments, max_steps = sensealg.max_steps, abstol, reltol)
    raw = forward!(h, convert(Matrix{Float64}, u0), p isa AbstractVector ? convert(Vector{Float64}, p) : convert(Matrix{Float64}, p))
    return HIPAdjSolution(h, raw, ts, p)
end

function SciMLSensitivity._adjoint_sensitivities(sol::HIPAdjSolution, sensealg::HIPBatchedAdjoint, alg;
        t = sol.t, dgdu_discrete = nothing, kwargs...)
    t == sol.t || error("t must equal the save times the forward solve was run with")
    dgdu_discrete === nothing && error("dgdu_discrete required (continuous costs: HIPAdj.register_model(...; dgdu, dgdp))")
    n, M, N = size(sol.u)
    Δ = zeros(Float64, n, M, N)
    for j in 1:N, i in 1:M
        pj = sol.p isa AbstractVector ? sol.p : view(sol.p, :, j)
        dgdu_discrete(view(Δ, :, i, j), view(sol.u, :, i, j), pj, t[i], i)
    end
    du0, dp = adjoint!(sol.handle, Δ)
    return du0, dp'                                                      # :500-508
end
SciMLSensitivity.adjoint_sensitivities(sol::HIPAdjSolution, args...; sensealg::HIPBatchedAdjoint, kwargs...) =
    SciMLSensitivity._adjoint_sensitivities(sol, sensealg, args...; kwargs...)

end # module
