# What a maintainer runs on a machine with Julia, SciMLSensitivity (with ext/SciMLSensitivityHIPAdjExt.jl wired in) and an MI355X:
#   HIPADJ_LIBRARY=/path/to/libhipadj.so julia --project=julia/test julia/test/runtests.jl
# Not executed in the build image (no Julia there).  Mirrors test/Core3/adjoint.jl:1157-1241 (Lorenz, dg = u - 2) on a matrix-state
# problem and compares the device gradients with the reference's own CPU adjoint at rtol 1e-6 (BASELINE.json north_star).
using Test, HIPAdj, SciMLSensitivity, OrdinaryDiffEq, Zygote, Random, ForwardDiff

@testset "layout" begin
    @test HIPAdj.check_layout()
    @test hipadj_version() == 110
    @test occursin("libhiprtc", runtime_compiler())          # the build toolkit's hiprtc (a Julia process carries no other)
end

lorenz!(du, u, p, t) = (du[1] = p[1] * (u[2] - u[1]); du[2] = u[1] * (p[2] - u[3]) - u[2]; du[3] = u[1] * u[2] - p[3] * u[3]; nothing)
# the same right-hand side over a matrix state: every column an independent trajectory (docs/src/tutorials/data_parallel.md:11-75)
function lorenz_cols!(dU, U, p, t)
    for j in axes(U, 2)
        lorenz!(view(dU, :, j), view(U, :, j), p, t)
    end
    return nothing
end

Random.seed!(1)
N = 64
U0 = [1.0, 0.0, 0.0] .+ 0.1 .* randn(3, N)
p = [10.0, 28.0, 8 / 3]
tspan = (0.0, 2.0)
prob = ODEProblem(lorenz_cols!, U0, tspan, p)
loss(sol) = sum(abs2, Array(sol) .- 2) / 2

for (inner, name) in ((InterpolatingAdjoint(), "Interpolating"), (BacksolveAdjoint(checkpointing = true), "Backsolve"),
                      (GaussAdjoint(), "Gauss"), (QuadratureAdjoint(abstol = 1e-12, reltol = 1e-12), "Quadrature"))
    @testset "$name, RK4 fixed step" begin
        dev = HIPBatchedAdjoint(inner; model = builtin_model(:lorenz))
        g_ref = Zygote.gradient((u0, p) -> loss(solve(prob, RK4(); u0, p, dt = 0.01, adaptive = false, saveat = 0.1, sensealg = inner)), U0, p)
        g_dev = Zygote.gradient((u0, p) -> loss(solve(prob, RK4(); u0, p, dt = 0.01, adaptive = false, saveat = 0.1, sensealg = dev)), U0, p)
        @test isapprox(g_dev[1], g_ref[1]; rtol = 1e-6)
        @test isapprox(g_dev[2], g_ref[2]; rtol = 1e-6)
    end
end

@testset "Julia f -> device (Symbolics)" begin
    using Symbolics
    m = HIPAdj.register_model(lorenz!, 3, 3; name = "lorenz_from_julia")
    dev = HIPBatchedAdjoint(InterpolatingAdjoint(); model = m)
    g_b = Zygote.gradient(p -> loss(solve(prob, RK4(); p, dt = 0.01, adaptive = false, saveat = 0.1,
                                          sensealg = HIPBatchedAdjoint(InterpolatingAdjoint(); model = builtin_model(:lorenz)))), p)
    g_s = Zygote.gradient(p -> loss(solve(prob, RK4(); p, dt = 0.01, adaptive = false, saveat = 0.1, sensealg = dev)), p)
    @test isapprox(g_s[1], g_b[1]; rtol = 1e-10)
end

@testset "mass matrix (test/Core3/adjoint.jl:1315-1376)" begin
    # M u' = A u + p + e2 sum(p): the reference's own mass-matrix problem, on a matrix state of N identical columns
    A = [1.0 2 3; 4 5 6; 7 8 9]; mm = -[1.0 2 4; 2 3 7; 1 3 41]
    m = HIPAdj.register_model("affine3_mm", 3, 3;
        f = "du[0] = 1.0*u[0] + 2.0*u[1] + 3.0*u[2] + p[0]; du[1] = 4.0*u[0] + 5.0*u[1] + 6.0*u[2] + p[1] + (p[0] + p[1] + p[2]); du[2] = 7.0*u[0] + 8.0*u[1] + 9.0*u[2] + p[2];",
        vjp_u = "out[0] = 1.0*lam[0] + 4.0*lam[1] + 7.0*lam[2]; out[1] = 2.0*lam[0] + 5.0*lam[1] + 8.0*lam[2]; out[2] = 3.0*lam[0] + 6.0*lam[1] + 9.0*lam[2];",
        vjp_p = "out[0] = lam[0] + lam[1]; out[1] = 2.0*lam[1]; out[2] = lam[2] + lam[1];", mass_matrix = mm)
    foo(du, u, p, t) = (du .= A * u .+ p; du[2] += sum(p); nothing)
    pm = [1.0, 2.0, 3.0]; ts = 0:0.01:1
    prob_mm = ODEProblem(ODEFunction(foo, mass_matrix = mm), [1.0, 2.0, 3.0], (0.0, 1.0), pm)
    sol_mm = solve(prob_mm, Rodas4(), reltol = 1.0e-14, abstol = 1.0e-14)
    dg(out, u, p, t, i) = out .= 1
    du0_ref, dp_ref = adjoint_sensitivities(sol_mm, Rodas4(); t = ts, dgdu_discrete = dg, abstol = 1.0e-14, reltol = 1.0e-14, sensealg = InterpolatingAdjoint())
    dev = HIPBatchedAdjoint(InterpolatingAdjoint(); model = m)
    cols = ODEProblem((dU, U, p, t) -> nothing, repeat([1.0, 2.0, 3.0], 1, 4), (0.0, 1.0), pm)      # the right-hand side lives in the device model
    sol = HIPAdj.hip_solve(cols, Tsit5(), dev; saveat = collect(ts), abstol = 1.0e-12, reltol = 1.0e-12)
    du0, dp = adjoint_sensitivities(sol, Tsit5(); sensealg = dev, dgdu_discrete = dg)
    @test isapprox(dp ./ 4, dp_ref; rtol = 1e-8)            # shared p: the device sums over the 4 identical columns
    @test isapprox(du0[:, 1], du0_ref; rtol = 1e-8)         # lam(t0), the reference's convention
end

# ---- round 6: Rosenbrock23 and the reference's singular-mass-matrix problem (test/Core3/adjoint.jl:1434-1530; test/Core2/stiff_adjoints.jl:66-80) ----------------------------
@testset "stiff: Rosenbrock23, semi-explicit DAE (test/Core3/adjoint.jl:1434-1530)" begin
    function rober(du, u, p, t)
        du[1] = -p[1] * u[1] + p[3] * u[2] * u[3]; du[2] = p[1] * u[1] - p[2] * u[2]^2 - p[3] * u[2] * u[3]; du[3] = u[1] + u[2] + u[3] - 1
        return nothing
    end
    Mdae = [1.0 0 0; 0 1.0 0; 0 0 0]
    p = [0.04, 3.0e7, 1.0e4]; ts = [50.0, 100.0]
    dg_singular(out, u, p, t, i) = (fill!(out, 0); out[end] = 1)
    prob = ODEProblem(ODEFunction(rober, mass_matrix = Mdae), [1.0, 0.0, 1.0], (0.0, 100.0), p)
    sol_ref = solve(prob, Rodas4(autodiff = AutoFiniteDiff()), reltol = 1.0e-12, abstol = 1.0e-12, initializealg = BrownFullBasicInit())
    _, dp_ref = adjoint_sensitivities(sol_ref, Rodas4(autodiff = AutoFiniteDiff()); t = ts, dgdu_discrete = dg_singular, abstol = 1.0e-8, reltol = 1.0e-8,
                                      sensealg = QuadratureAdjoint(), maxiters = Int(1.0e6), initializealg = BrownFullBasicInit())
    m = register_model("rober_dae_jl", 3, 3;
        f = "du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = u[0] + u[1] + u[2] - 1.0;", mass_matrix = Mdae)   # VJPs by dual numbers
    for inner in (InterpolatingAdjoint(), GaussAdjoint(), GaussKronrodAdjoint(), QuadratureAdjoint(abstol = 1.0e-14, reltol = 1.0e-8), InterpolatingAdjoint(checkpointing = true))
        dev = HIPBatchedAdjoint(inner; model = m)
        cols = ODEProblem((dU, U, p, t) -> nothing, repeat([1.0, 0.0, 1.0], 1, 4), (0.0, 100.0), p)
        sol = HIPAdj.hip_solve(cols, Rosenbrock23(), dev; saveat = ts, abstol = 1.0e-10, reltol = 1.0e-8)
        @test maximum(abs.(sum(sol.u[:, :, 1], dims = 1) .- 1)) < 1e-8                    # the constraint, from the inconsistent start [1, 0, 1]
        _, dp = adjoint_sensitivities(sol, Rosenbrock23(); sensealg = dev, dgdu_discrete = dg_singular)
        @test isapprox(dp ./ 4, dp_ref; rtol = 1e-5)                                       # the reference's own bar (:1483)
    end
    @test_throws Exception HIPAdj.hip_solve(ODEProblem((dU, U, p, t) -> nothing, repeat([1.0, 0.0, 0.0], 1, 4), (0.0, 1.0), p), Tsit5(),
                                            HIPBatchedAdjoint(InterpolatingAdjoint(); model = m); saveat = [1.0])      # a DAE model on an explicit stepper: refused by name
end

@testset "ContinuousCallback: the bouncing ball (test/Callbacks2/continuous_callbacks.jl:10-24, 212-224)" begin
    ball(du, u, p, t) = (du[1] = u[2]; du[2] = -p[1]; nothing)
    u0 = [5.0, 0.0]; p = [9.8, 0.8]; ts = collect(0.0:0.5:2.5)
    condition(u, t, integrator) = u[1]
    affect!(integrator) = (integrator.u[2] = -integrator.p[2] * integrator.u[2])
    cb = ContinuousCallback(condition, affect!, save_positions = (false, false))
    g(θ) = sum(Array(solve(ODEProblem(ball, θ[1:2], (0.0, 2.5), θ[3:4]), Tsit5(); callback = cb, abstol = 1.0e-12, reltol = 1.0e-12, saveat = ts)))
    dref = ForwardDiff.gradient(g, [u0; p])                                                # the reference's own yardstick (:129-139)
    m = register_model("ball_jl", 2, 2; f = "du[0] = u[1]; du[1] = -p[0];")              # VJPs by dual numbers
    set_continuous_callback!(m, "c = u[0];", "un[1] = -p[1] * u[1];")
    dg_ones(out, u, p, t, i) = (out .= 1)
    for inner in (InterpolatingAdjoint(), BacksolveAdjoint(), GaussAdjoint(), GaussKronrodAdjoint(), QuadratureAdjoint(abstol = 1.0e-14, reltol = 1.0e-12))
        dev = HIPBatchedAdjoint(inner; model = m)
        cols = ODEProblem((dU, U, p, t) -> nothing, repeat(u0, 1, 4), (0.0, 2.5), p)
        sol = HIPAdj.hip_solve(cols, Tsit5(), dev; saveat = ts, abstol = 1.0e-12, reltol = 1.0e-12)
        @test all(event_counts(sol.handle) .== 1)
        du0, dp = adjoint_sensitivities(sol, Tsit5(); sensealg = dev, dgdu_discrete = dg_ones)
        @test isapprox(du0[:, 1], dref[1:2]; rtol = 1e-5) && isapprox(dp ./ 4, dref[3:4]; rtol = 1e-5)      # the reference's bar (:140-145)
    end
    # save_positions = (true, true), the constructor's default: the saved event states come from event_states, their cotangents go back through set_event_cotangents!
    cb2 = ContinuousCallback(condition, affect!)
    g2(θ) = sum(Array(solve(ODEProblem(ball, θ[1:2], (0.0, 2.5), θ[3:4]), Tsit5(); callback = cb2, abstol = 1.0e-12, reltol = 1.0e-12, saveat = ts)))
    dref2 = ForwardDiff.gradient(g2, [u0; p])
    dev = HIPBatchedAdjoint(InterpolatingAdjoint(); model = m)
    sol = HIPAdj.hip_solve(ODEProblem((dU, U, p, t) -> nothing, repeat(u0, 1, 1), (0.0, 2.5), p), Tsit5(), dev; saveat = ts, abstol = 1.0e-12, reltol = 1.0e-12)
    tev, ul, ur = event_states(sol.handle)
    @test isapprox(tev[1, 1], sqrt(2 * 5 / 9.8); atol = 1e-10) && abs(ul[1, 1, 1]) < 1e-9 && isapprox(ur[2, 1, 1], -0.8 * ul[2, 1, 1]; atol = 1e-12)
    dl = zeros(size(ul)); dr = zeros(size(ur)); dl[:, 1, 1] .= 1; dr[:, 1, 1] .= 1
    set_event_cotangents!(sol.handle, dl, dr)
    du0, dp = adjoint_sensitivities(sol, Tsit5(); sensealg = dev, dgdu_discrete = dg_ones)
    @test isapprox(du0[:, 1], dref2[1:2]; rtol = 1e-5) && isapprox(dp, dref2[3:4]; rtol = 1e-5)
end

# ---- wide runtime models (ABI 106): the Julia emitter writes the same SPMD text as the Python host's (tests/golden/dense_chain_bodies.json), and the
# 2 -> 50 -> 2 neural ODE of docs/src/Benchmark.md:62 registers and compiles for gfx950
@testset "wide models" begin
    gold_path = joinpath(@__DIR__, "..", "..", "tests", "golden", "dense_chain_bodies.json")
    if isfile(gold_path)
        txt = read(gold_path, String)
        # no JSON package in the test environment: compare through the escaped form the file holds
        esc(s) = replace(replace(s, "\\" => "\\\\"), "\n" => "\\n", "\"" => "\\\"")
        f, vjp, np, nw = dense_chain_bodies((2, 50, 2); input_power = 3)
        @test np == 252
        @test occursin(esc(f), txt) && occursin(esc(vjp), txt)
        f3, vjp3, np3, _ = dense_chain_bodies((3, 16, 24, 3))
        @test np3 == 547 && occursin(esc(f3), txt) && occursin(esc(vjp3), txt)
        m = register_wide_model("node_2_50_2", 2, np; f = f, vjp = vjp, lds_doubles = nw)      # compiles forward + the four sweeps (no device needed)
        @test m.n == 2 && m.np == 252
        # ABI 109: the structure is declared, the library selects the family; widths that do not reproduce the model are refused
        @test declare_dense_chain!(m, (2, 50, 2); input_power = 3) === m
        @test_throws HIPAdj.HipadjError declare_dense_chain!(m, (2, 40, 2); input_power = 3)
        mc = dense_chain_model("chain_2_64_64_2", (2, 64, 64, 2))
        @test mc.n == 2 && mc.np == 2 * 64 + 64 + 64 * 64 + 64 + 2 * 64 + 2
    end
end
