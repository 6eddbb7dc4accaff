# What a maintainer runs on a machine with Julia, SciMLSensitivity (with ext/SciMLSensitivityHIPAdjExt.jl wired in) and an MI355X:
#   HIPADJ_LIBRARY=/path/to/libhipadj.so julia --project=julia/test julia/test/runtests.jl
# Not executed in the build image (no Julia there).  Mirrors test/Core3/adjoint.jl:1157-1241 (Lorenz, dg = u - 2) on a matrix-state
# problem and compares the device gradients with the reference's own CPU adjoint at rtol 1e-6 (BASELINE.json north_star).
using Test, HIPAdj, SciMLSensitivity, OrdinaryDiffEq, Zygote, Random

@testset "layout" begin
    @test HIPAdj.check_layout()
    @test hipadj_version() == 102
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
