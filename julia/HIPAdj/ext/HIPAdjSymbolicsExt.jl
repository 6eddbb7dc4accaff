"""
A path from a Julia right-hand side to the device: trace `f!(du, u, p, t)` with Symbolics, differentiate symbolically, and emit the
three C bodies `hipadj_model_register` takes — the reference's `ODEFunction(f!; vjp, vjp_p)` seam (src/derivative_wrappers.jl:284-359,
test/Core3/user_vjp.jl:77-134) with C text instead of closures.

    using HIPAdj, Symbolics
    lorenz!(du, u, p, t) = (du[1] = p[1] * (u[2] - u[1]); du[2] = u[1] * (p[2] - u[3]) - u[2]; du[3] = u[1] * u[2] - p[3] * u[3]; nothing)
    model = HIPAdj.register_model(lorenz!, 3, 3; name = "lorenz_from_julia")

Never executed in the build image (no Julia there); the C side of the same seam — text in, kernels out — is exercised by
tests/test_gpu_parity.py (`test_runtime_models_match_oracle`) and, from a traced host-language function, by the Python mirror
(`sa.DeviceFunction.from_callable`, tests/test_trace.py).
"""
module HIPAdjSymbolicsExt

using HIPAdj
using Symbolics

"""
Body (the text between the braces) of the C function Symbolics' own C back end prints for `exprs`:
`build_function(exprs, args...; target = CTarget(), lhsname, rhsnames)` emits `void f(double* lhs, const double* a, ...) { lhs[0] = ...; }`
with 0-based indexing, `pow` for powers and the libm names for elementary functions — the dialect `hipadj_model_register` compiles.
"""
function cbody(exprs, args, lhs::Symbol, rhsnames::Vector{Symbol})
    src = Symbolics.build_function(exprs, args...; target = Symbolics.CTarget(), fname = :hipadj_body, lhsname = lhs,
                                   rhsnames = rhsnames, header = false)
    i = findfirst('{', src); j = findlast('}', src)
    (i === nothing || j === nothing) && error("unexpected output of Symbolics' C target")
    return String(strip(src[(i + 1):(j - 1)]))
end

function bodies(f!, n::Integer, np::Integer)
    Symbolics.@variables t u[1:n] p[1:np] lam[1:n]
    us = collect(u); ps = collect(p); ls = collect(lam)
    du = Vector{Symbolics.Num}(undef, n)
    f!(du, us, ps, t)
    J = Symbolics.jacobian(du, us)              # df/du
    P = Symbolics.jacobian(du, ps)              # df/dp
    vju = Symbolics.simplify.(transpose(J) * ls)   # (df/du)^T lam, un-negated like f.vjp
    vjp = Symbolics.simplify.(transpose(P) * ls)   # (df/dp)^T lam, un-negated like f.vjp_p
    return cbody(du, (us, ps, t), :du, [:u, :p, :t]),
           cbody(vju, (ls, us, ps, t), :out, [:lam, :u, :p, :t]),
           cbody(vjp, (ls, us, ps, t), :out, [:lam, :u, :p, :t])
end

"""
    HIPAdj.register_model(f!, n, np; name = "julia_model", auto_vjp = false)

`auto_vjp = true` registers only `f` and lets the device build both VJPs with forward-mode dual numbers (`autojacvec = true`).
"""
function HIPAdj.register_model(f!::Function, n::Integer, np::Integer; name::AbstractString = "julia_model", auto_vjp::Bool = false)
    fb, vu, vp = bodies(f!, n, np)
    return auto_vjp ? HIPAdj.register_model(name, n, np; f = replace(fb, "double" => "real")) :
           HIPAdj.register_model(name, n, np; f = fb, vjp_u = vu, vjp_p = vp)
end

end # module
