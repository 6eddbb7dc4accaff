"""
    HIPAdj

Thin Julia binding of `libhipadj.so` (C ABI: `include/hipadj.h`, version 110) — the MI355X-native batched continuous-adjoint
engine.  This package holds ONLY the `ccall` layer and the types a SciMLSensitivity extension dispatches on:

  * `HIPBatchedAdjoint(inner; model, device)` — an `AbstractAdjointSensitivityAlgorithm` that wraps one of the reference's
    `InterpolatingAdjoint` / `BacksolveAdjoint` / `GaussAdjoint` / `GaussKronrodAdjoint` / `QuadratureAdjoint` and names the
    device model of the right-hand side.  Loading this package next to SciMLSensitivity activates
    `ext/SciMLSensitivityHIPAdjExt.jl`, which adds the `_concrete_solve_adjoint` / `_adjoint_sensitivities` methods for it
    (the same mechanism as `ext/SciMLSensitivityMooncakeExt.jl:123`).
  * `Handle`, `forward!`, `adjoint!` — the three calls behind it.
  * `register_model` — the `ODEFunction(f!; vjp, vjp_p)` seam (src/derivative_wrappers.jl:284-359) with C text instead of closures;
    with Symbolics loaded, `register_model(f!, n, np)` emits the text from a Julia function (ext/HIPAdjSymbolicsExt.jl).

Layouts: Julia is column-major, the ABI row-major — a Julia `Matrix{Float64}` of size `(n, N)` IS the ABI's `u0[N][n]`, an
`Array{Float64,3}` of size `(n, M, N)` IS `out[N][M][n]`.  No copies, no transposes.

This file was written in an image without Julia and has not been executed; `tests/c/julia_seam.c` performs the same call sequence
with the same struct layout from C (CPU: loads, exports, loud NO_DEVICE; GPU: numbers against the oracle), so the ABI side of every
`ccall` below is checked.
"""
module HIPAdj

import Libdl
using SciMLBase: SciMLBase

export HIPBatchedAdjoint, HIPAdjSolution, DeviceModel, builtin_model, register_model, register_wide_model, set_wide_cost!, dense_chain_bodies, dense_chain_model, declare_dense_chain!, set_mass_matrix!, set_affect!, set_continuous_callback!, event_counts, event_states, event_components, set_event_cotangents!, affect_apply, affect_vjp, Handle, forward!, adjoint!, hip_solve, ensemble_u0_p, hipadj_version, runtime_compiler

# ---------------------------------------------------------------------------------------------------------------------
# library
# ---------------------------------------------------------------------------------------------------------------------
const LIB = Ref{Ptr{Cvoid}}(C_NULL)

"Path of libhipadj.so: `ENV[\"HIPADJ_LIBRARY\"]`, else the loader's search path."
libpath() = get(ENV, "HIPADJ_LIBRARY", "libhipadj.so")

function lib()
    if LIB[] == C_NULL
        LIB[] = Libdl.dlopen(libpath(), Libdl.RTLD_NOW | Libdl.RTLD_GLOBAL)
        v = ccall(Libdl.dlsym(LIB[], :hipadj_version), Cint, ())
        v == 110 || error("libhipadj ABI version $v, this binding was written for 110")
    end
    return LIB[]
end
sym(name::Symbol) = Libdl.dlsym(lib(), name)

hipadj_version() = Int(ccall(sym(:hipadj_version), Cint, ()))
"Which hiprtc compiles the runtime-registered models (`hipadj_runtime_compiler`): the build toolkit's — a Julia process has no other."
function runtime_compiler()
    buf = Vector{UInt8}(undef, 1024)
    check(ccall(sym(:hipadj_runtime_compiler), Cint, (Ptr{UInt8}, Int32), buf, Int32(length(buf))))
    return GC.@preserve buf unsafe_string(pointer(buf))
end

# mirrors `hipadj_config` (include/hipadj.h) field by field; Julia lays an isbits struct out like C does
struct HipadjConfig
    struct_size::UInt32
    model::Int32
    alg::Int32
    stepper::Int32
    dims::NTuple{4, Int32}
    ntraj::Int64
    t0::Float64
    t1::Float64
    dt::Float64
    nsave::Int32
    save_times::Ptr{Float64}
    loss_kind::Int32
    loss_shift::Float64
    checkpointing::Int32
    ckpt_stride::Int32
    quad_abstol::Float64
    quad_reltol::Float64
    no_start::Int32
    p_shared::Int32
    device::Int32
    time_segments::Int32
    cont_cost::Int32
    max_steps::Int32
    abstol::Float64
    reltol::Float64
    ncheckpoints::Int32
    checkpoints::Ptr{Float64}
    loss_scale::Float64
    ndevices::Int32
    device_ids::Ptr{Int32}
    reference_literal::Int32
    family::Int32              # hipadj_family (ABI 109): FAMILY_AUTO = the library selects the kernel family of a declared dense chain, FAMILY_AS_REGISTERED
end

# byte offsets of include/hipadj.h as the C compiler sees them (tests/c/julia_seam.c asserts the same table with offsetof)
const CONFIG_OFFSETS = (0, 4, 8, 12, 16, 32, 40, 48, 56, 64, 72, 80, 88, 96, 100, 104, 112, 120, 124, 128, 132, 136, 140, 144,
                        152, 160, 168, 176, 184, 192, 200, 204)
const CONFIG_SIZE = 208

function check_layout()
    sizeof(HipadjConfig) == CONFIG_SIZE || error("HipadjConfig is $(sizeof(HipadjConfig)) bytes, the header says $CONFIG_SIZE")
    for (i, off) in enumerate(CONFIG_OFFSETS)
        fieldoffset(HipadjConfig, i) == off ||
            error("HipadjConfig.$(fieldname(HipadjConfig, i)) at $(fieldoffset(HipadjConfig, i)), the header says $off")
    end
    return true
end

# enums of the header
const ALG_INTERPOLATING, ALG_BACKSOLVE, ALG_GAUSS, ALG_QUADRATURE, ALG_GAUSS_KRONROD = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const STEPPER_RK4_FIXED, STEPPER_TSIT5_ADAPTIVE = Int32(0), Int32(1)
const STEPPER_ROSENBROCK23_ADAPTIVE = Int32(3)      # Rosenbrock23 for the lane-per-trajectory models (include/hipadj.h HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE; test/Core2/stiff_adjoints.jl:66-80)
const STEPPER_ETDRK4_FIXED = Int32(2)      # exponential RK4 for the PDE family (include/hipadj.h HIPADJ_STEPPER_ETDRK4_FIXED; OrdinaryDiffEq's ETDRK4)
const LOSS_COTANGENT, LOSS_LSQ_SHIFT, LOSS_LSQ_DATA, LOSS_MODEL = Int32(0), Int32(1), Int32(2), Int32(3)
const MODEL_USER_BASE = Int32(1000)
const FAMILY_AUTO, FAMILY_AS_REGISTERED, FAMILY_MFMA = Int32(0), Int32(1), Int32(3)      # hipadj_family
const ACT_TANH = Int32(1)                                                                 # hipadj_activation

struct HipadjError <: Exception
    status::Int
    msg::String
end
Base.showerror(io::IO, e::HipadjError) = print(io, "hipadj status ", e.status, ": ", e.msg)

last_error(h::Ptr{Cvoid}) = unsafe_string(ccall(sym(:hipadj_last_error), Cstring, (Ptr{Cvoid},), h))

"Non-zero status -> exception carrying `hipadj_last_error` (the reference `error(...)`s on misuse, src/interpolating_adjoint.jl:321-326)."
function check(rc::Integer, h::Ptr{Cvoid} = C_NULL)
    rc == 0 && return nothing
    throw(HipadjError(Int(rc), last_error(h)))
end

# ---------------------------------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------------------------------
"A right-hand side that lives on the device: a compiled-in registry entry or a runtime-registered one."
struct DeviceModel
    id::Int32
    dims::NTuple{4, Int32}
    n::Int
    np::Int
end

function model_sizes(id::Integer, dims::NTuple{4, Int32})
    n = Ref{Int32}(0); np = Ref{Int32}(0)
    d = Ref(dims)
    check(ccall(sym(:hipadj_model_sizes), Cint, (Int32, Ptr{Int32}, Ref{Int32}, Ref{Int32}), Int32(id), d, n, np))
    return Int(n[]), Int(np[])
end

const BUILTIN = Dict(:lv => 0, :lvt => 1, :lorenz => 2, :lindiag => 3, :fallmass => 4, :mlp => 5, :bruss => 6)

"`builtin_model(:lorenz)`, `builtin_model(:mlp; dims = (2, 128, 4096))`, `builtin_model(:bruss; dims = (32,))`"
function builtin_model(name::Symbol; dims = ())
    d = ntuple(i -> i <= length(dims) ? Int32(dims[i]) : Int32(0), 4)
    id = Int32(BUILTIN[name])
    n, np = model_sizes(id, d)
    return DeviceModel(id, d, n, np)
end

"""
    register_model(name, n, np; f, vjp_u = nothing, vjp_p = nothing, dgdu = nothing, dgdp = nothing, mass_matrix = nothing) -> DeviceModel

`f`, `vjp_u`, `vjp_p`: BODIES of the three device functions as C text over `du`/`out`, `u`, `p`, `lam`, `t` — exactly the argument
meaning of `ODEFunction(f!; vjp, vjp_p)` (both VJPs un-negated).  `vjp_u === vjp_p === nothing` selects forward-mode dual numbers on
the device (`autojacvec = true`): declare locals of `f` as `real`.  Compiled with hiprtc at `Handle` creation; `check = true` compiles
once now (no device needed) so that source errors surface here.
"""
function register_model(name::AbstractString, n::Integer, np::Integer; f::AbstractString, vjp_u = nothing, vjp_p = nothing,
        dgdu = nothing, dgdp = nothing, mass_matrix = nothing, check_now::Bool = true)
    id = Ref{Int32}(0)
    # Julia Strings are NUL-terminated in memory: pointer(s) is a valid `const char *` while s is preserved
    cs(x) = x === nothing ? Ptr{UInt8}(C_NULL) : pointer(x)
    sname, sf = String(name), String(f)
    sv = vjp_u === nothing ? nothing : String(vjp_u); sw = vjp_p === nothing ? nothing : String(vjp_p)
    GC.@preserve sname sf sv sw begin
        check(ccall(sym(:hipadj_model_register), Cint, (Ptr{UInt8}, Int32, Int32, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ref{Int32}),
                    pointer(sname), Int32(n), Int32(np), pointer(sf), cs(sv), cs(sw), id))
    end
    if dgdu !== nothing
        su = String(dgdu); sp = dgdp === nothing ? nothing : String(dgdp)
        GC.@preserve su sp check(ccall(sym(:hipadj_model_set_cost), Cint, (Int32, Ptr{UInt8}, Ptr{UInt8}), id[], pointer(su), cs(sp)))
    end
    mass_matrix === nothing || set_mass_matrix!(id[], n, mass_matrix)
    check_now && check(ccall(sym(:hipadj_model_check), Cint, (Int32,), id[]))
    return DeviceModel(id[], (Int32(0), Int32(0), Int32(0), Int32(0)), Int(n), Int(np))
end

"""
    register_wide_model(name, n, np; f, vjp, threads = 0, lds_doubles = 0, nacc = 0, acc_first = 0) -> DeviceModel

A model for the workgroup-per-trajectory family (`hipadj_wmodel_register`, ABI 106): more than 8 states or more than 32 parameters, up to
`n = 4096`.  `f` and `vjp` are SPMD bodies run by every thread of the workgroup that owns a trajectory (`HIPADJ_W_FOR(i, count)`, `wg_sync()`,
`wg_sum(x)`, `wg_sum2(a, b, sa, sb)`, `ws[...]`; include/hipadj.h); `vjp` is the joint product of `vecjacobian!` (src/derivative_wrappers.jl:256-267): `dlam` and, under
`if (WP)`, the parameter gradient (`gp[j] += w * ...` for entries owned by one thread, `acc[q] += w * ...` for the `nacc` parameters
`acc_first + q` fed by every component).  Fixed-step RK4 (loss times on the step grid) and adaptive Tsit5 (arbitrary loss times, per-trajectory step control):
the four sensealgs and GaussKronrodAdjoint on both, the built-in continuous costs.
`dense_chain_bodies(widths; input_power)` below writes the two bodies for `Lux.Chain(x -> x.^k, Dense(..., tanh), ..., Dense(...))`.
"""
function register_wide_model(name::AbstractString, n::Integer, np::Integer; f::AbstractString, vjp::AbstractString, threads::Integer = 0,
        lds_doubles::Integer = 0, nacc::Integer = 0, acc_first::Integer = 0, check_now::Bool = true)
    id = Ref{Int32}(0)
    sname, sf, sv = String(name), String(f), String(vjp)
    GC.@preserve sname sf sv begin
        check(ccall(sym(:hipadj_wmodel_register), Cint, (Ptr{UInt8}, Int32, Int32, Int32, Int32, Int32, Int32, Ptr{UInt8}, Ptr{UInt8}, Ref{Int32}),
                    pointer(sname), Int32(n), Int32(np), Int32(threads), Int32(lds_doubles), Int32(nacc), Int32(acc_first), pointer(sf), pointer(sv), id))
    end
    check_now && check(ccall(sym(:hipadj_model_check), Cint, (Int32,), id[]))
    return DeviceModel(id[], (Int32(0), Int32(0), Int32(0), Int32(0)), Int(n), Int(np))
end

"""
    set_wide_cost!(m::DeviceModel, body)

The continuous cost of a wide model (`hipadj_wmodel_set_cost`, ABI 107): one SPMD body that adds `dg/du` into `dlam` and, under `if (WP)`, `w * dg/dp` into `gp` / `acc`
— `dgdu_continuous` / `dgdp_continuous` of `adjoint_sensitivities` for a model of the workgroup-per-trajectory family.  `nothing` removes it.
"""
function set_wide_cost!(m::DeviceModel, body::Union{AbstractString, Nothing})
    if body === nothing
        check(ccall(sym(:hipadj_wmodel_set_cost), Cint, (Int32, Ptr{UInt8}), m.id, C_NULL))
    else
        sb = String(body)
        GC.@preserve sb check(ccall(sym(:hipadj_wmodel_set_cost), Cint, (Int32, Ptr{UInt8}), m.id, pointer(sb)))
    end
    return m
end

"""
    declare_dense_chain!(m::DeviceModel, widths; input_power = 1)

Tells the library that wide model `m` IS the tanh chain `widths` (`hipadj_wmodel_declare_dense_chain`, ABI 109): `hipadj_create` then selects the kernel family itself — a
chain `2 -> H -> H -> 2`, `H in (32, 64, 128)`, on fixed-step RK4 with shared weights and a multiple of 16 trajectories runs on the FP64-MFMA family (the H x H contractions on
the matrix cores), everything else on the workgroup-per-trajectory family the bodies were written for.  `widths = nothing` withdraws the declaration.
"""
function declare_dense_chain!(m::DeviceModel, widths; input_power::Integer = 1)
    if widths === nothing
        check(ccall(sym(:hipadj_wmodel_declare_dense_chain), Cint, (Int32, Ptr{Int32}, Int32, Int32, Int32), m.id, C_NULL, Int32(0), ACT_TANH, Int32(1)))
    else
        w = collect(Int32, widths)
        GC.@preserve w check(ccall(sym(:hipadj_wmodel_declare_dense_chain), Cint, (Int32, Ptr{Int32}, Int32, Int32, Int32), m.id, pointer(w), Int32(length(w)), ACT_TANH, Int32(input_power)))
    end
    return m
end

"""
    dense_chain_model(name, widths; input_power = 1, threads = 0) -> DeviceModel

`Lux.Chain(x -> x.^input_power, Dense(w1, w2, tanh), ..., Dense(wL, wL+1))` as a device model in one call: the bodies of `dense_chain_bodies`, registered with
`register_wide_model` and declared with `declare_dense_chain!` — the library picks the kernel family (what `WideDeviceFunction.dense_chain` of the Python host does).
"""
function dense_chain_model(name::AbstractString, widths; input_power::Integer = 1, threads::Integer = 0, check_now::Bool = false)
    f, vjp, np, nw = dense_chain_bodies(widths; input_power = input_power)
    m = register_wide_model(name, first(widths), np; f = f, vjp = vjp, threads = threads, lds_doubles = nw, check_now = check_now)
    return declare_dense_chain!(m, widths; input_power = input_power)
end

"""
    dense_chain_bodies(widths; input_power = 1) -> (f, vjp, np, lds_doubles)

The SPMD bodies of `u' = Chain(x -> x.^input_power, Dense(w1, w2, tanh), ..., Dense(w_{L}, w_{L+1}))(u)` with the parameters in Lux's flattening
order (per layer: weight `[out x in]` column-major, then bias).  Lanes run over the outputs of a layer; a contraction with at most 8 outputs over at
least 16 inputs runs lanes-over-inputs with one `wg_sum` (wavefront shuffles) per output; a chain with ONE hidden layer and at most 8 inputs keeps every hidden unit
in its thread's registers (no LDS scratch, no barrier).  Mirrors `WideDeviceFunction.dense_chain` of the Python host.
"""
function dense_chain_bodies(widths; input_power::Integer = 1)
    w = collect(Int, widths); L = length(w) - 1
    (L >= 1 && w[1] == w[end]) || error("widths = (n, ..., n) with at least one layer")
    Woff = zeros(Int, L); Boff = zeros(Int, L); off = 0
    for l in 1:L
        Woff[l] = off; off += w[l + 1] * w[l]; Boff[l] = off; off += w[l + 1]
    end
    A = zeros(Int, L); for l in 2:L; A[l] = A[l - 1] + w[l - 1]; end          # activations a_0 .. a_{L-1} in ws (a_{l-1} at A[l])
    G = zeros(Int, L); o = A[L] + w[L]
    for l in L:-1:1; G[l] = o; o += w[l]; end                                   # backward vectors g_{l-1} at G[l]
    inp = join(fill("u[i]", input_power), " * ")
    dinp(j) = input_power == 1 ? "1.0" : string(Float64(input_power), " * ", join(fill("u[$j]", input_power - 1), " * "))
    function matvec(out, nout, nin, coef, vec; bias = nothing, post = s -> s)
        if nout <= 8 && nin >= 16
            return [string("{ double part = 0.0; HIPADJ_W_FOR(j, $nin) part += ", coef(string(i), "j"), " * ", vec("j"), "; const double s = ",
                           bias === nothing ? "" : string(bias(string(i)), " + "), "wg_sum(part); if (tid == 0) ", out(string(i)), " = ", post("s", string(i)), "; }") for i in 0:(nout - 1)]
        end
        return [string("HIPADJ_W_FOR(i, $nout) { double s = ", bias === nothing ? "0.0" : bias("i"), "; for (int j = 0; j < $nin; ++j) s += ", coef("i", "j"), " * ", vec("j"),
                       "; ", out("i"), " = ", post("s", "i"), "; }")]
    end
    if L == 2 && w[1] <= 8
        # ONE hidden layer between few inputs / outputs (the published 2-50-2 net): a hidden unit's activation, back-propagated value and gradient entries stay in the
        # registers of the thread that owns it; only the d outputs / the d entries of dlam cross lanes (in pairs: wg_sum2).  No LDS scratch, no barrier.  Every multiply-add
        # is an explicit fma(): the dense and the checkpointed sweep are compared bit for bit and must not depend on how each kernel's compile contracted a sum of products.
        d, H = w[1], w[2]
        W1, B1, W2, B2 = Woff[1], Boff[1], Woff[2], Boff[2]
        xs = [string("const double x$k = ", join(fill("u[$k]", input_power), " * "), ";") for k in 0:(d - 1)]
        hid = string("double s = p[$B1 + i]; ", join(["s = fma(p[$W1 + i + $(k * H)], x$k, s);" for k in 0:(d - 1)], " "), " const double h = tanh(s);")
        fb = vcat(xs, [join(["double part$m = 0.0;" for m in 0:(d - 1)], " "),
                       string("HIPADJ_W_FOR(i, $H) { $hid ", join(["part$m = fma(p[$W2 + $m + i * $d], h, part$m);" for m in 0:(d - 1)], " "), " }")],
                  [m + 1 < d ? "{ double s$m, s$(m + 1); wg_sum2(part$m, part$(m + 1), s$m, s$(m + 1)); if (tid >= $m && tid < $(m + 2)) du[tid] = p[$B2 + tid] + (tid == $m ? s$m : s$(m + 1)); }" :
                                "{ const double s = p[$B2 + $m] + wg_sum(part$m); if (tid == $m) du[$m] = s; }" for m in 0:2:(d - 1)])
        vb = vcat(xs, [join(["const double l$m = lam[$m];" for m in 0:(d - 1)], " "), join(["double part$k = 0.0;" for k in 0:(d - 1)], " "),
                       "HIPADJ_W_FOR(i, $H) { $hid",
                       string("  double gh = p[$W2 + 0 + i * $d] * l0; ", join(["gh = fma(p[$W2 + $m + i * $d], l$m, gh);" for m in 1:(d - 1)], " ")),
                       string("  if (WP) { const double wh = w * h; ", join(["gp[$W2 + $m + i * $d] = fma(l$m, wh, gp[$W2 + $m + i * $d]);" for m in 0:(d - 1)], " "), " }"),
                       "  gh *= 1.0 - h * h;",
                       string("  if (WP) { const double wg = w * gh; ", join(["gp[$W1 + i + $(k * H)] = fma(wg, x$k, gp[$W1 + i + $(k * H)]);" for k in 0:(d - 1)], " "), " gp[$B1 + i] += wg; }"),
                       string("  ", join(["part$k = fma(p[$W1 + i + $(k * H)], gh, part$k);" for k in 0:(d - 1)], " "), " }"),
                       "if (WP) { HIPADJ_W_FOR(m, $d) gp[$B2 + m] = fma(w, lam[m], gp[$B2 + m]); }"],
                  [k + 1 < d ? "{ double s$k, s$(k + 1); wg_sum2(part$k, part$(k + 1), s$k, s$(k + 1)); if (tid >= $k && tid < $(k + 2)) dlam[tid] = (tid == $k ? s$k : s$(k + 1)) * $(dinp("tid")); }" :
                                "{ const double s = wg_sum(part$k); if (tid == $k) dlam[$k] = s * $(dinp(k)); }" for k in 0:2:(d - 1)])
        return join(fb, "\n"), join(vb, "\n"), off, 1
    end
    fwd = String["HIPADJ_W_FOR(i, $(w[1])) ws[$(A[1]) + i] = $inp;", "wg_sync();"]
    for l in 1:(L - 1)
        append!(fwd, matvec(i -> "ws[$(A[l + 1]) + $i]", w[l + 1], w[l], (i, j) -> "p[$(Woff[l]) + $i + $j * $(w[l + 1])]", j -> "ws[$(A[l]) + $j]";
                            bias = i -> "p[$(Boff[l]) + $i]", post = (s, i) -> "tanh($s)"))
        push!(fwd, "wg_sync();")
    end
    fb = vcat(fwd, matvec(i -> "du[$i]", w[L + 1], w[L], (i, j) -> "p[$(Woff[L]) + $i + $j * $(w[L + 1])]", j -> "ws[$(A[L]) + $j]"; bias = i -> "p[$(Boff[L]) + $i]", post = (s, i) -> s))
    vb = copy(fwd)
    for l in L:-1:1
        gl = l == L ? "lam" : "(ws + $(G[l + 1]))"
        push!(vb, "if (WP) { HIPADJ_W_FOR(e, $(w[l + 1] * w[l])) gp[$(Woff[l]) + e] += w * $gl[e % $(w[l + 1])] * ws[$(A[l]) + e / $(w[l + 1])]; HIPADJ_W_FOR(i, $(w[l + 1])) gp[$(Boff[l]) + i] += w * $gl[i]; }")
        append!(vb, matvec(l == 1 ? (j -> "dlam[$j]") : (j -> "ws[$(G[l]) + $j]"), w[l], w[l + 1], (j, i) -> "p[$(Woff[l]) + $i + $j * $(w[l + 1])]", i -> "$gl[$i]";
                           post = (s, j) -> l == 1 ? "$s * $(dinp(j))" : "$s * (1.0 - ws[$(A[l]) + $j] * ws[$(A[l]) + $j])"))
        l > 1 && push!(vb, "wg_sync();")
    end
    return join(fb, "\n"), join(vb, "\n"), off, o
end

"""
    set_mass_matrix!(model, M)          # M === nothing (or I) removes it

`ODEFunction(f!; mass_matrix = M)` for a registered model: `M u' = f` with a constant NON-SINGULAR matrix (test/Core3/adjoint.jl:1315-1376).
The ABI takes M row-major, Julia stores column-major, so the transpose is materialised once here.  `du0` keeps the reference's meaning
(lam(t0) of `M' lam' = -J' lam`, src/sensitivity_interface.jl:500).  A singular `M` of the semi-explicit form `[Md 0; 0 0]` (zero rows that are
also zero columns: the algebraic variables of src/adjoint_common.jl:116-135) makes the model a DAE: `STEPPER_ROSENBROCK23_ADAPTIVE` integrates it in
mass-matrix form from a consistent state (test/Core3/adjoint.jl:1434-1530); the explicit steppers refuse such a model at `Handle` creation.
Any other singular `M` is refused here.
"""
function set_mass_matrix!(id::Integer, n::Integer, M)
    if M === nothing
        check(ccall(sym(:hipadj_model_set_mass_matrix), Cint, (Int32, Ptr{Float64}), Int32(id), Ptr{Float64}(C_NULL)))
        return nothing
    end
    size(M) == (n, n) || error("mass_matrix must be $n x $n")
    Mt = Matrix{Float64}(permutedims(M))           # column-major M' == row-major M
    GC.@preserve Mt check(ccall(sym(:hipadj_model_set_mass_matrix), Cint, (Int32, Ptr{Float64}), Int32(id), pointer(Mt)))
    return nothing
end
set_mass_matrix!(m::DeviceModel, M) = set_mass_matrix!(m.id, m.n, M)

"""
    set_affect!(model, body)            # body === nothing removes it

The `affect!` of a `DiscreteCallback` at preset times as device text (test/Callbacks1/discrete_callbacks.jl:260-330): `body` edits `un` — which starts
as a copy of `u` — and / or `pn` (a copy of `p`) from `u`, `p`, `t` (locals `real`; the reverse callback's Jacobian products are generated by dual numbers).  An event problem is a chain
of ordinary handles, one per span between consecutive event times: `affect_apply` maps the end state of a piece to the start state of the next,
`affect_vjp` maps `du0` of the upper piece to the extra cotangent at the end of the lower one and returns the parameter term
(src/callback_tracking.jl:330-452 for a DiscreteCallback).  The Python host mirror (`scimlsensitivity.jl_amd/events.py`) is the executed reference of
that composition; a `ContinuousCallback` is a property of the model instead: `set_continuous_callback!`.
"""
function set_affect!(m::DeviceModel, body)
    s = body === nothing ? nothing : String(body)
    GC.@preserve s check(ccall(sym(:hipadj_model_set_affect), Cint, (Int32, Ptr{UInt8}), m.id, s === nothing ? Ptr{UInt8}(C_NULL) : pointer(s)))
    return m
end
"""
    set_continuous_callback!(model, condition, affect = nothing; max_events = 0)        # condition === nothing removes it

`ContinuousCallback(condition, affect!; save_positions = (false, false))` as device text (`hipadj_model_set_continuous_callback`; src/callback_tracking.jl:232-479,
test/Callbacks2/continuous_callbacks.jl): `condition` assigns `c` from `u`, `p`, `t` — the event is its zero crossing, either direction —, `affect` edits `un` (a copy of `u`).
The bouncing ball: `set_continuous_callback!(m, "c = u[0];", "un[1] = -p[1] * u[1];")`.  Every later handle on the model with `stepper = STEPPER_TSIT5_ADAPTIVE` or
`STEPPER_ROSENBROCK23_ADAPTIVE` locates the events of each trajectory on the dense output; every sensealg differentiates through them, event times included.  `event_counts(handle)` returns the events per trajectory of the last forward solve.
"""
function set_continuous_callback!(m::DeviceModel, condition, affect = nothing; max_events::Integer = 0, ncond::Integer = 1, direction::Integer = 0)
    c = condition === nothing ? nothing : String(condition); a = affect === nothing ? nothing : String(affect)
    pc = c === nothing ? Ptr{UInt8}(C_NULL) : pointer(c); pa = a === nothing ? Ptr{UInt8}(C_NULL) : pointer(a)
    if ncond > 1        # VectorContinuousCallback(condition, affect!, ncond): `out[k]` in the condition body, `idx` in the affect body
        GC.@preserve c a check(ccall(sym(:hipadj_model_set_vector_continuous_callback), Cint, (Int32, Int32, Ptr{UInt8}, Ptr{UInt8}, Int32), m.id, Int32(ncond), pc, pa, Int32(max_events)))
    else
        GC.@preserve c a check(ccall(sym(:hipadj_model_set_continuous_callback), Cint, (Int32, Ptr{UInt8}, Ptr{UInt8}, Int32), m.id, pc, pa, Int32(max_events)))
    end
    # direction = +1: only upcrossings fire (ContinuousCallback(condition, affect!, nothing)), -1: only downcrossings
    direction != 0 && c !== nothing && check(ccall(sym(:hipadj_model_set_callback_direction), Cint, (Int32, Int32), m.id, Int32(direction)))
    return m
end
function affect_apply(m::DeviceModel, u::Matrix{Float64}, p::Union{Vector{Float64}, Matrix{Float64}}, t::Real; device::Integer = 0)
    out = similar(u); pout = Matrix{Float64}(undef, m.np, size(u, 2))     # (n, N) / (np, N) column-major == the ABI's [N][n] / [N][np]
    check(ccall(sym(:hipadj_affect_apply), Cint, (Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ptr{Float64}),
                m.id, Int32(device), size(u, 2), u, p, Int32(p isa Vector), Float64(t), out, pout))
    return out, pout
end
"`(lam_out, gp_out)` of the reverse callback for the map `(u, p) -> (un, pn)`; `gp` `(np, N)`: gradient w.r.t. the parameters after the event of everything later in time"
function affect_vjp(m::DeviceModel, u::Matrix{Float64}, p::Union{Vector{Float64}, Matrix{Float64}}, t::Real, lam::Matrix{Float64}, gp::Matrix{Float64}; device::Integer = 0)
    lam_out = similar(lam); gp_out = similar(gp)
    check(ccall(sym(:hipadj_affect_vjp), Cint, (Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.id, Int32(device), size(u, 2), u, p, Int32(p isa Vector), Float64(t), lam, gp, lam_out, gp_out))
    return lam_out, gp_out
end

# ---------------------------------------------------------------------------------------------------------------------
# the sensealg the extension dispatches on (seam B1 of SURVEY.md §8b)
# ---------------------------------------------------------------------------------------------------------------------
"""
    HIPBatchedAdjoint(inner = nothing; model, device = 0, devices = nothing, time_segments = 0, max_steps = 0)

`inner`: the reference algorithm whose semantics are wanted (`InterpolatingAdjoint()`, `BacksolveAdjoint(checkpointing = true)`,
`GaussAdjoint()`, `GaussKronrodAdjoint()`, `QuadratureAdjoint(abstol, reltol)`; read by the extension).  The state of the problem is a
MATRIX whose columns are independent trajectories of `model` — the documented batching pattern of the reference
(docs/src/tutorials/data_parallel.md:11-75, test/Core5/size_handling_adjoint.jl:37-70); `p` is a vector shared by all columns or an
`np x N` matrix.
`devices = :all` (or a list of ordinals): ONE `solve` call uses every GPU of the node — the columns are cut into contiguous ranges, one per device, inside the library
(`hipadj_config.device_ids`, ABI 108): the single-process counterpart of the reference's `EnsembleDistributed` pattern (docs/src/tutorials/data_parallel.md:77-136,
test/Core4/distributed.jl:15-41), with no worker processes and no id exchange.
"""
struct HIPBatchedAdjoint{A} <: SciMLBase.AbstractAdjointSensitivityAlgorithm{0, false, Val{:central}}
    inner::A
    model::DeviceModel
    device::Int32
    time_segments::Int32
    max_steps::Int32
    devices::Union{Nothing, Symbol, Vector{Int32}}
end
HIPBatchedAdjoint(inner; model::DeviceModel, device::Integer = 0, devices = nothing, time_segments::Integer = 0, max_steps::Integer = 0) =
    HIPBatchedAdjoint{typeof(inner)}(inner, model, Int32(device), Int32(time_segments), Int32(max_steps),
                                     devices === nothing || devices isa Symbol ? devices : collect(Int32, devices))

# ---------------------------------------------------------------------------------------------------------------------
# handle + the three calls
# ---------------------------------------------------------------------------------------------------------------------
mutable struct Handle
    ptr::Ptr{Cvoid}
    n::Int
    np::Int
    N::Int
    M::Int
    p_shared::Bool
    function Handle(cfg::HipadjConfig, n, np, keep...)
        check_layout()
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = GC.@preserve keep ccall(sym(:hipadj_create), Cint, (Ref{HipadjConfig}, Ref{Ptr{Cvoid}}), Ref(cfg), out)
        check(rc)                                   # message of a failed create: hipadj_last_error(NULL)
        h = new(out[], n, np, Int(cfg.ntraj), Int(cfg.nsave), cfg.p_shared != 0)
        finalizer(destroy!, h)
        return h
    end
end

function destroy!(h::Handle)
    if h.ptr != C_NULL
        ccall(sym(:hipadj_destroy), Cint, (Ptr{Cvoid},), h.ptr)
        h.ptr = C_NULL
    end
    return nothing
end

"""
    Handle(model; alg, stepper, N, tspan, dt, ts, loss_kind = LOSS_COTANGENT, loss_shift = 0.0, checkpointing = false,
           checkpoints = nothing, quad_abstol = 1e-6, quad_reltol = 1e-3, no_start = false, p_shared = true, device = 0,
           time_segments = 0, cont_cost = 0, max_steps = 0, abstol = 1e-6, reltol = 1e-3, loss_scale = 0.0, devices = nothing, reference_literal = false, family = FAMILY_AUTO)

`devices = [0, 1, ..., 7]` (or `:all`): ONE handle over several devices — the ensemble is cut into contiguous trajectory ranges, one per device, and the host-pointer calls
scatter / gather / sum over them (`hipadj_config.device_ids`, ABI 108): what a single `solve` call needs to use a whole node.
`family = FAMILY_AS_REGISTERED` keeps a declared dense chain (`declare_dense_chain!`) on the workgroup-per-trajectory family; the default lets the library choose (ABI 109).
`loss_kind = LOSS_LSQ_DATA` with `loss_scale` (and `set_loss_data!`) keeps `sum(abs2, sol .- data)` on the device; `LOSS_MODEL` runs the model's discrete-loss bodies.
"""
function Handle(model::DeviceModel; alg::Int32, stepper::Int32, N::Integer, tspan, dt::Real, ts::Vector{Float64},
        loss_kind::Int32 = LOSS_COTANGENT, loss_shift::Real = 0.0, checkpointing::Bool = false, checkpoints = nothing,
        quad_abstol::Real = 1e-6, quad_reltol::Real = 1e-3, no_start::Bool = false, p_shared::Bool = true, device::Integer = 0,
        time_segments::Integer = 0, cont_cost::Integer = 0, max_steps::Integer = 0, abstol::Real = 1e-6, reltol::Real = 1e-3,
        loss_scale::Real = 0.0, devices = nothing, reference_literal::Bool = false, family::Integer = FAMILY_AUTO)
    cks = checkpoints === nothing ? Float64[] : sort(collect(Float64, checkpoints))
    devs = devices === nothing ? Int32[] : (devices === :all ? collect(Int32, 0:(device_count() - 1)) : collect(Int32, devices))
    cfg = HipadjConfig(UInt32(sizeof(HipadjConfig)), model.id, alg, stepper, model.dims, Int64(N),
        Float64(tspan[1]), Float64(tspan[2]), Float64(dt), Int32(length(ts)), isempty(ts) ? Ptr{Float64}(C_NULL) : pointer(ts),
        loss_kind, Float64(loss_shift), Int32(checkpointing), Int32(0), Float64(quad_abstol), Float64(quad_reltol),
        Int32(no_start), Int32(p_shared), Int32(device), Int32(time_segments), Int32(cont_cost), Int32(max_steps),
        Float64(abstol), Float64(reltol), Int32(length(cks)), isempty(cks) ? Ptr{Float64}(C_NULL) : pointer(cks),
        Float64(loss_scale), Int32(length(devs)), isempty(devs) ? Ptr{Int32}(C_NULL) : pointer(devs), Int32(reference_literal), Int32(family))
    return Handle(cfg, model.n, model.np, ts, cks, devs)   # ts / cks / devs are copied by hipadj_create; kept alive across the call
end

"Number of HIP devices the library sees (`hipadj_device_count`): what `devices = :all` expands to."
device_count() = Int(ccall(sym(:hipadj_device_count), Cint, ()))

"""
    set_loss_data!(h, data)

The data block `(n, M, N)` of a device-resident loss (`hipadj_set_loss_data`): `loss_kind = LOSS_LSQ_DATA` differentiates `loss_scale / 2 * sum(abs2, sol .- data)` inside
the reverse kernels — `adjoint!(h, nothing)` then takes no cotangents, and nothing crosses the host link between the forward and the reverse pass — and the
discrete-loss bodies of a model (`set_discrete_loss!`, `LOSS_MODEL`) see it as `d`.
"""
function set_loss_data!(h::Handle, data::Array{Float64, 3})
    size(data) == (h.n, h.M, h.N) || throw(DimensionMismatch("data must be ($(h.n), $(h.M), $(h.N)), got $(size(data))"))
    check(ccall(sym(:hipadj_set_loss_data), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, data), h.ptr)
    return h
end

"`loss_value(h, out)`: the loss of a device-resident discrete loss summed over the ensemble, evaluated on the device from `out = forward!(...)` (`hipadj_loss_value`)."
function loss_value(h::Handle, out::Array{Float64, 3})
    size(out) == (h.n, h.M, h.N) || throw(DimensionMismatch("out must be ($(h.n), $(h.M), $(h.N)), got $(size(out))"))
    l = Ref{Float64}(0.0)
    check(ccall(sym(:hipadj_loss_value), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), h.ptr, out, l), h.ptr)
    return l[]
end

"""
    set_discrete_loss!(model; dgdu = nothing, dgdp = nothing, l = nothing)

`dgdu_discrete` / `dgdp_discrete` of `adjoint_sensitivities` as device text (`hipadj_model_set_discrete_loss[_function]`): bodies writing `out` from `u`, `p`, `t`,
`i` (0-based index of the loss time: the reference's `i - 1`) and `d` (the data column of this trajectory and time), or the loss itself (`l = "l = u[0]*u[0] + p[0];"`,
gradients by dual numbers).  Selected per handle with `loss_kind = LOSS_MODEL`.
"""
function set_discrete_loss!(model::DeviceModel; dgdu = nothing, dgdp = nothing, l = nothing)
    if l !== nothing
        check(ccall(sym(:hipadj_model_set_discrete_loss_function), Cint, (Int32, Cstring), model.id, l))
    else
        check(ccall(sym(:hipadj_model_set_discrete_loss), Cint, (Int32, Cstring, Cstring), model.id, dgdu === nothing ? C_NULL : dgdu, dgdp === nothing ? C_NULL : dgdp))
    end
    return model
end

"`out = forward!(h, u0, p)`: u0 `(n, N)`, p `(np,)` or `(np, N)`; returns `out` `(n, M, N)` = sol(ts) of every trajectory."
"events per trajectory of the last forward solve of a handle whose model carries a ContinuousCallback (`hipadj_event_counts`)"
function event_counts(h::Handle)
    counts = Vector{Int32}(undef, h.N)
    check(ccall(sym(:hipadj_event_counts), Cint, (Ptr{Cvoid}, Ptr{Int32}), h.ptr, counts), h.ptr)
    return counts
end
"""
    event_states(h; max_events = 64) -> (t, ul, ur)        set_event_cotangents!(h, dl, dr)

`save_positions = (true, true)`: the event times `(max_events, N)` and the states just before / after the affect `(n, max_events, N)` of the last forward solve
(`hipadj_event_states`), and the cotangents of a loss on them for the following `adjoint!` calls (`hipadj_set_event_cotangents`; `nothing` = zero).
"""
function event_states(h::Handle; max_events::Integer = 64)
    t = zeros(Float64, max_events, h.N); ul = zeros(Float64, h.n, max_events, h.N); ur = similar(ul)     # column-major == the ABI's [N][max_events][n]
    check(ccall(sym(:hipadj_event_states), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h.ptr, t, ul, ur), h.ptr)
    return t, ul, ur
end
"which component of a VectorContinuousCallback fired at each event, `(max_events, N)`: 0 for a scalar condition, + 256 = the event terminated the trajectory, -1 beyond the count (`hipadj_event_components`)"
function event_components(h::Handle; max_events::Integer = 64)
    idx = Matrix{Int32}(undef, max_events, h.N)
    check(ccall(sym(:hipadj_event_components), Cint, (Ptr{Cvoid}, Ptr{Int32}), h.ptr, idx), h.ptr)
    return idx
end
function set_event_cotangents!(h::Handle, dl::Union{Nothing, Array{Float64, 3}}, dr::Union{Nothing, Array{Float64, 3}})
    pl = dl === nothing ? Ptr{Float64}(C_NULL) : pointer(dl); pr = dr === nothing ? Ptr{Float64}(C_NULL) : pointer(dr)
    GC.@preserve dl dr check(ccall(sym(:hipadj_set_event_cotangents), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ptr, pl, pr), h.ptr)
    return h
end
function forward!(h::Handle, u0::Matrix{Float64}, p::VecOrMat{Float64}; want_out::Bool = true)
    size(u0) == (h.n, h.N) || throw(DimensionMismatch("u0 must be ($(h.n), $(h.N)), got $(size(u0))"))
    (h.p_shared ? size(p) == (h.np,) : size(p) == (h.np, h.N)) || throw(DimensionMismatch("p has size $(size(p))"))
    out = want_out && h.M > 0 ? Array{Float64}(undef, h.n, h.M, h.N) : nothing
    check(ccall(sym(:hipadj_forward), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h.ptr, u0, p, out === nothing ? Ptr{Float64}(C_NULL) : pointer(out)), h.ptr)
    return out
end

"`(du0, dp) = adjoint!(h, Δ)`: Δ `(n, M, N)` cotangents of `out` (or `nothing` for the fused LSQ loss); du0 `(n, N)`, dp `(np,)` summed over the ensemble or `(np, N)`."
function adjoint!(h::Handle, Δ::Union{Nothing, Array{Float64, 3}})
    Δ === nothing || size(Δ) == (h.n, h.M, h.N) || throw(DimensionMismatch("Δ must be ($(h.n), $(h.M), $(h.N)), got $(size(Δ))"))
    du0 = Matrix{Float64}(undef, h.n, h.N)
    dp = h.p_shared ? Vector{Float64}(undef, h.np) : Matrix{Float64}(undef, h.np, h.N)
    check(ccall(sym(:hipadj_adjoint), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h.ptr, Δ === nothing ? Ptr{Float64}(C_NULL) : pointer(Δ), du0, dp), h.ptr)
    return du0, dp
end

# ---------------------------------------------------------------------------------------------------------------------
# the direct API (seam B2): forward solution object + the generic the SciMLSensitivity extension adds its method to
# ---------------------------------------------------------------------------------------------------------------------
"Forward solution of the device engine: what `adjoint_sensitivities` consumes in place of an ODESolution."
struct HIPAdjSolution
    handle::Handle
    u::Array{Float64, 3}      # (n, M, N) = sol(ts) of every trajectory
    t::Vector{Float64}
    p
end

"`hip_solve(prob, alg, sensealg::HIPBatchedAdjoint; u0, p, saveat, dt, ...) -> HIPAdjSolution` (method: ext/SciMLSensitivityHIPAdjExt.jl)"
function hip_solve end

"Stack an `EnsembleProblem`'s trajectories into the matrix state: `(u0 (n, N), p (np,) or (np, N))` from `prob_func(prob, i, 1)`."
function ensemble_u0_p(ens::SciMLBase.EnsembleProblem, trajectories::Integer)
    probs = [ens.prob_func(ens.prob, i, 1) for i in 1:trajectories]
    u0 = reduce(hcat, (vec(pr.u0) for pr in probs))
    ps = [pr.p for pr in probs]
    p = all(q -> q == ps[1], ps) ? collect(Float64, ps[1]) : reduce(hcat, (vec(q) for q in ps))
    return u0, p
end

# sharded ensembles: one process (Distributed worker) per GPU; the library all-reduces dp over RCCL in-stream (include/hipadj.h, hipadj_comm_*)
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall(sym(:hipadj_comm_unique_id), Cint, (Ptr{UInt8},), id))
    return id
end
comm_init_rank!(h::Handle, id::Vector{UInt8}, nranks::Integer, rank::Integer) =
    check(ccall(sym(:hipadj_comm_init_rank), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), h.ptr, id, nranks, rank), h.ptr)

end # module
