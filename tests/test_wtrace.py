"""wtrace.py: automatic joint VJP for wide runtime models (the reference's AD-generated vecjacobian!, src/derivative_wrappers.jl:649-1145).

CPU part: the emitted SPMD bodies are compiled with g++ under ONE-thread semantics (HIPADJ_W_FOR = a plain loop, wg_sum(x) = x, wg_sync() = nothing) and f / the
joint VJP are compared with the oracle's hand-derived models (orc_model_f / orc_model_vjp) at 1e-13; the same text is compiled for gfx950 by hiprtc (no device).
GPU part: traced models against the hand-written emitters of problems.py (1e-12) and against the oracle's gradients (rtol 1e-6), all four sensealgs."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle as O


def ring(u, p, t, ops):                      # du_i = p_i (u_{i+1} - u_i) + p_n sin(u_{i-1})      (oracle: RING)
    n = u.length
    return p[0:n] * (ops.roll(u, -1) - u) + p[n] * ops.sin(ops.roll(u, 1))


def idxaff(R, Cc):                           # df[i, j] = p1 i + p2 j on an R x Cc state, column-major   (test/Core5/size_handling_adjoint.jl:41-48; oracle: IDXAFF)
    def f(u, p, t, ops):
        c = np.arange(R * Cc)
        return p[0] * ops.const(c % R + 1.0) + p[1] * ops.const(c // R + 1.0)
    return f


def denselin(n):                             # u' = A u, A = reshape(p, n, n)                              (oracle: DENSELIN)
    return lambda u, p, t, ops: ops.matvec(p[0:n * n], n, u)


def mlp1(d, H):                              # Chain(x -> x.^3, Dense(d, H, tanh), Dense(H, d)), docs/src/Benchmark.md:62   (oracle: MLP1)
    def f(u, p, t, ops):
        h = ops.tanh(ops.matvec(p[0:H * d], H, u ** 3) + p[H * d:H * d + H])
        o = H * d + H
        return ops.matvec(p[o:o + d * H], d, h) + p[o + d * H:o + d * H + d]
    return f


def rxn(n):                                  # a reaction chain with a conserved-total feedback: du = k_i (u_{i-1} - u_i) u_i / (1 + u_i^2) + p_n sum(u) exp(-u_i)   (no oracle model: finite differences)
    def f(u, p, t, ops):
        return p[0:n] * (ops.roll(u, 1) - u) * u / (1.0 + u * u) + (p[n] * ops.sum(u)) * ops.exp(-u) + p[n + 1] * t
    return f


CASES = {
    "ring64": (ring, 64, 65, "RING", (64, 0, 0, 0)),
    "ring5": (ring, 5, 6, "RING", (5, 0, 0, 0)),
    "idxaff_30x50": (idxaff(30, 50), 1500, 2, "IDXAFF", (30, 50, 0, 0)),
    "denselin12": (denselin(12), 12, 144, "DENSELIN", (12, 0, 0, 0)),
    "mlp1_2_50": (mlp1(2, 50), 2, 252, "MLP1", (2, 50, 0, 0)),
    "rxn40": (rxn(40), 40, 42, None, None),
}

HARNESS = r'''
#include <cmath>
#define HIPADJ_W_FOR(i, n) for (int i = tid; i < (n); i += T)
#define wg_sync() ((void)0)
#define wg_sum(x) (x)
static const int T = 1;
extern "C" void model_f(double* du, const double* u, const double* p, double t, double* ws) {
    const int tid = 0; (void)ws; (void)t; (void)p; (void)u;
%(f)s
}
template <bool WP> static void vjp_t(double* dlam, double* gp, double (&acc)[%(na)d], double w, const double* lam, const double* u, const double* p, double t, double* ws, int tid) {
    (void)gp; (void)acc; (void)w; (void)u; (void)p; (void)t; (void)ws; (void)tid;
%(vjp)s
}
extern "C" void model_vjp(int wp, double* dlam, double* gp, double* acc, double w, const double* lam, const double* u, const double* p, double t, double* ws) {
    double a[%(na)d] = {0};
    if (wp) vjp_t<true>(dlam, gp, a, w, lam, u, p, t, ws, 0); else vjp_t<false>(dlam, gp, a, w, lam, u, p, t, ws, 0);
    for (int q = 0; q < %(na)d; ++q) acc[q] = a[q];
}
'''


def host_model(fn, n, npar):
    from scimlsensitivity_jl_amd import wtrace
    fb, vb, nw, nacc, a0 = wtrace.bodies(fn, n, npar)
    d = tempfile.mkdtemp(prefix="wtrace_")
    src = os.path.join(d, "m.cpp")
    open(src, "w").write(HARNESS % dict(f=fb, vjp=vb, na=max(nacc, 1)))
    so = os.path.join(d, "m.so")
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    P = C.POINTER(C.c_double)
    L.model_f.argtypes = [P, P, P, C.c_double, P]
    L.model_vjp.argtypes = [C.c_int, P, P, P, C.c_double, P, P, P, C.c_double, P]
    ptr = lambda a: a.ctypes.data_as(P)

    def f(u, p, t):
        du, ws = np.zeros(n), np.zeros(max(nw, 1))
        L.model_f(ptr(du), ptr(u), ptr(p), t, ptr(ws))
        return du

    def vjp(lam, u, p, t, w=1.0, wp=True):
        dlam, gp, acc, ws = np.full(n, np.nan), np.zeros(npar), np.zeros(max(nacc, 1)), np.zeros(max(nw, 1))
        L.model_vjp(int(wp), ptr(dlam), ptr(gp), ptr(acc), w, ptr(lam), ptr(u), ptr(p), t, ptr(ws))
        for q in range(nacc):
            gp[a0 + q] += acc[q]
        return dlam, gp
    return f, vjp


@pytest.mark.parametrize("case", list(CASES))
def test_emitted_bodies_equal_the_hand_derived_models_on_the_host(case):
    fn, n, npar, omodel, dims = CASES[case]
    f, vjp = host_model(fn, n, npar)
    rng = np.random.default_rng(3)
    u, p, lam, t = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, npar), rng.standard_normal(n), 0.37
    du = f(u, p, t)
    dlam, gp = vjp(lam, u, p, t, w=0.7)
    if omodel:
        rdu = O.model_f(omodel, u, p, t, dims)
        rdl, rgp = O.model_vjp(omodel, lam, u, p, t, dims)
        assert np.max(np.abs(du - rdu)) <= 1e-13 * max(1.0, np.max(np.abs(rdu)))
        assert np.max(np.abs(dlam - rdl)) <= 1e-13 * max(1.0, np.max(np.abs(rdl)))
        assert np.max(np.abs(gp - 0.7 * rgp)) <= 1e-13 * max(1.0, np.max(np.abs(rgp)))
    # central differences of lam . f (every model, incl. the one without an oracle counterpart)
    eps = 1e-6
    for x, g, scale in ((u, dlam, 1.0), (p, gp, 0.7)):
        for k in rng.choice(len(x), size=min(len(x), 12), replace=False):
            xp, xm = x.copy(), x.copy(); xp[k] += eps; xm[k] -= eps
            fd = (lam @ (f(xp, p, t) if x is u else f(u, xp, t)) - lam @ (f(xm, p, t) if x is u else f(u, xm, t))) / (2 * eps)
            assert abs(g[k] - scale * fd) < 1e-7 * max(1.0, abs(fd)), (case, k)
    dl2, gp2 = vjp(lam, u, p, t, w=0.7, wp=False)      # WP = false: no gradient contribution, same dlam
    assert np.array_equal(dl2, dlam) and not gp2.any()


@pytest.mark.parametrize("case", list(CASES))
def test_traced_models_compile_for_gfx950(sa, case):
    fn, n, npar, _, _ = CASES[case]
    from scimlsensitivity_jl_amd import _lib
    fun = sa.WideDeviceFunction.from_callable(f"wt_{case}", fn, n, npar)
    _lib.check_model(fun.id)


def test_untraceable_programs_are_refused(sa):
    from scimlsensitivity_jl_amd import wtrace
    with pytest.raises(TypeError):
        wtrace.bodies(lambda u, p, t, ops: u if u else u, 4, 2)
    with pytest.raises(ValueError):
        wtrace.bodies(lambda u, p, t, ops: u + p[0:3], 4, 4)
    with pytest.raises(ValueError):
        wtrace.bodies(lambda u, p, t, ops: ops.gather(u, [0, 9]), 2, 1)
    with pytest.raises(TypeError):
        wtrace.bodies(lambda u, p, t, ops: u[0:2], 4, 1)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.gpu
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
@pytest.mark.parametrize("case", ["ring64", "idxaff_30x50", "denselin12", "mlp1_2_50"])
@pytest.mark.parametrize("shared", [True, False])
def test_traced_models_on_the_device_match_the_oracle(sa, case, alg, oalg, shared):
    fn, n, npar, omodel, dims = CASES[case]
    fun = sa.WideDeviceFunction.from_callable(f"wtg_{case}", fn, n, npar) if f"wtg_{case}" not in _FUN else _FUN[f"wtg_{case}"]
    _FUN[f"wtg_{case}"] = fun
    rng = np.random.default_rng(17)
    N, T, dt = 5, 0.5, 0.01
    ts = np.linspace(0.0, T, 6)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(-0.6, 0.7, npar)
    pp = p if shared else p * (1 + 0.05 * rng.standard_normal((N, npar)))
    delta = rng.standard_normal((N, len(ts), n))
    sens = dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(checkpointing=True), gauss=sa.GaussAdjoint(),
                quadrature=sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12))[alg]
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), pp if shared else pp[0]), u0, pp), sa.RK4(), dt=dt, saveat=ts, sensealg=sens)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    ref = O.Problem(omodel, alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, checkpointing=(alg == "backsolve"), dims=dims, quad_abstol=1e-12, quad_reltol=1e-12)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(sol.u, rout) < 1e-6 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6
    sol.engine.close()


_FUN = {}


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["idxaff", "denselin", "chain"])
def test_traced_models_equal_the_hand_written_emitters(sa, which):
    """the same model through wtrace and through the hand-written emitter of problems.py: gradients agree at 1e-12 (summation orders differ)"""
    if which == "idxaff":
        a, b, n, npar = sa.WideDeviceFunction.from_callable("wth_idx", idxaff(30, 50), 1500, 2), sa.WideDeviceFunction.index_affine("wth_idx_h", 30, 50), 1500, 2
    elif which == "denselin":
        a, b, n, npar = sa.WideDeviceFunction.from_callable("wth_lin", denselin(100), 100, 10000), sa.WideDeviceFunction.dense_linear("wth_lin_h", 100), 100, 10000
    else:
        a, b, n, npar = sa.WideDeviceFunction.from_callable("wth_chain", mlp1(2, 50), 2, 252), sa.WideDeviceFunction.dense_chain("wth_chain_h", (2, 50, 2), input_power=3), 2, 252
    rng = np.random.default_rng(23)
    N, T, dt = 7, 0.4, 0.01
    ts = np.linspace(0.0, T, 5)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(-0.3, 0.3, npar) / (10.0 if which == "denselin" else 1.0)
    delta = rng.standard_normal((N, len(ts), n))
    res = []
    for fun in (a, b):
        for stepper in ("rk4", "tsit5"):
            kw = dict(dt=dt) if stepper == "rk4" else dict(abstol=1e-9, reltol=1e-9)
            salg = sa.RK4() if stepper == "rk4" else sa.Tsit5()
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sa.GaussAdjoint(), **kw)
            res.append(sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta))
            sol.engine.close()
    for k in (0, 1):
        tol = 1e-12 if k == 0 else 1e-8          # adaptive: the controller's norms are summed in another order, a step size may differ in its last bits
        assert rel(res[k][0], res[2 + k][0]) < tol and rel(res[k][1], res[2 + k][1]) < tol


def test_emitted_text_is_pinned_by_a_golden_file():
    """tests/golden/wtrace_bodies.json holds the bodies wtrace writes for two models: the text a second host-language emitter (the Julia extension's Symbolics path,
    never executed in this image) has to reproduce, and a tripwire for unintended changes of the generated code."""
    import json
    from scimlsensitivity_jl_amd import wtrace
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wtrace_bodies.json")))
    for case, g in gold.items():
        fn, n, npar, _, _ = CASES[case]
        fb, vb, nw, nacc, a0 = wtrace.bodies(fn, n, npar)
        assert (fb, vb, nw, nacc, a0) == (g["f"], g["vjp"], g["lds_doubles"], g["nacc"], g["acc_first"]), case


def test_constant_matrix_products_on_the_host():
    """ops.matvec_const (the inverse of a mass matrix at trace level): f and the joint VJP of F = A f(u) against numpy"""
    n = 7
    rng = np.random.default_rng(9)
    A = rng.standard_normal((n, n)) + 3.0 * np.eye(n)
    f, vjp = host_model(lambda u, p, t, ops: ops.matvec_const(A, ring(u, p, t, ops)), n, n + 1)
    u, p, lam = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, n + 1), rng.standard_normal(n)
    base = O.model_f("RING", u, p, 0.0, (n, 0, 0, 0))
    assert np.max(np.abs(f(u, p, 0.0) - A @ base)) < 1e-13
    dl, gp = vjp(lam, u, p, 0.0)
    rdl, rgp = O.model_vjp("RING", A.T @ lam, u, p, 0.0, (n, 0, 0, 0))
    assert np.max(np.abs(dl - rdl)) < 1e-12 and np.max(np.abs(gp - rgp)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_traced_wide_model_with_a_mass_matrix(sa, alg, oalg, stepper):
    """ODEFunction(f; mass_matrix = M) for a model beyond the lane family (test/Core3/adjoint.jl:1315-1376 restated on a 12-state ring): M^{-1} is folded into the traced
    right-hand side, the device integrates nu = M' lam, the host maps du0 = M^{-T} nu(t0); against the oracle, which carries M the reference's way."""
    n = 12
    rng = np.random.default_rng(13)
    M = np.eye(n) * 2.0 + 0.3 * rng.standard_normal((n, n))
    name = "wt_ring_mm"
    fun = _FUN.get(name) or sa.WideDeviceFunction.from_callable(name, ring, n, n + 1, mass_matrix=M)
    _FUN[name] = fun
    N, T, dt = 5, 0.6, 0.01
    ts = np.linspace(0.0, T, 5)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.2, 0.7, n + 1)
    delta = rng.standard_normal((N, len(ts), n))
    sens = dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(checkpointing=True), gauss=sa.GaussAdjoint(), quadrature=sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12))[alg]
    salg, kw, okw = (sa.RK4(), dict(dt=dt), dict(stepper="RK4", dt=dt)) if stepper == "rk4" else (sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta)
    sol.engine.close()
    with O.mass_matrix(M):
        ref = O.Problem("RING", alg=oalg, t0=0.0, t1=T, save_times=ts, checkpointing=(alg == "backsolve"), dims=(n, 0, 0, 0), quad_abstol=1e-12, quad_reltol=1e-12, **okw)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(sol.u, rout) < 1e-6 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6


@pytest.mark.gpu
def test_traced_mass_matrix_model_through_the_device_pointer_entry_points(sa):
    """The du0 = M^{-T} nu(t0) map of a traced mass-matrix model also on the device-pointer path (Engine.adjoint_dev, the torch autograd bridge's entry point): the same numbers
    as the host-pointer call."""
    import torch
    n = 12
    rng = np.random.default_rng(13)
    M = np.eye(n) * 2.0 + 0.3 * rng.standard_normal((n, n))
    name = "wt_ring_mm"
    fun = _FUN.get(name) or sa.WideDeviceFunction.from_callable(name, ring, n, n + 1, mass_matrix=M)
    _FUN[name] = fun
    N, T, dt = 5, 0.6, 0.01
    ts = np.linspace(0.0, T, 5)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.2, 0.7, n + 1)
    delta = rng.standard_normal((N, len(ts), n))
    eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, dt, save_times=ts)
    eng.forward(u0, p)
    du0_h, dp_h = eng.adjoint(delta)
    dev = torch.device("cuda", eng.device)
    d_u0, d_p, d_delta = (torch.as_tensor(a, device=dev) for a in (u0, p, delta))
    d_du0 = torch.empty((N, n), dtype=torch.float64, device=dev); d_dp = torch.empty(n + 1, dtype=torch.float64, device=dev)
    eng.forward_dev(d_u0, d_p, None)
    eng.adjoint_dev(d_delta, d_du0, d_dp)
    eng.synchronize(); torch.cuda.synchronize()
    assert rel(d_du0.cpu().numpy(), du0_h) < 1e-13 and rel(d_dp.cpu().numpy(), dp_h) < 1e-13
    eng.close()
