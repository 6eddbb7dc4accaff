"""sa.DeviceFunction.from_callable / trace.py — a host-language f(du, u, p, t) traced into the three C bodies of
hipadj_model_register (the ODEFunction(f!; vjp, vjp_p) seam, src/derivative_wrappers.jl:284-359).
CPU: the traced VJP graphs against the oracle's hand-derived model VJPs and against finite differences; the emitted text compiles for
gfx950 (hipadj_model_check, no device).  GPU: a traced Lorenz gives the gradients of the compiled-in Lorenz and of the oracle."""
import os

import numpy as np
import pytest

import oracle as O


def lorenz(du, u, p, t):
    du[0] = p[0] * (u[1] - u[0])
    du[1] = u[0] * (p[1] - u[2]) - u[1]
    du[2] = u[0] * u[1] - p[2] * u[2]


def ring5(du, u, p, t):
    from scimlsensitivity_jl_amd import trace as T
    n = 5
    for i in range(n):
        du[i] = p[i] * (u[(i + 1) % n] - u[i]) + p[n] * T.sin(u[(i - 1) % n])


def wild(du, u, p, t):
    """every traced operation at least once, with shared subexpressions and time dependence"""
    from scimlsensitivity_jl_amd import trace as T
    s = T.tanh(u[0] * p[0]) + T.exp(-u[1] * u[1]) / (1.0 + p[1] ** 2)
    du[0] = s * T.cos(t) - u[0] ** 3 + T.sqrt(1.0 + u[1] * u[1]) * p[2]
    du[1] = T.log(2.0 + T.sin(u[0]) * T.cos(u[1])) - s / (1.5 + T.atan(p[0] * t)) + T.sinh(0.1 * u[0]) * T.cosh(0.2 * u[1]) + 2.5 ** 2 - T.fabs(p[2]) * u[1] ** 1.5


def _env(u, p, t, lam):
    e = {f"u[{i}]": v for i, v in enumerate(u)}
    e.update({f"p[{i}]": v for i, v in enumerate(p)}); e.update({f"lam[{i}]": v for i, v in enumerate(lam)}); e["t"] = t
    return e


@pytest.mark.parametrize("fn,omodel,n,npar,dims", [(lorenz, "LORENZ", 3, 3, (0, 0, 0, 0)), (ring5, "RING", 5, 6, (5, 0, 0, 0))])
def test_traced_vjps_equal_the_oracles_hand_derived_ones(fn, omodel, n, npar, dims):
    from scimlsensitivity_jl_amd import trace as T
    rng = np.random.default_rng(5)
    outs, u, p, t = T.trace(fn, n, npar)
    lam = [T.Node("var", name=f"lam[{i}]") for i in range(n)]
    gu, gp = T.vjp_graphs(outs, u, lam), T.vjp_graphs(outs, p, lam)
    for _ in range(5):
        uv, pv, lv = rng.standard_normal(n), rng.standard_normal(npar), rng.standard_normal(n)
        env = _env(uv, pv, 0.3, lv)
        assert np.allclose(T.evaluate(outs, env), O.model_f(omodel, uv, pv, 0.3, dims), rtol=1e-14, atol=1e-14)
        rdl, rdg = O.model_vjp(omodel, lv, uv, pv, 0.3, dims)
        assert np.allclose(T.evaluate(gu, env), rdl, rtol=1e-13, atol=1e-13) and np.allclose(T.evaluate(gp, env), rdg, rtol=1e-13, atol=1e-13)


def test_traced_vjps_against_finite_differences_for_every_operation():
    from scimlsensitivity_jl_amd import trace as T
    rng = np.random.default_rng(6)
    n, npar = 2, 3
    outs, u, p, t = T.trace(wild, n, npar)
    lam = [T.Node("var", name=f"lam[{i}]") for i in range(n)]
    gu, gp = T.vjp_graphs(outs, u, lam), T.vjp_graphs(outs, p, lam)
    uv, pv, lv, tv = np.array([0.4, 0.9]), np.array([0.7, -0.3, 1.1]), rng.standard_normal(n), 0.8
    f = lambda uu, pp: np.array(T.evaluate(outs, _env(uu, pp, tv, lv)))
    h = 1e-6
    fd_u = np.array([lv @ (f(uv + h * e, pv) - f(uv - h * e, pv)) / (2 * h) for e in np.eye(n)])
    fd_p = np.array([lv @ (f(uv, pv + h * e) - f(uv, pv - h * e)) / (2 * h) for e in np.eye(npar)])
    env = _env(uv, pv, tv, lv)
    assert np.allclose(T.evaluate(gu, env), fd_u, rtol=1e-7, atol=1e-8) and np.allclose(T.evaluate(gp, env), fd_p, rtol=1e-7, atol=1e-8)


def test_emitted_bodies_compile_for_gfx950_and_untraceable_code_is_refused():
    import scimlsensitivity_jl_amd as sa
    from scimlsensitivity_jl_amd import trace as T
    sa.build_extension()
    for name, fn, n, npar, auto in (("trace_lorenz", lorenz, 3, 3, False), ("trace_wild", wild, 2, 3, False), ("trace_wild_auto", wild, 2, 3, True)):
        fun = sa.DeviceFunction.from_callable(name, fn, n, npar, auto_vjp=auto, check=True)     # hipadj_model_check: hiprtc for gfx950, no device
        assert "du[0] =" in fun.source["f"] and (auto or "lam[" in fun.source["vjp"])
    assert "const real w" in sa.DeviceFunction.from_callable("trace_wild_auto2", wild, 2, 3, auto_vjp=True).source["f"]      # shared subexpressions become `real` temporaries

    def branching(du, u, p, t):
        du[0] = u[0] if u[0] > 0 else -u[0]
    with pytest.raises(TypeError, match="not traceable"):
        T.trace(branching, 1, 1)
    with pytest.raises(TypeError, match="in-place"):
        T.trace(lambda du, u, p, t: [u[0]], 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("auto", [False, True])
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
def test_traced_lorenz_on_device(alg, oalg, auto):
    import scimlsensitivity_jl_amd as sa
    rng = np.random.default_rng(9)
    N, T_, dt = 100, 1.5, 0.01
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.linspace(0, T_, 16)
    fun = sa.DeviceFunction.from_callable(f"lorenz_traced_{int(auto)}", lorenz, 3, 3, auto_vjp=auto)
    algs = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint(), "quadrature": sa.QuadratureAdjoint()}
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0, T_), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=algs[alg], dgdu_discrete=sa.LsqShift(2.0))
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts)
    ref = O.Problem("LORENZ", alg=oalg, stepper="RK4", t0=0, t1=T_, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, checkpointing=(alg == "backsolve"))
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    assert rel(sol.u, rout) < 1e-6 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6
    sol.engine.close()


def test_traced_vjp_bodies_compile_for_column_bundles(tmp_path, monkeypatch):
    """With bundle = True the tracer writes lam-dependent temporaries as `auto`, so the emitted VJP bodies compile for lam = Cols<G> (a bundle of segment columns through
    one pass of the body, csrc/hipadj_models.hpp): the generated model keeps HAS_COLS = true and its segmented kernels build without a device.  A
    hand-written body with a `double` temporary holding a lam term builds too — in the per-column form (HAS_COLS = false in its final translation unit)."""
    import glob
    from scimlsensitivity_jl_amd import _lib, trace

    def f(du, u, p, t):
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1] + trace.sin(u[2])
        du[1] = -p[2] * u[1] + p[3] * u[0] * u[1] / (1.0 + u[2] * u[2])
        du[2] = -u[2] * p[0] + trace.exp(-u[0])
    import scimlsensitivity_jl_amd as sa
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    F = sa.DeviceFunction.from_callable("traced_cols_probe", f, 3, 4, bundle=True)
    _lib.check_model(F.id)
    tus = sorted(glob.glob(str(tmp_path / "traced_cols_probe_*.hip")))
    assert tus and "HAS_COLS = true" in open(tus[-1]).read() and "const auto w" in open(tus[-1]).read()
    G = sa.DeviceFunction("double_temporaries_probe", 2, 4, "du[0] = p[0]*u[0] - p[1]*u[0]*u[1]; du[1] = -p[2]*u[1] + p[3]*u[0]*u[1];",
                          "double a = lam[0], b = lam[1]; out[0] = (p[0] - p[1]*u[1])*a + p[3]*u[1]*b; out[1] = -p[1]*u[0]*a + (-p[2] + p[3]*u[0])*b;",
                          "const double xy = u[0]*u[1]; double a = lam[0]; out[0] = u[0]*a; out[1] = -xy*a; out[2] = -u[1]*lam[1]; out[3] = xy*lam[1];")
    _lib.check_model(G.id)
    tus = sorted(glob.glob(str(tmp_path / "double_temporaries_probe_*.hip")))
    assert tus and "HAS_COLS = false" in open(tus[-1]).read()


def test_traced_vjp_bodies_give_the_same_numbers_through_a_bundle(tmp_path):
    """Host check of the bundle semantics on tracer output: the emitted vjp_u / vjp_p text, compiled by g++ against csrc/hipadj_models.hpp, evaluated once with
    lam = Cols<3> (three columns in one pass) and three times with lam = double — identical to the last bit (same operations per column)."""
    import subprocess
    from scimlsensitivity_jl_amd import trace

    def f(du, u, p, t):
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1] + trace.sin(u[2])
        du[1] = -p[2] * u[1] + p[3] * u[0] * u[1] / (1.0 + u[2] * u[2])
        du[2] = -u[2] * p[0] + trace.exp(-u[0]) * t
    fb, vu, vp = trace.bodies(f, 3, 4, bundle=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "bundle_host.cpp"
    src.write_text('''
#include <cmath>
#include <cstdio>
using std::sin; using std::cos; using std::exp; using std::pow; using std::log; using std::sqrt; using std::tanh;
#include "hipadj_models.hpp"
using namespace hipadj;
constexpr int N = 3, NP = 4;
template <class LT> void vjp_u_t(LT (&out)[N], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { (void)t; %s }
template <class LT> void vjp_p_t(LT (&out)[NP], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { (void)t; %s }
int main() {
    const double u[N] = {0.7, 1.3, -0.4}, p[NP] = {1.5, 1.0, 3.0, 0.8}, t = 0.37;
    const double L[3][N] = {{0.3, -1.1, 0.6}, {1.0, 0.0, 0.0}, {-0.2, 0.9, 2.5}};
    Cols<3> lam[N], ou[N], op[NP];
    for (int j = 0; j < N; ++j) for (int g = 0; g < 3; ++g) lam[j].v[g] = L[g][j];
    vjp_u_t<Cols<3>>(ou, lam, u, p, t); vjp_p_t<Cols<3>>(op, lam, u, p, t);
    int bad = 0;
    for (int g = 0; g < 3; ++g) {
        double l[N], a[N], b[NP];
        for (int j = 0; j < N; ++j) l[j] = L[g][j];
        vjp_u_t<double>(a, l, u, p, t); vjp_p_t<double>(b, l, u, p, t);
        for (int j = 0; j < N; ++j) bad += a[j] != ou[j].v[g];
        for (int j = 0; j < NP; ++j) bad += b[j] != op[j].v[g];
    }
    std::printf("bad %%d\\n", bad);
    return bad;
}
''' % (vu, vp))
    exe = tmp_path / "bundle_host"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "scimlsensitivity.jl_amd", "csrc"), str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr
