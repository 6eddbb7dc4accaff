"""Device-resident discrete losses, the parts that need no GPU: the oracle's new loss kinds against independent scipy gradients (tests/golden/discrete_losses.json,
make_discrete_losses.py), the generated device code of models with discrete-loss bodies compiled for gfx950 (hiprtc cross-compiles without a device), and the host logic."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O
import user_models as UM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGS = ["INTERPOLATING", "BACKSOLVE", "GAUSS", "QUADRATURE"]


@pytest.fixture(scope="module")
def G():
    with open(os.path.join(ROOT, "tests", "golden", "discrete_losses.json")) as f:
        return json.load(f)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("name,kw", [("u1sq_p1", dict(loss="TEST", dloss_id=2)), ("u1sq_p2", dict(loss="TEST", dloss_id=3)), ("full", dict(loss="TEST", dloss_id=4)),
                                     ("lsq_data", dict(loss="LSQ_DATA", loss_scale=2.0))])
def test_oracle_discrete_losses_against_scipy_forward_sensitivities(G, alg, name, kw):
    """dgdu_discrete + dgdp_discrete (src/adjoint_common.jl:771-779; the discrete cost of test/Core7/mixed_costs.jl:199-390 and the discrete part of its mixed cost :391-570):
    the restatement meets ForwardDiff's stand-in at 1e-10 with every sensealg — GaussAdjoint under the repo's convention (the reference drops dgdp_discrete there)."""
    ts = np.array(G["ts"]); u0 = np.array(G["u0"])[None]; p = np.array(G["p"]); data = np.array(G["data"])[None]
    ref = O.Problem("LV", alg=alg, stepper="TSIT5", t0=0.0, t1=10.0, dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, quad_abstol=1e-13, quad_reltol=1e-13,
                    checkpointing=(alg == "BACKSOLVE"), **kw)
    du0, dp, out, _ = ref.adjoint_ensemble(u0, p, data)
    g = G["losses"][name]
    assert rel(out[0], np.array(G["u"])) < 1e-10
    assert rel(du0[0], g["du0"]) < 1e-10 and rel(dp, g["dp"]) < 1e-10


def test_oracle_loss_kinds_are_consistent():
    """LSQ_DATA with scale 2 == the test loss sum |u - d|^2 == the cotangent path with Delta = 2 (out - data), bit for bit (one code path, three spellings); and the reference-literal
    GaussAdjoint drops exactly sum_i dgdp_discrete."""
    rng = np.random.default_rng(0)
    ts = np.linspace(0, 2, 11); u0 = np.array([[1.0, 1.0]]); p = np.array([1.5, 1.0, 3.0, 1.0]); d = rng.standard_normal((1, len(ts), 2))
    for alg in ALGS:
        kw = dict(alg=alg, stepper="RK4", t0=0, t1=2, dt=0.01, save_times=ts, checkpointing=(alg == "BACKSOLVE"))
        a = O.Problem("LV", loss="LSQ_DATA", loss_scale=2.0, **kw).adjoint_ensemble(u0, p, d)
        b = O.Problem("LV", loss="TEST", dloss_id=1, **kw).adjoint_ensemble(u0, p, d)
        c = O.Problem("LV", loss="COTANGENT", **kw).adjoint_ensemble(u0, p, 2.0 * (a[2] - d))
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    kw = dict(alg="GAUSS", stepper="RK4", t0=0, t1=2, dt=0.01, save_times=ts, loss="TEST", dloss_id=2)
    x = O.Problem("LV", **kw).adjoint_ensemble(u0, p, None)
    y = O.Problem("LV", reference_literal=True, **kw).adjoint_ensemble(u0, p, None)
    assert np.array_equal(x[0], y[0]) and np.allclose(x[1] - y[1], [len(ts), 0, 0, 0], atol=1e-12)


FULL_DGDU = "for (int j = 0; j < N; ++j) out[j] = 0.0; out[0] += (i + 1) * p[0] * u[N - 1] + sin(t); out[N - 1] += (i + 1) * p[0] * u[0] + p[1] * p[1] * d[0];"
FULL_DGDP = "for (int j = 0; j < NP; ++j) out[j] = 0.0; out[0] = (i + 1) * u[0] * u[N - 1]; out[1] = 2.0 * p[1] * d[0] * u[N - 1];"


def _cfg(sa, model_id, alg, stepper, ts, kind=3, ck=0):
    from scimlsensitivity_jl_amd import _lib
    c = _lib.HipadjConfig(); c.struct_size = C.sizeof(c); c.model = model_id; c.alg = alg; c.stepper = stepper
    c.ntraj = 100; c.t0 = 0.0; c.t1 = 10.0; c.dt = 0.01; c.nsave = len(ts)
    c._ts = np.ascontiguousarray(ts, dtype=np.float64); c.save_times = c._ts.ctypes.data_as(C.POINTER(C.c_double))
    c.loss_kind = kind; c.checkpointing = ck; c.p_shared = 1; c.abstol = c.reltol = 1e-8; c.quad_abstol = c.quad_reltol = 1e-8
    return c


@pytest.mark.parametrize("how", ["bodies", "function"])
def test_lane_model_with_discrete_loss_bodies_compiles_for_gfx950(sa, how):
    """hipadj_model_check_config compiles every kernel a HIPADJ_LOSS_MODEL handle would launch (no device needed): the loss bodies are instantiated in the sweeps of the
    fixed-step and the adaptive family (compiled-in models never instantiate that code, so this is where a syntax slip in it would surface)."""
    L = sa.load_library()
    f = sa.DeviceFunction(f"lv_dloss_cpu_{how}", 2, 4, UM.LV["f"], UM.LV["vjp"], UM.LV["vjp_p"])
    if how == "bodies":
        f.set_discrete_loss(dgdu=FULL_DGDU, dgdp=FULL_DGDP)
    else:
        f.set_discrete_loss(l="l = (i + 1) * p[0] * u[0] * u[N - 1] + sin(t) * u[0] + p[1] * p[1] * d[0] * u[N - 1];")
    ts = np.arange(1.0, 9.5, 1.0)
    for stepper, alg, ck in ((0, 0, 0), (0, 3, 0), (0, 1, 1), (1, 2, 0)):
        c = _cfg(sa, f.id, alg, stepper, ts, ck=ck)
        rc = L.hipadj_model_check_config(C.byref(c))
        assert rc == 0, L.hipadj_last_error(None).decode()[:2000]


def test_loss_kind_validation(sa):
    """HIPADJ_LOSS_MODEL needs a runtime model WITH bodies; unknown kinds are refused; a body that does not compile fails with the compiler's log."""
    L = sa.load_library()
    ts = np.arange(1.0, 9.5, 1.0)
    plain = sa.DeviceFunction("lv_no_dloss", 2, 4, UM.LV["f"], UM.LV["vjp"], UM.LV["vjp_p"])
    from scimlsensitivity_jl_amd import _lib
    c = _cfg(sa, _lib.MODEL["lorenz"], 0, 0, ts, kind=3)
    assert L.hipadj_model_check_config(C.byref(c)) == 0           # built-in models have nothing to compile ...
    h = C.c_void_p()
    rc = L.hipadj_create(C.byref(c), C.byref(h))                  # ... and creation refuses the kind (before it looks for a device)
    assert rc == _lib.ERR_INVALID_ARG and b"HIPADJ_LOSS_MODEL" in L.hipadj_last_error(None)
    c = _cfg(sa, plain.id, 0, 0, ts, kind=7)
    assert L.hipadj_create(C.byref(c), C.byref(h)) == _lib.ERR_INVALID_ARG and b"loss_kind" in L.hipadj_last_error(None)
    bad = sa.DeviceFunction("lv_bad_dloss", 2, 4, UM.LV["f"], UM.LV["vjp"], UM.LV["vjp_p"])
    bad.set_discrete_loss(dgdu="out[0] = undeclared_name; out[1] = 0.0;")
    c = _cfg(sa, bad.id, 0, 0, ts)
    assert L.hipadj_model_check_config(C.byref(c)) != 0 and b"undeclared_name" in L.hipadj_last_error(None)
    with pytest.raises(sa.HipadjError):
        sa.WideDeviceFunction.dense_chain("chain_lane_loss", (2, 8, 2)).set_discrete_loss(dgdu="out[0] = 0.0;")      # a wide model takes one SPMD body


def test_wide_model_with_a_discrete_loss_body_compiles_for_gfx950(sa):
    L = sa.load_library()
    f = sa.WideDeviceFunction.dense_chain("chain_dloss_cpu", (2, 16, 2))
    f.set_discrete_loss(body="if (tid == 0) { dlam[0] += 2.0 * (u[0] - (d ? d[0] : 0.0)); if (WP) gp[0] += u[1]; }")
    ts = np.linspace(0.0, 10.0, 11)
    for stepper, alg in ((0, 0), (0, 3), (1, 1), (1, 3)):
        c = _cfg(sa, f.id, alg, stepper, ts)
        rc = L.hipadj_model_check_config(C.byref(c))
        assert rc == 0, L.hipadj_last_error(None).decode()[:2000]


def test_traced_discrete_loss_bodies_are_the_gradients_of_the_callable():
    """trace.discrete_loss_bodies: the emitted dgdu / dgdp text of a host-language loss l(u, p, t, i, d), evaluated as C, equals central differences of the callable, and the
    emitted loss body returns its value (compiled with gcc: the same text the device compiles)."""
    import ctypes, math, subprocess, tempfile
    from scimlsensitivity_jl_amd import trace

    def l(u, p, t, i, d, sin=trace.sin):
        return (i + 1.0) * p[0] * u[0] * u[1] + sin(t) * u[0] + p[1] ** 2 * d[0] * u[1] + u[1] / (1.0 + d[1] * d[1])
    gu, gp, lb = trace.discrete_loss_bodies(l, 2, 4)
    src = f"""#include <math.h>
typedef double real;
void dgdu(double* out, const double* u, const double* p, double t, int i, const double* d) {{ {gu} }}
void dgdp(double* out, const double* u, const double* p, double t, int i, const double* d) {{ {gp} }}
double lval(const double* u, const double* p, double t, int i, const double* d) {{ real l = 0.0; {lb} return l; }}
"""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "dl.c"); so = os.path.join(td, "dl.so")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-O1", "-shared", "-fPIC", c, "-o", so, "-lm"], check=True)
        L = ctypes.CDLL(so)
        A = ctypes.POINTER(ctypes.c_double)
        L.lval.restype = ctypes.c_double
        for fn in (L.dgdu, L.dgdp, L.lval):
            fn.argtypes = [A, A, A, ctypes.c_double, ctypes.c_int, A][(0 if fn is not L.lval else 1):]
        u = np.array([1.3, 0.7]); p = np.array([1.5, 1.0, 3.0, 1.0]); d = np.array([0.9, 1.7]); t = 2.5; i = 3
        ptr = lambda a: a.ctypes.data_as(A)
        f = lambda uu, pp: l(uu, pp, t, float(i), d, sin=math.sin)
        assert abs(L.lval(ptr(u), ptr(p), t, i, ptr(d)) - f(u, p)) < 1e-14
        out_u = np.zeros(2); out_p = np.zeros(4)
        L.dgdu(ptr(out_u), ptr(u), ptr(p), t, i, ptr(d)); L.dgdp(ptr(out_p), ptr(u), ptr(p), t, i, ptr(d))
        h = 1e-6
        for j in range(2):
            e = np.zeros(2); e[j] = h
            assert abs(out_u[j] - (f(u + e, p) - f(u - e, p)) / (2 * h)) < 1e-8
        for j in range(4):
            e = np.zeros(4); e[j] = h
            assert abs(out_p[j] - (f(u, p + e) - f(u, p - e)) / (2 * h)) < 1e-8
