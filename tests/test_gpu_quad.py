"""Four lanes per trajectory (csrc/hipadj_quad.hpp, hipadj_quad_ts5.hpp): the forward solves and the adaptive Interpolating / Backsolve / Gauss sweeps of
models with a component form, against the one-lane-per-trajectory kernels (HIPADJ_QUAD=0) and the oracle.  The component forms reassociate the right-hand sides, so the two
mappings agree to roundoff, not bitwise."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import rel, lorenz_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,omodel,n,p", [("lorenz", "LORENZ", 3, [10.0, 28.0, 8.0 / 3.0]), ("lv", "LV", 2, [1.5, 1.0, 3.0, 1.0]), ("lvt", "LVT", 2, [1.5, 1.0, 3.0, 1.0])])
@pytest.mark.parametrize("N", [1, 61, 1000])
def test_forward_rk4_quad_equals_lane_and_oracle(sa, monkeypatch, model, omodel, n, p, N):
    rng = np.random.default_rng(N)
    u0 = (np.array([1.0, 0.0, 0.0]) if n == 3 else np.array([1.0, 1.0])) + 0.1 * rng.standard_normal((N, n)); p = np.array(p)
    T, dt = 2.0, 0.01
    ts = np.linspace(0.0, T, 9)
    delta = rng.standard_normal((N, len(ts), n))
    res = {}
    for quad in ("1", "0"):
        monkeypatch.setenv("HIPADJ_QUAD", quad)
        eng = sa.Engine(model, "interpolating", N, 0.0, T, dt, save_times=ts)
        out = eng.forward(u0, p)
        res[quad] = (out,) + eng.adjoint(delta)
        eng.close()
    for a, b in zip(res["1"], res["0"]):
        assert rel(a, b) < 1e-11
    ref = O.Problem(omodel, alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(res["1"][0], rout) < 1e-10 and rel(res["1"][1], rdu0) < 1e-6 and rel(res["1"][2], rdp) < 1e-6


def test_forward_rk4_quad_backsolve_checkpoints_and_yT(sa, monkeypatch):
    """the quad forward solve also writes Backsolve's checkpoints and y(T)"""
    N, T, dt = 130, 1.0, 0.01
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0.0, T, 11)
    res = {}
    for quad in ("1", "0"):
        monkeypatch.setenv("HIPADJ_QUAD", quad)
        eng = sa.Engine("lorenz", "backsolve", N, 0.0, T, dt, save_times=ts, loss_kind=1, loss_shift=2.0, checkpointing=True)
        eng.forward(u0, p, want_out=False)
        res[quad] = eng.adjoint(None)
        eng.close()
    assert rel(res["1"][0], res["0"][0]) < 1e-10 and rel(res["1"][1], res["0"][1]) < 1e-10


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS")])
@pytest.mark.parametrize("tol", [1e-8, 1e-11])
@pytest.mark.parametrize("loss", ["lsq", "cot"])
def test_adaptive_sweeps_quad_against_the_oracle_with_interior_loss_times(sa, monkeypatch, alg, oalg, tol, loss):
    """Tight tolerances and many interior loss times: the case that exposed a wrong one-component instantiation of the Gauss sweep (hipadj_quad_ts5.hpp QuadNZ)."""
    monkeypatch.setenv("HIPADJ_QUAD", "1")
    N, T = 37, 1.0
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0.0, T, 11)
    rng = np.random.default_rng(3)
    delta = rng.standard_normal((N, len(ts), 3)) if loss == "cot" else None
    kw = dict(loss_kind=1, loss_shift=2.0) if loss == "lsq" else {}
    eng = sa.Engine("lorenz", alg, N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=tol, reltol=tol, checkpointing=(alg == "backsolve"), p_shared=False, **kw)
    eng.forward(u0, np.tile(p, (N, 1)), want_out=False)
    du0, dp = eng.adjoint(delta)
    eng.close()
    ref = O.Problem("LORENZ", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=tol, reltol=tol, save_times=ts, checkpointing=(alg == "backsolve"),
                    **(dict(loss="LSQ_SHIFT", loss_shift=2.0) if loss == "lsq" else {}))
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, np.tile(p, (N, 1)), delta)
    bound = 1e-8 if tol == 1e-8 else 1e-11
    assert rel(du0, rdu0) < bound and rel(dp, rdp) < bound


def test_adaptive_forward_quad_out_and_records(sa, monkeypatch):
    """out = sol(ts) at arbitrary save times and the dense records of the quad forward solve feed the lane family's Quadrature kernels unchanged"""
    N, T = 70, 1.5
    u0, p = lorenz_inputs(N)
    ts = np.array([0.0, 0.137, 0.61, 1.0, 1.5])
    rng = np.random.default_rng(5)
    delta = rng.standard_normal((N, len(ts), 3))
    res = {}
    for quad in ("1", "0"):
        monkeypatch.setenv("HIPADJ_QUAD", quad)
        eng = sa.Engine("lorenz", "quadrature", N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=1e-10, reltol=1e-10, quad_abstol=1e-12, quad_reltol=1e-12)
        out = eng.forward(u0, p)
        res[quad] = (out,) + eng.adjoint(delta)
        eng.close()
    for a, b in zip(res["1"], res["0"]):
        assert rel(a, b) < 1e-8
    ref = O.Problem("LORENZ", alg="QUADRATURE", stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, quad_abstol=1e-12, quad_reltol=1e-12)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(res["1"][0], rout) < 1e-8 and rel(res["1"][1], rdu0) < 1e-6 and rel(res["1"][2], rdp) < 1e-6


@pytest.mark.parametrize("p_shared", [True, False])
@pytest.mark.parametrize("model,omodel", [("lv", "LV"), ("lvt", "LVT")])
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS")])
def test_lotka_volterra_adaptive_sweeps_quad_against_lane_and_oracle(sa, monkeypatch, alg, oalg, model, omodel, p_shared):
    """QuadAdjLV (hipadj_quad_ts5.hpp): two lanes carry y and lam, all four a parameter sum; plain and time-dependent LV, shared and per-trajectory parameters, 67 trajectories
    (a partial last wavefront), off-grid loss times — against the one-lane-per-trajectory kernels and the oracle."""
    rng = np.random.default_rng(23)
    N, T = 67, 3.0
    u0 = np.array([1.0, 1.0]) + 0.2 * rng.standard_normal((N, 2))
    p = np.array([1.5, 1.0, 3.0, 1.0]) * (1 + 0.05 * rng.standard_normal((N, 4)))
    if p_shared:
        p = p[0]
    ts = np.array([0.0, 0.4, 1.0, 1.7, 2.5, T])
    delta = rng.standard_normal((N, len(ts), 2))
    ck = alg == "backsolve"
    res = {}
    for quad in ("2", "0"):
        monkeypatch.setenv("HIPADJ_QUAD", quad)
        eng = sa.Engine(model, alg, N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=1e-10, reltol=1e-10, checkpointing=ck, p_shared=p_shared)
        out = eng.forward(u0, p)
        res[quad] = (out,) + eng.adjoint(delta)
        eng.close()
    for a, b in zip(res["2"], res["0"]):
        assert rel(a, b) < 1e-9
    ref = O.Problem(omodel, alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, checkpointing=ck, loss="COTANGENT")
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(res["2"][0], rout) < 1e-10 and rel(res["2"][1], rdu0) < 1e-8 and rel(res["2"][2], rdp) < 1e-8
