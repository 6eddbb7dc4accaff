"""The host-level composition of event problems (scimlsensitivity.jl_amd/events.py) on the CPU: `solve`, `adjoint_sensitivities` and the two affect
entry points are injected, so the chain logic — which save time belongs to which piece, the extra cotangent at a piece's end, the parameter
gradient carried through parameter-changing affects — runs here against ORACLE-backed stand-ins and is checked by finite differences of the
chained oracle forward solves.  (The device versions of the same pieces are `-m gpu`: tests/test_gpu_events.py.)"""
import types

import numpy as np
import pytest

import oracle as O
from scimlsensitivity_jl_amd import events, _lib
from scimlsensitivity_jl_amd.problems import ODEProblem, EnsembleProblem, PresetTimeCallback, LsqShift


def _np_affect(u, p, t):        # u .+= p[2]/8 sin.(u); p <- 1.1 p - 0.05 (state- and parameter-dependent, parameter-changing)
    return u + p[:, 1:2] / 8.0 * np.sin(u), 1.1 * p - 0.05


def _np_affect_vjp(u, p, t, lam, gp):
    lo = lam * (1.0 + p[:, 1:2] / 8.0 * np.cos(u))
    go = 1.1 * gp
    go[:, 1] += (lam * np.sin(u)).sum(axis=1) / 8.0
    return lo, go


class _Sol(types.SimpleNamespace):
    pass


def _fake_solve(ensprob, alg, *, dt=None, saveat=None, sensealg="INTERPOLATING", **kw):
    t0, t1 = ensprob.prob.tspan
    ts = np.asarray(saveat, dtype=np.float64)
    pr = O.Problem("LV", alg=sensealg, stepper="RK4", t0=t0, t1=t1, dt=dt, save_times=ts, loss="COTANGENT")
    N = ensprob.u0.shape[0]
    _, _, out, _ = pr.adjoint_ensemble(ensprob.u0, ensprob.p, np.zeros((N, len(ts), 2)))
    return _Sol(u=out, t=ts, pr=pr, u0=ensprob.u0.copy(), p=ensprob.p.copy(), engine=types.SimpleNamespace(close=lambda: None))


def _fake_adjoint(sol, alg, *, t=None, dgdu_discrete=None, **kw):
    du0, dp, _, _ = sol.pr.adjoint_ensemble(sol.u0, sol.p, np.ascontiguousarray(dgdu_discrete))
    return du0, dp


@pytest.fixture
def lv_with_affect(monkeypatch):
    monkeypatch.setitem(_lib.MODEL, "lv_fake_user", _lib.MODEL_USER_BASE + 12345)
    monkeypatch.setattr(_lib, "affect_apply", lambda mid, u, p, t, npar, device=0: _np_affect(np.asarray(u), np.broadcast_to(p, (len(u), npar)), t))
    monkeypatch.setattr(_lib, "affect_vjp", lambda mid, u, p, t, lam, gp, device=0: _np_affect_vjp(np.asarray(u), np.broadcast_to(p, gp.shape), t, np.asarray(lam), np.array(gp)))

    def ts_of(tspan, saveat, dt, everystep, save_start, save_end):
        return np.asarray(saveat, dtype=np.float64)
    return ts_of


@pytest.mark.parametrize("shared", [True, False])
def test_event_chain_gradient_is_the_derivative_of_the_chained_loss(lv_with_affect, shared):
    rng = np.random.default_rng(7)
    N, T, dt = 3, 3.0, 0.01
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2))
    p = np.array([1.5, 1.0, 3.0, 1.0]) if shared else np.array([1.5, 1.0, 3.0, 1.0]) + 0.05 * rng.standard_normal((N, 4))
    ts = np.array([0.5, 1.0, 1.5, 2.0, 2.5, 3.0])        # 1.0 and 2.0 are event times as well: right limits
    w = rng.standard_normal((N, len(ts), 2))
    cb = PresetTimeCallback([1.0, 2.0, 7.0])

    def run(u0_, p_, grad):
        ens = EnsembleProblem(ODEProblem("lv_fake_user", u0_[0], (0.0, T), p_ if p_.ndim == 1 else p_[0]), u0_, p_)
        sol = events.solve_with_events(_fake_solve, lv_with_affect, ens, None, cb, saveat=ts, dt=dt, sensealg="INTERPOLATING")
        L = float((sol.u * w).sum())
        return (L, events.adjoint_sensitivities_events(_fake_adjoint, sol, None, dgdu_discrete=w), sol) if grad else (L, None, sol)

    L, (du0, dp), sol = run(u0, p, True)
    assert [list(c) for c in sol.piece_cols] == [[0], [1, 2], [3, 4, 5]] and np.allclose(sol.edges, [0.0, 1.0, 2.0, 3.0])
    assert dp.shape == p.shape and du0.shape == u0.shape
    h = 1e-6
    for k in range(4):
        e = np.zeros_like(p); e[..., k] = h
        fd = (run(u0, p + e, False)[0] - run(u0, p - e, False)[0]) / (2 * h)
        assert abs(fd - dp[..., k].sum()) < 2e-6 * max(1.0, abs(fd))
    e = np.zeros_like(u0); e[1, 0] = h
    fd = (run(u0 + e, p, False)[0] - run(u0 - e, p, False)[0]) / (2 * h)
    assert abs(fd - du0[1, 0]) < 2e-6 * max(1.0, abs(fd))
    # LsqShift is turned into explicit cotangents at the saved (right-limit) states
    _, (du0b, dpb), _ = (None, events.adjoint_sensitivities_events(_fake_adjoint, sol, None, dgdu_discrete=LsqShift(0.25)), None)
    _, (du0c, dpc) = None, events.adjoint_sensitivities_events(_fake_adjoint, sol, None, dgdu_discrete=sol.u - 0.25)
    assert np.array_equal(du0b, du0c) and np.array_equal(dpb, dpc)


def test_event_chain_misuse(lv_with_affect):
    u0 = np.ones((1, 2)); p = np.array([1.5, 1.0, 3.0, 1.0])
    ens = EnsembleProblem(ODEProblem("lv_fake_user", u0[0], (0.0, 1.0), p), u0, p)
    with pytest.raises(ValueError):
        events.solve_with_events(_fake_solve, lv_with_affect, ens, None, PresetTimeCallback([0.5]), saveat=[1.0], dt=0.01, checkpoints=[0.5])
    with pytest.raises(ValueError):
        events.solve_with_events(_fake_solve, lv_with_affect, EnsembleProblem(ODEProblem("lv", u0[0], (0.0, 1.0), p), u0, p), None, PresetTimeCallback([0.5]), saveat=[1.0], dt=0.01)
    sol = events.solve_with_events(_fake_solve, lv_with_affect, ens, None, PresetTimeCallback([0.5, 0.9]), saveat=[0.25, 0.5], dt=0.01, sensealg="INTERPOLATING")
    assert np.allclose(sol.edges, [0.0, 0.5, 1.0])          # the event after the last loss time is dropped
    with pytest.raises(ValueError):
        events.adjoint_sensitivities_events(_fake_adjoint, sol, None, dgdu_discrete=np.ones((1, 2, 2)), dgdp_discrete=np.ones((1, 2, 4)))
    with pytest.raises(ValueError):
        events.adjoint_sensitivities_events(_fake_adjoint, sol, None)


def test_event_chain_with_a_mass_matrix_converts_lambda_to_the_state_gradient(lv_with_affect, monkeypatch):
    """With a mass matrix a piece returns the reference's lam(t0) = M^-T dG/du0 (src/sensitivity_interface.jl:500), while the reverse callback acts on
    dG/du and the lower piece expects a dG/du cotangent: events.py converts with lam' M at every event (_lib.MASS).  Oracle-backed stand-ins in the
    mass-matrix formulation; the parameter gradient of the chain against central differences of the chained forward solves, du0 against M^-T times
    the finite-difference state gradient."""
    rng = np.random.default_rng(11)
    N, T, dt = 2, 2.0, 0.01
    M = np.array([[1.4, 0.3], [-0.2, 0.9]])
    mid = _lib.MODEL["lv_fake_user"]
    monkeypatch.setitem(_lib.MASS, mid, M)
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0]) + 0.05 * rng.standard_normal((N, 4))
    ts = np.array([0.5, 1.0, 1.5, 2.0]); w = rng.standard_normal((N, len(ts), 2))
    cb = PresetTimeCallback([0.7, 1.0])

    def run(u0_, p_, grad):
        with O.mass_matrix(M):
            ens = EnsembleProblem(ODEProblem("lv_fake_user", u0_[0], (0.0, T), p_[0]), u0_, p_)
            sol = events.solve_with_events(_fake_solve, lv_with_affect, ens, None, cb, saveat=ts, dt=dt, sensealg="INTERPOLATING")
            L = float((sol.u * w).sum())
            return (L, events.adjoint_sensitivities_events(_fake_adjoint, sol, None, dgdu_discrete=w)) if grad else (L, None)

    L, (du0, dp) = run(u0, p, True)
    h = 1e-6
    for k in range(4):
        e = np.zeros_like(p); e[:, k] = h
        fd = (run(u0, p + e, False)[0] - run(u0, p - e, False)[0]) / (2 * h)
        assert abs(fd - dp[:, k].sum()) < 3e-6 * max(1.0, abs(fd))
    g = np.zeros(2)
    for j in range(2):
        e = np.zeros_like(u0); e[1, j] = h
        g[j] = (run(u0 + e, p, False)[0] - run(u0 - e, p, False)[0]) / (2 * h)
    assert np.allclose(du0[1], np.linalg.solve(M.T, g), rtol=0, atol=3e-6 * max(1.0, np.abs(g).max()))     # lam(t0) = M^-T dG/du0
