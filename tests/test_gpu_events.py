"""DiscreteCallback at preset times with a state affect (`-m gpu`): test/Callbacks1/discrete_callbacks.jl's cases — a constant dose, several event times,
a state- and parameter-dependent affect, a state reset — on the device (pieces chained by hipadj_affect_apply / hipadj_affect_vjp, events.py)
against the same chain built from the ORACLE's per-piece adjoints with the affect and its VJP written in numpy, and against finite differences
of the chained oracle forward solves (the reference's own assertion: adjoint ≈ ForwardDiff through the solve with the callback)."""
import numpy as np
import pytest

import oracle as O
import user_models as UM

pytestmark = pytest.mark.gpu
_reg = {}

AFFECTS = {
    # name: (device text, numpy affect (u [N][n], p [N][np]) -> (un, pn), numpy reverse callback (u, p, lam, gp) -> (lam_out, gp_out))
    "dose": ("un[0] += 2.0;", lambda u, p: (u + np.array([2.0, 0.0]), p), lambda u, p, l, g: (l.copy(), g.copy())),
    "sin": ("for (int i = 0; i < N; ++i) un[i] += p[1] / 8.0 * sin(u[i]);",
            lambda u, p: (u + p[:, 1:2] / 8.0 * np.sin(u), p),
            lambda u, p, l, g: (l * (1.0 + p[:, 1:2] / 8.0 * np.cos(u)), g + np.stack([np.zeros(len(u)), (l * np.sin(u)).sum(axis=1) / 8.0, np.zeros(len(u)), np.zeros(len(u))], axis=1))),
    "reset": ("un[0] = 2.0;", lambda u, p: (np.stack([np.full(len(u), 2.0), u[:, 1]], axis=1), p), lambda u, p, l, g: (np.stack([np.zeros(len(u)), l[:, 1]], axis=1), g.copy())),
    # test/Callbacks1/discrete_callbacks.jl:303-312: integrator.p .= 2 p .- 0.5 (here milder, so that the dynamics stay tame), plus a state term that reads p
    "pchange": ("for (int k = 0; k < NP; ++k) pn[k] = 1.1 * p[k] - 0.05; un[1] += 0.1 * p[3] * u[0];",
                lambda u, p: (np.stack([u[:, 0], u[:, 1] + 0.1 * p[:, 3] * u[:, 0]], axis=1), 1.1 * p - 0.05),
                lambda u, p, l, g: (np.stack([l[:, 0] + 0.1 * p[:, 3] * l[:, 1], l[:, 1]], axis=1),
                                    1.1 * g + np.stack([np.zeros(len(u)), np.zeros(len(u)), np.zeros(len(u)), 0.1 * u[:, 0] * l[:, 1]], axis=1))),
}


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def fun(sa, name):
    if name not in _reg:
        m = UM.LV
        _reg[name] = sa.DeviceFunction("lv_event_" + name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"]).set_affect(AFFECTS[name][0])
    return _reg[name]


def oracle_chain(name, events, ts, T, u0, pp, delta, alg, okw, shared):
    """the same composition from the oracle's pieces (per-trajectory parameters throughout): returns (u at ts, du0, dp)"""
    aff, vjp = AFFECTS[name][1], AFFECTS[name][2]
    N = len(u0)
    P = np.ascontiguousarray(np.broadcast_to(pp, (N, 4)))
    ev = sorted(e for e in events if 0.0 < e < T and e <= ts[-1])
    edges = [0.0] + ev + [T]
    pieces, u, out, ul = [], u0, np.zeros((N, len(ts), 2)), []
    for j in range(len(edges) - 1):
        a, b = edges[j], edges[j + 1]; last = j == len(edges) - 2
        own = [i for i, s in enumerate(ts) if (a <= s < b) or (last and s == b)]
        sv = np.array([ts[i] for i in own] + ([] if last else [b]))
        pr = O.Problem("LV", alg=alg, t0=a, t1=b, save_times=sv, loss="COTANGENT", checkpointing=(alg == "BACKSOLVE"), quad_abstol=1e-12, quad_reltol=1e-12, **okw)
        _, _, o, _ = pr.adjoint_ensemble(u, P, np.zeros((N, len(sv), 2)))
        pieces.append((pr, own, u.copy(), P.copy()))
        for q, i in enumerate(own):
            out[:, i] = o[:, q]
        if not last:
            ul.append(o[:, -1].copy()); u, P = aff(o[:, -1], P); P = np.ascontiguousarray(P)
    gp = np.zeros((N, 4)); lam_in = None; du0 = None
    for j in range(len(pieces) - 1, -1, -1):
        pr, own, ustart, Pj = pieces[j]
        cot = [delta[:, i] for i in own] + ([lam_in] if j < len(pieces) - 1 else [])
        du0, dpj, _, _ = pr.adjoint_ensemble(ustart, Pj, np.ascontiguousarray(np.stack(cot, axis=1)))
        gp = gp + dpj
        if j > 0:
            lam_in, gp = vjp(ul[j - 1], pieces[j - 1][3], du0, gp)
    return out, du0, (gp.sum(axis=0) if shared else gp)


ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")]


def sensealg_of(sa, alg):
    return {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint(), "quadrature": sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12)}[alg]


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("name,events", [("dose", [5.0]), ("dose", [2.0, 4.0, 8.0]), ("sin", [5.0]), ("reset", [5.0]), ("pchange", [5.0]), ("pchange", [3.0, 6.5])])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_discrete_callback_matches_oracle_chain(sa, alg, oalg, name, events, stepper):
    """test/Callbacks1/discrete_callbacks.jl:260-330: saveat 0.5 on (0, 10), Lotka-Volterra, g = sum(sol) replaced by random cotangents; the
    event time 5.0 (and 2.0, 4.0, 8.0) is a save time as well: the saved state there is the right limit."""
    rng = np.random.default_rng(61)
    N, T = 70, 10.0
    shared = name not in ("sin",)
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2))
    pp = np.array([1.5, 1.0, 3.0, 1.0]) if shared else np.array([1.5, 1.0, 3.0, 1.0]) + 0.05 * rng.standard_normal((N, 4))
    ts = np.arange(0.0, T + 1e-9, 0.5)
    delta = rng.standard_normal((N, len(ts), 2))
    f = fun(sa, name)
    if stepper == "rk4":
        salg, kw, okw = sa.RK4(), dict(dt=0.01), dict(stepper="RK4", dt=0.01)
    else:
        salg, kw, okw = sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), pp if shared else pp[0]), u0, pp), salg, saveat=ts, sensealg=sensealg_of(sa, alg),
                   callback=sa.PresetTimeCallback(events), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, dgdu_discrete=delta)
    rout, rdu0, rdp = oracle_chain(name, events, ts, T, u0, pp, delta, oalg, okw, shared)
    assert rel(sol.u, rout) < 1e-8 and rel(du0, rdu0) < 1e-7 and rel(dp, rdp) < 1e-7
    sol.close()


@pytest.mark.parametrize("name", ["sin", "pchange"])
def test_discrete_callback_gradient_is_the_derivative_of_the_loss(sa, name):
    """The reference's assertion (discrete_callbacks.jl:200-216): adjoint ≈ ForwardDiff through the solve with the callback — here central
    differences of the loss through the device's own forward solves with the callback (state- and parameter-dependent affect, two events)."""
    rng = np.random.default_rng(62)
    T, events = 6.0, [2.0, 4.0]
    u0 = np.array([[1.0, 1.0]]); pp = np.array([1.5, 1.0, 3.0, 1.0])
    ts = np.arange(0.5, T + 1e-9, 0.5)
    w = rng.standard_normal((1, len(ts), 2))
    f = fun(sa, name)

    def loss_and_grad(u0_, p_, grad):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0_[0], (0.0, T), p_), u0_, p_), sa.RK4(), dt=0.005, saveat=ts, sensealg=sa.InterpolatingAdjoint(),
                       callback=sa.PresetTimeCallback(events))
        L = float((sol.u * w).sum())
        g = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=w) if grad else None
        sol.close()
        return L, g
    _, (du0, dp) = loss_and_grad(u0, pp, True)
    h = 1e-6
    for k in range(4):
        e = np.zeros(4); e[k] = h
        fd = (loss_and_grad(u0, pp + e, False)[0] - loss_and_grad(u0, pp - e, False)[0]) / (2 * h)
        assert abs(fd - dp[k]) < 2e-6 * max(1.0, abs(dp[k]))
    for k in range(2):
        e = np.zeros((1, 2)); e[0, k] = h
        fd = (loss_and_grad(u0 + e, pp, False)[0] - loss_and_grad(u0 - e, pp, False)[0]) / (2 * h)
        assert abs(fd - du0[0, k]) < 2e-6 * max(1.0, abs(du0[0, k]))


def test_discrete_callback_misuse(sa):
    u0 = np.array([[1.0, 1.0]]); pp = np.array([1.5, 1.0, 3.0, 1.0])
    with pytest.raises(ValueError):      # a compiled-in model carries no affect
        sa.solve(sa.EnsembleProblem(sa.ODEProblem("lv", u0[0], (0.0, 1.0), pp), u0, pp), sa.RK4(), dt=0.01, saveat=[1.0], callback=sa.PresetTimeCallback([0.5]))
    f = fun(sa, "dose")
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 1.0), pp), u0, pp), sa.RK4(), dt=0.01, saveat=[0.5, 1.0], callback=sa.PresetTimeCallback([0.5, 7.0, 0.0]))
    assert len(sol.pieces) == 2 and np.allclose(sol.edges, [0.0, 0.5, 1.0])      # events outside (t0, t1) are ignored
    with pytest.raises(ValueError):
        sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=np.ones((1, 2, 2)), dgdp_discrete=np.zeros((1, 2, 4)))
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=sa.LsqShift(0.5))
    assert np.all(np.isfinite(du0)) and np.all(np.isfinite(dp))
    sol.close()


def test_concrete_solve_adjoint_with_a_callback(sa):
    """The (out, pullback) contract (src/concrete_solve.jl:523-1042) with `callback` among the solve keywords (:563-575 track_callbacks)."""
    rng = np.random.default_rng(63)
    N, T = 8, 4.0
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); pp = np.array([1.5, 1.0, 3.0, 1.0])
    ts = np.arange(0.5, T + 1e-9, 0.5)
    f = fun(sa, "sin")
    out, pullback = sa.concrete_solve_adjoint(sa.ODEProblem(f, u0[0], (0.0, T), pp), sa.RK4(), sa.GaussAdjoint(), u0, pp, dt=0.01, saveat=ts, callback=sa.PresetTimeCallback([2.0]))
    delta = rng.standard_normal(out.shape)
    du0, dp = pullback(delta)
    rout, rdu0, rdp = oracle_chain("sin", [2.0], ts, T, u0, pp, delta, "GAUSS", dict(stepper="RK4", dt=0.01), True)
    assert rel(out, rout) < 1e-9 and rel(du0, rdu0) < 1e-8 and rel(dp, rdp) < 1e-8


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("name,events", [("dose", [2.0, 4.0, 8.0]), ("sin", [5.0]), ("pchange", [3.0, 6.5])])
def test_discrete_callback_with_the_stiff_stepper(sa, alg, oalg, name, events):
    """The same chains with Rosenbrock23 (round 6): every piece's forward and reverse solve on the stiff stepper, the pieces joined by the affect and its reverse callback."""
    rng = np.random.default_rng(62)
    N, T = 40, 10.0
    shared = name not in ("sin",)
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2))
    pp = np.array([1.5, 1.0, 3.0, 1.0]) if shared else np.array([1.5, 1.0, 3.0, 1.0]) + 0.05 * rng.standard_normal((N, 4))
    ts = np.arange(0.0, T + 1e-9, 0.5)
    delta = rng.standard_normal((N, len(ts), 2))
    f = fun(sa, name)
    salg, kw, okw = sa.Rosenbrock23(), dict(abstol=1e-9, reltol=1e-9), dict(stepper="ROS23", dt=0.0, abstol=1e-9, reltol=1e-9)
    sens = sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10) if alg == "quadrature" else sensealg_of(sa, alg)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), pp if shared else pp[0]), u0, pp), salg, saveat=ts, sensealg=sens, callback=sa.PresetTimeCallback(events), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, dgdu_discrete=delta)
    rout, rdu0, rdp = oracle_chain(name, events, ts, T, u0, pp, delta, oalg, okw, shared)
    bar = 1e-4 if alg == "backsolve" else 1e-5      # two implementations of one adaptive controller over ~3000 reverse steps per piece (and BacksolveAdjoint's growth)
    assert rel(sol.u, rout) < 1e-7 and rel(du0, rdu0) < bar and rel(dp, rdp) < bar
    sol.close()
