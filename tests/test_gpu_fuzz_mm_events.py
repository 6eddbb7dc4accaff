"""Randomized combinations (`-m gpu`) of the round's host-facing additions on runtime models: a constant mass matrix, DiscreteCallback events (state- and
parameter-dependent, parameter-changing), both steppers, all four sensealgs, shared / per-trajectory parameters — device vs the same chain built
from the oracle's pieces (the oracle in its mass-matrix formulation), 56 seeds (the last 16 on 6- and 8-state models)."""
import numpy as np
import pytest

import oracle as O
import user_models as UM

pytestmark = pytest.mark.gpu
_reg = {}
ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")]
AFFECT = "for (int i = 0; i < N; ++i) un[i] += 0.2 * p[0] * sin(u[i]); pn[1] = 1.05 * p[1] + 0.01 * u[0];"


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def np_affect(u, p):
    pn = p.copy(); pn[:, 1] = 1.05 * p[:, 1] + 0.01 * u[:, 0]
    return u + 0.2 * p[:, 0:1] * np.sin(u), pn


def np_affect_vjp(u, p, lam, gp):
    lo = lam * (1.0 + 0.2 * p[:, 0:1] * np.cos(u))
    lo[:, 0] += 0.01 * gp[:, 1]
    go = gp.copy(); go[:, 1] = 1.05 * gp[:, 1]
    go[:, 0] += 0.2 * (lam * np.sin(u)).sum(axis=1)
    return lo, go


def oracle_chain(omodel, dims, n, npar, events, ts, T, u0, pp, delta, alg, okw, shared, M):
    N = len(u0)
    P = np.ascontiguousarray(np.broadcast_to(pp, (N, npar)))
    ev = sorted(e for e in events if 0.0 < e < T and e <= ts[-1])
    edges = [0.0] + ev + [T]
    pieces, u, out, ul = [], u0, np.zeros((N, len(ts), n)), []
    for j in range(len(edges) - 1):
        a, b = edges[j], edges[j + 1]; last = j == len(edges) - 2
        own = [i for i, s in enumerate(ts) if (a <= s < b) or (last and s == b)]
        sv = np.array([ts[i] for i in own] + ([] if last else [b]))
        pr = O.Problem(omodel, alg=alg, t0=a, t1=b, save_times=sv, loss="COTANGENT", checkpointing=(alg == "BACKSOLVE"), dims=dims, quad_abstol=1e-12, quad_reltol=1e-12, **okw)
        _, _, o, _ = pr.adjoint_ensemble(u, P, np.zeros((N, len(sv), n)))
        pieces.append((pr, own, u.copy(), P.copy()))
        for q, i in enumerate(own):
            out[:, i] = o[:, q]
        if not last:
            ul.append(o[:, -1].copy()); u, P = np_affect(o[:, -1], P); P = np.ascontiguousarray(P)
    gp = np.zeros((N, npar)); lam_in = None; du0 = None
    for j in range(len(pieces) - 1, -1, -1):
        pr, own, ustart, Pj = pieces[j]
        cot = [delta[:, i] for i in own] + ([lam_in] if j < len(pieces) - 1 else [])
        du0, dpj, _, _ = pr.adjoint_ensemble(ustart, Pj, np.ascontiguousarray(np.stack(cot, axis=1)))
        gp = gp + dpj
        if j > 0:
            lam_true = du0 if M is None else du0 @ M
            lam_in, gp = np_affect_vjp(ul[j - 1], pieces[j - 1][3], lam_true, gp)
    return out, du0, (gp.sum(axis=0) if shared else gp)


@pytest.mark.parametrize("seed", range(56))
def test_random_mass_matrix_and_event_combinations(sa, seed):
    rng = np.random.default_rng(7000 + seed)
    name = ["rober", "ring4", "ring5"][int(rng.integers(3))] if seed < 40 else ["ring6", "ring8"][int(rng.integers(2))]   # 40+: the wide models (per-column segment lanes)
    m, omodel, dims = (UM.ROBER, "ROBER", (0, 0, 0, 0)) if name == "rober" else (UM.ring(int(name[4:])), "RING", (int(name[4:]), 0, 0, 0))
    n, npar = m["n"], m["np"]
    alg, oalg = ALGS[int(rng.integers(4))]
    stepper = "rk4" if rng.random() < 0.5 else "tsit5"
    use_mm, use_ev, auto = bool(rng.random() < 0.6), bool(rng.random() < 0.6), bool(rng.random() < 0.4)
    shared = bool(rng.random() < 0.5)
    key = f"{name}_mmev_{int(auto)}"
    if key not in _reg:
        _reg[key] = sa.DeviceFunction(key, n, npar, m["f"], *(() if auto else (m["vjp"], m["vjp_p"]))).set_affect(AFFECT)
    f = _reg[key]
    M = (np.eye(n) * 1.5 + 0.3 * rng.standard_normal((n, n))) if use_mm else None
    f.set_mass_matrix(M)
    N, T = int(rng.integers(1, 90)), 2.0
    u0 = rng.uniform(0.3, 1.0, (N, n))
    pp = rng.uniform(0.4, 1.2, npar) if shared else rng.uniform(0.4, 1.2, (N, npar))
    ts = np.unique(np.round(rng.uniform(0.05, T, int(rng.integers(1, 6))), 2)); ts = np.unique(np.concatenate([ts, [T]]))
    events = list(np.unique(np.round(rng.uniform(0.2, 1.8, int(rng.integers(1, 3))), 1))) if use_ev else []
    delta = rng.standard_normal((N, len(ts), n))
    if stepper == "rk4":
        salg, kw, okw = sa.RK4(), dict(dt=0.01), dict(stepper="RK4", dt=0.01)
    else:
        salg, kw, okw = sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10)
    sens = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint(), "quadrature": sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12)}[alg]
    prob = sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), pp if shared else pp[0], dims), u0, pp)
    sol = sa.solve(prob, salg, saveat=ts, sensealg=sens, **({"callback": sa.PresetTimeCallback(events)} if use_ev else {}), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, dgdu_discrete=delta)
    ctx = O.mass_matrix(M) if use_mm else None
    if ctx:
        ctx.__enter__()
    try:
        rout, rdu0, rdp = oracle_chain(omodel, dims, n, npar, events, ts, T, u0, pp, delta, oalg, okw, shared, M)
    finally:
        if ctx:
            ctx.__exit__(None, None, None)
    msg = dict(name=name, alg=alg, stepper=stepper, mm=use_mm, events=events, auto=auto, shared=shared, N=N, ts=ts.tolist())
    assert rel(sol.u, rout) < 1e-7 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6, msg
    (sol.close() if use_ev else sol.engine.close())
    f.set_mass_matrix(None)
