"""DiscreteCallback at preset times on a WIDE runtime model (`-m gpu`; hipadj_wmodel_set_affect, round 5): the cases of tests/test_gpu_events.py — a constant dose, a state- and
parameter-dependent affect, a parameter-changing affect, several event times — on a 12-state ring of the workgroup-per-trajectory family (traced joint VJP, wtrace.py), the affect
and its reverse callback as serial device text, against the same chain composed from the ORACLE's per-piece adjoints with the affect and its VJP written in numpy
(test/Callbacks1/discrete_callbacks.jl:260-330; src/callback_tracking.jl:232-470)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
_reg = {}
NR = 12
NP = NR + 1

AFFECTS = {
    # name: (affect text, reverse-callback text, numpy affect (u, p) -> (un, pn), numpy reverse callback (u, p, lam, gp) -> (lam_out, gp_out))
    "dose": ("un[0] += 2.0;", "", lambda u, p: (u + np.eye(1, NR)[0] * 2.0, p), lambda u, p, l, g: (l.copy(), g.copy())),
    "sin": ("for (int i = 0; i < N; ++i) un[i] += p[1] / 8.0 * sin(u[i]);",
            "for (int i = 0; i < N; ++i) { lo[i] = lam[i] * (1.0 + p[1] / 8.0 * cos(u[i])); go[1] += lam[i] * sin(u[i]) / 8.0; }",
            lambda u, p: (u + p[:, 1:2] / 8.0 * np.sin(u), p),
            lambda u, p, l, g: (l * (1.0 + p[:, 1:2] / 8.0 * np.cos(u)), g + np.eye(1, NP, 1)[0][None, :] * ((l * np.sin(u)).sum(axis=1) / 8.0)[:, None])),
    "pchange": ("for (int k = 0; k < NP; ++k) pn[k] = 1.1 * p[k] - 0.02; un[1] += 0.1 * p[3] * u[0];",
                "lo[0] += 0.1 * p[3] * lam[1]; for (int k = 0; k < NP; ++k) go[k] = 1.1 * gp[k]; go[3] += 0.1 * u[0] * lam[1];",
                lambda u, p: (u + np.eye(1, NR, 1)[0][None, :] * (0.1 * p[:, 3] * u[:, 0])[:, None], 1.1 * p - 0.02),
                lambda u, p, l, g: (l + np.eye(1, NR)[0][None, :] * (0.1 * p[:, 3] * l[:, 1])[:, None], 1.1 * g + np.eye(1, NP, 3)[0][None, :] * (0.1 * u[:, 0] * l[:, 1])[:, None])),
}


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def fun(sa, name):
    if name not in _reg:
        def ring(u, p, t, ops):
            n = u.length
            return p[0:n] * (ops.roll(u, -1) - u) + p[n] * ops.sin(ops.roll(u, 1))
        f = sa.WideDeviceFunction.from_callable("ring_event_" + name, ring, NR, NP)
        f.set_affect(AFFECTS[name][0], AFFECTS[name][1])
        _reg[name] = f
    return _reg[name]


def oracle_chain(name, events, ts, T, u0, pp, delta, alg, okw, shared):
    aff, vjp = AFFECTS[name][2], AFFECTS[name][3]
    N = len(u0)
    P = np.ascontiguousarray(np.broadcast_to(pp, (N, NP)))
    ev = sorted(e for e in events if 0.0 < e < T and e <= ts[-1])
    edges = [0.0] + ev + [T]
    pieces, u, out, ul = [], u0, np.zeros((N, len(ts), NR)), []
    for j in range(len(edges) - 1):
        a, b = edges[j], edges[j + 1]; last = j == len(edges) - 2
        own = [i for i, s in enumerate(ts) if (a <= s < b) or (last and s == b)]
        sv = np.array([ts[i] for i in own] + ([] if last else [b]))
        pr = O.Problem("RING", alg=alg, t0=a, t1=b, save_times=sv, loss="COTANGENT", checkpointing=(alg == "BACKSOLVE"), quad_abstol=1e-12, quad_reltol=1e-12, dims=(NR, 0, 0, 0), **okw)
        _, _, o, _ = pr.adjoint_ensemble(u, P, np.zeros((N, len(sv), NR)))
        pieces.append((pr, own, u.copy(), P.copy()))
        for q, i in enumerate(own):
            out[:, i] = o[:, q]
        if not last:
            ul.append(o[:, -1].copy()); u, P = aff(o[:, -1], P); u = np.ascontiguousarray(u); P = np.ascontiguousarray(P)
    gp = np.zeros((N, NP)); lam_in = None; du0 = None
    for j in range(len(pieces) - 1, -1, -1):
        pr, own, ustart, Pj = pieces[j]
        cot = [delta[:, i] for i in own] + ([lam_in] if j < len(pieces) - 1 else [])
        du0, dpj, _, _ = pr.adjoint_ensemble(ustart, Pj, np.ascontiguousarray(np.stack(cot, axis=1)))
        gp = gp + dpj
        if j > 0:
            lam_in, gp = vjp(ul[j - 1], pieces[j - 1][3], du0, gp)
    return out, du0, (gp.sum(axis=0) if shared else gp)


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
@pytest.mark.parametrize("name,events", [("dose", [1.0]), ("sin", [0.5, 1.5]), ("pchange", [1.0])])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_wide_model_discrete_callback_matches_oracle_chain(sa, alg, oalg, name, events, stepper):
    rng = np.random.default_rng(71)
    N, T = 5, 2.0
    shared = name != "sin"
    u0 = rng.uniform(0.3, 1.0, (N, NR))
    pp = rng.uniform(0.2, 0.6, NP) if shared else rng.uniform(0.2, 0.6, (N, NP))
    ts = np.arange(0.0, T + 1e-9, 0.25)
    delta = rng.standard_normal((N, len(ts), NR))
    f = fun(sa, name)
    if stepper == "rk4":
        salg, kw, okw = sa.RK4(), dict(dt=0.01), dict(stepper="RK4", dt=0.01)
    else:
        salg, kw, okw = sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10)
    sens = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint(), "quadrature": sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12)}[alg]
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), pp if shared else pp[0]), u0, pp), salg, saveat=ts, sensealg=sens,
                   callback=sa.PresetTimeCallback(events), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, dgdu_discrete=delta)
    rout, rdu0, rdp = oracle_chain(name, events, ts, T, u0, pp, delta, oalg, okw, shared)
    assert rel(sol.u, rout) < 1e-8 and rel(du0, rdu0) < 1e-7 and rel(dp, rdp) < 1e-7
    sol.close()


def test_wide_affect_setter_rules(sa):
    f = fun(sa, "dose")
    with pytest.raises(sa.HipadjError):
        from scimlsensitivity_jl_amd import _lib
        _lib.set_model_affect(f.id, "un[0] += 1.0;")            # the lane setter refuses a wide model and names the right entry point
