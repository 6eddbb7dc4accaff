"""Randomised differential test of the workgroup-per-trajectory family (`-m gpu`): model x stepper x sensealg (the four + GaussKronrod) x loss form x loss-time pattern x
parameter sharing x continuous cost x no_start, device through the C ABI vs the CPU oracle on the same seeded inputs.

The patterns aim at the edges of the callback / tstop logic the two steppers restate: no loss time at all (only a continuous cost drives the
adjoint), a loss time only at T (PresetTimeCallback at initialisation), only at t0, loss times at both ends, one trajectory, a workgroup wider than
the state (n = 3 on 64 lanes) and narrower (n = 130 on 64 lanes: three components per lane with a ragged last row)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE"), ("gausskronrod", "GAUSS_KRONROD")]
_reg = {}


def _model(sa, which):
    if which not in _reg:
        if which == "linear":
            _reg[which] = (sa.WideDeviceFunction.dense_linear("fz_lin8", 8), "DENSELIN", (8, 0, 0, 0), 8, 64)
        elif which == "index":
            _reg[which] = (sa.WideDeviceFunction.index_affine("fz_idx_13x10", 13, 10), "IDXAFF", (13, 10, 0, 0), 130, 2)
        else:
            _reg[which] = (sa.WideDeviceFunction.dense_chain("fz_chain_3_8_3", (3, 8, 3), input_power=3), "MLP1", (3, 8, 0, 0), 3, 59)
    return _reg[which]


@pytest.mark.parametrize("seed", range(60))
def test_random_wide_configurations(sa, seed):
    rng = np.random.default_rng(9100 + seed)
    which = ["linear", "index", "chain"][int(rng.integers(3))]
    fun, oname, dims, n, npar = _model(sa, which)
    alg, oalg = ALGS[int(rng.integers(5))]
    adaptive = bool(rng.random() < 0.5)
    T = 1.0
    N = int(rng.choice([1, 2, 5]))
    shared = bool(rng.random() < 0.6)
    cost = int(rng.choice([0, 0, 1, 2]))
    lsq = bool(rng.random() < 0.4)
    dt = 0.02
    pattern = int(rng.integers(6))
    if adaptive:
        inner = np.unique(np.round(rng.uniform(0.03, 0.97, int(rng.integers(1, 5))), 3))
    else:
        inner = np.unique(np.round(rng.uniform(0.03, 0.97, int(rng.integers(1, 5))) / dt) * dt)
    ts = {0: np.array([]), 1: np.array([T]), 2: np.array([0.0]), 3: np.concatenate([[0.0], inner, [T]]), 4: inner, 5: np.concatenate([inner, [T]])}[pattern]
    if len(ts) == 0 and cost == 0:
        cost = 1                                       # something has to drive the adjoint
    no_start = bool(len(ts) > 0 and ts[0] == 0.0 and rng.random() < 0.5)
    if which == "linear":
        mk = lambda: (rng.standard_normal((8, 8)) / np.sqrt(8) - 0.5 * np.eye(8)).flatten(order="F")
    elif which == "index":
        mk = lambda: 0.3 * rng.random(2)
    else:
        mk = lambda: np.concatenate([rng.standard_normal(24) * 0.4, 0.1 * rng.standard_normal(8), rng.standard_normal(24) * 0.3, 0.1 * rng.standard_normal(3)])
    p = mk() if shared else np.stack([mk() for _ in range(N)])
    u0 = 0.6 * rng.standard_normal((N, n))
    g = {0: None, 1: sa.HalfSquaredSum(), 2: sa.FirstStateSquaredPlusFirstParam()}[cost]
    sens = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(checkpointing=bool(rng.random() < 0.5)), "gauss": sa.GaussAdjoint(),
            "quadrature": sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11), "gausskronrod": sa.GaussKronrodAdjoint()}[alg]
    if adaptive:
        salg, kw, okw = sa.Tsit5(), dict(abstol=1e-9, reltol=1e-9), dict(stepper="TSIT5", dt=0.0, abstol=1e-9, reltol=1e-9)
    else:
        salg, kw, okw = sa.RK4(), dict(dt=dt), dict(stepper="RK4", dt=dt)
    delta = rng.standard_normal((N, len(ts), n))
    prob = sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p if shared else p[0]), u0, None if shared else p)
    skw = dict(g=g) if g is not None else {}
    sol = sa.solve(prob, salg, saveat=(ts if len(ts) else None), sensealg=sens, no_start=no_start, save_everystep=False,
                   **(dict(dgdu_discrete=sa.LsqShift(0.25)) if lsq else {}), **skw, **kw)
    if lsq or len(ts) == 0:
        du0, dp = sa.adjoint_sensitivities(sol, salg, **(dict(t=ts, dgdu_discrete=sa.LsqShift(0.25)) if (lsq and len(ts)) else {}), **skw)
    else:
        du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta, **skw)
    out = sol.u
    sol.engine.close()
    ref = O.Problem(oname, alg=oalg, t0=0.0, t1=T, save_times=ts, dims=dims, checkpointing=(oalg == "BACKSOLVE" and sens.checkpointing), no_start=no_start,
                    quad_abstol=1e-11, quad_reltol=1e-11, cont_cost=cost, **(dict(loss="LSQ_SHIFT", loss_shift=0.25) if (lsq or len(ts) == 0) else dict(loss="COTANGENT")), **okw)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, None if (lsq or len(ts) == 0) else delta)
    msg = dict(which=which, alg=alg, adaptive=adaptive, N=N, shared=shared, cost=cost, lsq=lsq, ts=ts.tolist(), no_start=no_start, ckpt=getattr(sens, "checkpointing", None))
    tol = 1e-7 if adaptive else 1e-9

    def relz(a, b):                                   # a gradient may be exactly zero (a loss time at t0 only, suppressed by no_start, no cost)
        return float(np.max(np.abs(np.asarray(a) - b))) / max(float(np.max(np.abs(b))), 1e-12)
    if len(ts):
        assert relz(out, rout) < tol, msg
    assert relz(du0, rdu0) < tol and relz(dp, rdp) < tol, msg
