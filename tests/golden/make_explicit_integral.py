"""Generate tests/golden/explicit_integral.json — the reference's own pin for this path, restated independently with scipy.

test/Core3/adjoint.jl:352-404 checks EVERY sensealg x VJP combination against one number that is computed without any of them:

    adj_sol  = solve(ODEAdjointProblem(sol, QuadratureAdjoint(1e-14, 1e-14), Tsit5(), t, dg), Tsit5(), abstol = reltol = 1e-14)
    res, err = quadgk(AdjointSensitivityIntegrand(sol, adj_sol, ...), 0.0, 10.0, atol = 1e-14, rtol = 1e-12)      # = int lam(t)^T f_p(u(t), p, t) dt
    @test isapprox(res, easy_res*, rtol = 1e-9 ... 1e-10)

i.e. dL/dp = int_0^T lam^T (df/dp) dt with lam' = -(df/du)^T lam integrated backward from lam(T+) = 0 through the jumps
lam(t_i-) = lam(t_i+) + dg(u(t_i)) (docs/src/sensitivity_math.md:72-138).  This script builds exactly that number with scipy only
(DOP853 dense forward solution, DOP853 backward pieces between the loss times, `quad` of the integrand per loss interval and
component) — no code of oracle/ or csrc/ takes part — for the problems the reference tests use:

  lvt      time-dependent Lotka-Volterra `fb`, dg = u - 2 at t = 0:0.5:10      test/Core3/adjoint.jl:8-51, 352-404
  lorenz   Lorenz-63, dg = u - 2 at t = 0:0.1:2 (the horizon on which 1e-9 is attainable; the reference's Lorenz test at T = 10 only
           compares Backsolve with Interpolating at 1e-5, :1201-1241)
  lv       Lotka-Volterra with loss = sum(sol) at saveat = 0.5                  test/Core1/concrete_solve_derivatives.jl:106-165

    python tests/golden/make_explicit_integral.py        (scipy; ~1 minute)
"""
import json
import os

import numpy as np
from scipy.integrate import quad, solve_ivp

from make_golden import lorenz, lv, lvt

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-13


def explicit_adjoint_integral(model, u0, p, tspan, ts, dgdu):
    """(du0, dp) = (lam(t0), int lam^T f_p dt) by the continuous adjoint, pieces between the loss times."""
    n, npar = len(u0), len(p)
    t0, t1 = tspan
    fwd = solve_ivp(lambda t, u: model(u, p, t)[0], tspan, np.asarray(u0, float), method="DOP853", rtol=TOL, atol=TOL, dense_output=True)
    assert fwd.success
    u_of = fwd.sol

    def lam_rhs(t, lam):
        return -(model(u_of(t), p, t)[1].T @ lam)

    # breakpoints of the backward sweep: T, the loss times (descending), t0
    stops = sorted(set([t0, t1] + [float(t) for t in ts]), reverse=True)
    loss_at = {float(t): i for i, t in enumerate(ts)}
    lam = np.zeros(n)
    if t1 in loss_at:
        lam = lam + dgdu(u_of(t1), loss_at[t1])
    dp = np.zeros(npar)
    for a, b in zip(stops[:-1], stops[1:]):          # integrate lam from a down to b
        piece = solve_ivp(lam_rhs, (a, b), lam, method="DOP853", rtol=TOL, atol=TOL, dense_output=True)
        assert piece.success
        for j in range(npar):
            val, err = quad(lambda t: float(model(u_of(t), p, t)[2][:, j] @ piece.sol(t)), b, a, epsabs=1e-13, epsrel=1e-13, limit=400)
            dp[j] += val
        lam = piece.y[:, -1]
        if b in loss_at:
            lam = lam + dgdu(u_of(b), loss_at[b])
    return lam, dp


def main():
    out = {}
    p4 = np.array([1.5, 1.0, 3.0, 1.0])
    ts = np.arange(0, 10.0001, 0.5)
    du0, dp = explicit_adjoint_integral(lvt, [1.0, 1.0], p4, (0.0, 10.0), ts, lambda u, i: u - 2.0)
    out["lvt"] = dict(model="lvt", u0=[1.0, 1.0], p=p4.tolist(), tspan=[0.0, 10.0], ts=ts.tolist(), loss="lsq_shift 2.0", du0=du0.tolist(), dp=dp.tolist(),
                      anchor="test/Core3/adjoint.jl:352-404")
    p3 = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.linspace(0, 2, 21)
    du0, dp = explicit_adjoint_integral(lorenz, [1.0, 0.0, 0.0], p3, (0.0, 2.0), ts, lambda u, i: u - 2.0)
    out["lorenz_T2"] = dict(model="lorenz", u0=[1.0, 0.0, 0.0], p=p3.tolist(), tspan=[0.0, 2.0], ts=ts.tolist(), loss="lsq_shift 2.0", du0=du0.tolist(), dp=dp.tolist(),
                            anchor="test/Core3/adjoint.jl:1157-1172 (setup), :352-404 (relation)")
    ts = np.arange(0, 10.0001, 0.5)
    du0, dp = explicit_adjoint_integral(lv, [1.0, 1.0], p4, (0.0, 10.0), ts, lambda u, i: np.ones(2))
    out["lv_sum"] = dict(model="lv", u0=[1.0, 1.0], p=p4.tolist(), tspan=[0.0, 10.0], ts=ts.tolist(), loss="sum", du0=du0.tolist(), dp=dp.tolist(),
                         anchor="test/Core1/concrete_solve_derivatives.jl:106-165")
    with open(os.path.join(HERE, "explicit_integral.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, "du0", v["du0"], "dp", v["dp"])


if __name__ == "__main__":
    main()
