#!/usr/bin/env python
"""Golden gradients for the stiff-stepper cases (Rosenbrock23), computed INDEPENDENTLY of the oracle by forward sensitivities S = du/d(p, u0) integrated
with scipy (rtol = atol of 1e-12 .. 1e-13) — the role ForwardDiff-through-RadauIIA5 plays in /root/reference/test/Core2/stiff_adjoints.jl:30-48.

  lv      the Lotka-Volterra fit of test/Core2/stiff_adjoints.jl:10-48, 66-80: u0 = [1, 1], tspan (0, 10), saveat 0:0.5:10, target data = the trajectory of
          p0 = [1.5, 1.0, 3.0, 1.0], loss(p) = sum(abs2, prediction - target) at p = [1.5, 1.2, 1.4, 1.6]; the reference asks Rosenbrock23 (abstol = reltol = 1e-8)
          to agree with the ForwardDiff gradient to rtol 1e-3 (in place, :80) and 1e-4 (out of place, :157).  (The reference's target comes from RadauIIA5 at its default
          tolerances; here it is the converged trajectory — the data are an input of the test either way and travel in the fixture.)
  rober   Robertson kinetics in ODE form (`rober` of test/Core3/adjoint.jl:1434-1441 with its third row the rate equation instead of the conservation constraint),
          p = [0.04, 3e7, 1e4] (:1458), u0 = [1, 0, 0], tspan (0, 100), G = y3(50) + y3(100) (ts and dg_singular of :1465-1466): stiffness ratio ~ 1e4 .. 1e9 — the
          problem class the stepper exists for.  Radau with the analytic Jacobian of the augmented system.

Writes tests/golden/stiff_adjoints.json.  Needs numpy + scipy only."""
import json
import os

import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))


def lv():
    u0 = np.array([1.0, 1.0]); p0 = np.array([1.5, 1.0, 3.0, 1.0]); p = np.array([1.5, 1.2, 1.4, 1.6])
    ts = np.arange(0.0, 10.0 + 1e-12, 0.5)

    def rhs(q):
        def f(t, z):
            x, y = z[0], z[1]
            S = z[2:].reshape(2, 6)                      # columns: d/dp (4), d/du0 (2)
            fu = np.array([q[0] * x - q[1] * x * y, -q[2] * y + q[3] * x * y])
            J = np.array([[q[0] - q[1] * y, -q[1] * x], [q[3] * y, -q[2] + q[3] * x]])
            fp = np.zeros((2, 6)); fp[0, 0] = x; fp[0, 1] = -x * y; fp[1, 2] = -y; fp[1, 3] = x * y
            return np.concatenate([fu, (J @ S + fp).ravel()])
        return f
    z0 = np.zeros(2 + 12); z0[:2] = u0; S0 = np.zeros((2, 6)); S0[0, 4] = 1.0; S0[1, 5] = 1.0; z0[2:] = S0.ravel()
    tgt = solve_ivp(rhs(p0), (0.0, 10.0), z0, method="DOP853", rtol=1e-13, atol=1e-13, t_eval=ts).y[:2].T
    sol = solve_ivp(rhs(p), (0.0, 10.0), z0, method="DOP853", rtol=1e-13, atol=1e-13, t_eval=ts)
    u = sol.y[:2].T; S = sol.y[2:].T.reshape(len(ts), 2, 6)
    r = u - tgt
    loss = float((r * r).sum())
    g = 2.0 * np.einsum("mi,mik->k", r, S)
    return dict(u0=u0.tolist(), p=p.tolist(), p0=p0.tolist(), tspan=[0.0, 10.0], ts=ts.tolist(), target=tgt.tolist(), loss=loss, dp=g[:4].tolist(), du0=g[4:].tolist())


def rober():
    p = np.array([0.04, 3.0e7, 1.0e4]); u0 = np.array([1.0, 0.0, 0.0]); ts = [50.0, 100.0]

    def parts(z):
        y1, y2, y3 = z[0], z[1], z[2]
        fu = np.array([-p[0] * y1 + p[2] * y2 * y3, p[0] * y1 - p[1] * y2 * y2 - p[2] * y2 * y3, p[1] * y2 * y2])
        J = np.array([[-p[0], p[2] * y3, p[2] * y2], [p[0], -2.0 * p[1] * y2 - p[2] * y3, -p[2] * y2], [0.0, 2.0 * p[1] * y2, 0.0]])
        fp = np.zeros((3, 6)); fp[0, 0] = -y1; fp[1, 0] = y1; fp[1, 1] = -y2 * y2; fp[2, 1] = y2 * y2; fp[0, 2] = y2 * y3; fp[1, 2] = -y2 * y3
        return fu, J, fp

    def f(t, z):
        fu, J, fp = parts(z)
        S = z[3:].reshape(3, 6)
        return np.concatenate([fu, (J @ S + fp).ravel()])
    z0 = np.zeros(3 + 18); z0[:3] = u0; S0 = np.zeros((3, 6)); S0[0, 3] = S0[1, 4] = S0[2, 5] = 1.0; z0[3:] = S0.ravel()
    out = {}
    for tol in (1e-10, 1e-12):
        sol = solve_ivp(f, (0.0, 100.0), z0, method="Radau", rtol=tol, atol=tol * 1e-4, t_eval=ts)
        S = sol.y[3:].T.reshape(2, 3, 6)
        out[tol] = (sol.y[:3].T, S[0, 2] + S[1, 2])
    u, g = out[1e-12]
    spread = float(np.max(np.abs(out[1e-10][1] - g) / np.maximum(np.abs(g), 1e-300)))
    return dict(u0=u0.tolist(), p=p.tolist(), tspan=[0.0, 100.0], ts=ts, u_at_ts=u.tolist(), G=float(u[0, 2] + u[1, 2]), dp=g[:3].tolist(), du0=g[3:].tolist(),
                spread_between_tolerances=spread)


def rober_dae_kappa(kappa=5.0):
    """The semi-explicit DAE of test/Core3/adjoint.jl:1434-1454 with a constraint that depends on a parameter — y1 + y2 + y3 = 1 + kappa (p1 - 0.04), NOT from the reference: with
    the reference's own constraint the parameter term of the loss jumps, f_p' [0; dlam_a] (src/adjoint_common.jl:803, src/sensitivity_interface.jl:510-521), is identically zero
    and no test could see it missing.  Reduced to the ODE in (y1, y2) with y3 = c(p) - y1 - y2; forward sensitivities with respect to p and (y1(0), y2(0)); G = y3(50) + y3(100)."""
    p = np.array([0.04, 3.0e7, 1.0e4]); ts = [50.0, 100.0]
    dc = np.array([kappa, 0.0, 0.0, 0.0, 0.0])                  # d c / d (p1, p2, p3, y1_0, y2_0)

    def f(t, z):
        y1, y2 = z[0], z[1]
        S = z[2:].reshape(2, 5)
        y3 = 1.0 + kappa * (p[0] - 0.04) - y1 - y2
        dy3 = dc - S[0] - S[1]
        fu = np.array([-p[0] * y1 + p[2] * y2 * y3, p[0] * y1 - p[1] * y2 * y2 - p[2] * y2 * y3])
        J = np.array([[-p[0], p[2] * y3], [p[0], -2.0 * p[1] * y2 - p[2] * y3]])
        d3 = np.array([p[2] * y2, -p[2] * y2])                   # d f / d y3
        fp = np.zeros((2, 5)); fp[0, 0] = -y1; fp[1, 0] = y1; fp[1, 1] = -y2 * y2; fp[0, 2] = y2 * y3; fp[1, 2] = -y2 * y3
        return np.concatenate([fu, (J @ S + np.outer(d3, dy3) + fp).ravel()])
    z0 = np.zeros(2 + 10); z0[0] = 1.0; S0 = np.zeros((2, 5)); S0[0, 3] = 1.0; S0[1, 4] = 1.0; z0[2:] = S0.ravel()
    res = {}
    for tol in (1e-10, 1e-12):
        sol = solve_ivp(f, (0.0, 100.0), z0, method="Radau", rtol=tol, atol=tol * 1e-4, t_eval=ts)
        S = sol.y[2:].T.reshape(2, 2, 5)
        y3 = 1.0 - sol.y[0] - sol.y[1]
        res[tol] = (np.stack([sol.y[0], sol.y[1], y3], axis=1), (dc - S[0, 0] - S[0, 1]) + (dc - S[1, 0] - S[1, 1]))
    u, g = res[1e-12]
    return dict(kappa=kappa, u0=[1.0, 0.0, 0.0], p=p.tolist(), tspan=[0.0, 100.0], ts=ts, u_at_ts=u.tolist(), G=float(u[0, 2] + u[1, 2]), dp=g[:3].tolist(), du0_differential=g[3:].tolist(),
                spread_between_tolerances=float(np.max(np.abs(res[1e-10][1] - g) / np.maximum(np.abs(g), 1e-300))))


if __name__ == "__main__":
    out = dict(lv=lv(), rober=rober(), rober_dae_kappa=rober_dae_kappa(), source="tests/golden/make_stiff_adjoints.py: scipy forward sensitivities (DOP853 1e-13 on lv; Radau 1e-12 on rober)")
    json.dump(out, open(os.path.join(HERE, "stiff_adjoints.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "target"} if isinstance(v, dict) else v for k, v in out.items()}, indent=1))
