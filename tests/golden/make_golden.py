"""Generate tests/golden/*.json — independent high-accuracy gradients for the oracle to be pinned against.

The reference (pure Julia) cannot run in this image, and its tests for this path assert RELATIONS
(adjoint == ForwardDiff-through-the-solver == explicit quadgk), not literal vectors.
This script plays the role ForwardDiff plays in those tests: it integrates the forward-sensitivity
system dS/dt = J S + df/dtheta with scipy DOP853 at rtol = atol = 1e-13 and contracts with dL/du at the
loss times.  Model right-hand sides are restated here in numpy, independently of oracle/ and csrc/.

    python tests/golden/make_golden.py        (needs scipy; ~1 minute)

Problems (reference anchors):
  lvt     time-dependent Lotka-Volterra, dg = u - 2 at t = 0:0.5:10      test/Core3/adjoint.jl:8-51
  lv_sum  Lotka-Volterra, loss = sum(solve(...; saveat = 0.1))            test/Core1/concrete_solve_derivatives.jl:106-165
  lorenz  Lorenz-63, dg = u - 2 at t = 0:0.1:10                           test/Core3/adjoint.jl:1157-1172
  fallmass / lindiag: the two literal known answers of the reference tests
          [-27.675, 0.0] (test/Core7/physical_ode_regression.jl:42-51), exp.(p) (test/Core1/sparse_adjoint.jl:32-33)
"""
import json
import os
import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))


def lv(u, p, t):
    f = np.array([p[0] * u[0] - p[1] * u[0] * u[1], -p[2] * u[1] + p[3] * u[0] * u[1]])
    J = np.array([[p[0] - p[1] * u[1], -p[1] * u[0]], [p[3] * u[1], -p[2] + p[3] * u[0]]])
    P = np.array([[u[0], -u[0] * u[1], 0, 0], [0, 0, -u[1], u[0] * u[1]]])
    return f, J, P


def lvt(u, p, t):
    f = np.array([p[0] * u[0] - p[1] * u[0] * u[1] * t, -p[2] * u[1] + t * p[3] * u[0] * u[1]])
    J = np.array([[p[0] - p[1] * u[1] * t, -p[1] * u[0] * t], [t * p[3] * u[1], -p[2] + t * p[3] * u[0]]])
    P = np.array([[u[0], -u[0] * u[1] * t, 0, 0], [0, 0, -u[1], t * u[0] * u[1]]])
    return f, J, P


def lorenz(u, p, t):
    f = np.array([p[0] * (u[1] - u[0]), u[0] * (p[1] - u[2]) - u[1], u[0] * u[1] - p[2] * u[2]])
    J = np.array([[-p[0], p[0], 0], [p[1] - u[2], -1, -u[0]], [u[1], u[0], -p[2]]])
    P = np.array([[u[1] - u[0], 0, 0], [0, u[0], 0], [0, 0, -u[2]]])
    return f, J, P


def fallmass(u, p, t):
    return np.array([u[1], -p[0]]), np.array([[0, 1.0], [0, 0]]), np.array([[0, 0], [-1.0, 0]])


def lindiag(u, p, t):
    return p * u, np.diag(p), np.diag(u)


def gradient(model, u0, p, tspan, ts, dgdu, rtol=1e-13, atol=1e-13):
    """dL/du0, dL/dp for L = sum_i l_i(u(t_i)) with dl_i/du = dgdu(u, i)."""
    n, npar = len(u0), len(p)
    nth = n + npar

    def rhs(t, z):
        u = z[:n]
        S = z[n:].reshape(n, nth)
        f, J, P = model(u, p, t)
        dS = J @ S
        dS[:, n:] += P
        return np.concatenate([f, dS.ravel()])

    S0 = np.zeros((n, nth))
    S0[:, :n] = np.eye(n)
    z0 = np.concatenate([np.asarray(u0, float), S0.ravel()])
    sol = solve_ivp(rhs, tspan, z0, method="DOP853", rtol=rtol, atol=atol, t_eval=ts)
    assert sol.success
    g = np.zeros(nth)
    us = []
    for i in range(len(ts)):
        u = sol.y[:n, i]
        S = sol.y[n:, i].reshape(n, nth)
        g += dgdu(u, i) @ S
        us.append(u.tolist())
    return g[:n], g[n:], us


def main():
    out = {}
    ts = np.arange(0, 10.0001, 0.5)
    du0, dp, us = gradient(lvt, [1.0, 1.0], np.array([1.5, 1.0, 3.0, 1.0]), (0, 10), ts, lambda u, i: u - 2.0)
    out["lvt"] = dict(u0=[1.0, 1.0], p=[1.5, 1.0, 3.0, 1.0], tspan=[0, 10], ts=ts.tolist(), loss="lsq_shift 2.0",
                      du0=du0.tolist(), dp=dp.tolist(), u=us)

    ts = np.linspace(0, 10, 101)
    du0, dp, us = gradient(lv, [1.0, 1.0], np.array([1.5, 1.0, 3.0, 1.0]), (0, 10), ts, lambda u, i: np.ones(2))
    out["lv_sum"] = dict(u0=[1.0, 1.0], p=[1.5, 1.0, 3.0, 1.0], tspan=[0, 10], ts=ts.tolist(), loss="sum",
                         du0=du0.tolist(), dp=dp.tolist(), u=us)

    ts = np.linspace(0, 10, 101)
    du0, dp, us = gradient(lorenz, [1.0, 0.0, 0.0], np.array([10.0, 28.0, 8.0 / 3.0]), (0, 10), ts,
                           lambda u, i: u - 2.0)
    out["lorenz"] = dict(u0=[1.0, 0.0, 0.0], p=[10.0, 28.0, 8.0 / 3.0], tspan=[0, 10], ts=ts.tolist(),
                         loss="lsq_shift 2.0", du0=du0.tolist(), dp=dp.tolist(), u=us)

    # shorter-horizon Lorenz (T = 2): well-conditioned, used for tight comparisons
    ts = np.linspace(0, 2, 21)
    du0, dp, us = gradient(lorenz, [1.0, 0.0, 0.0], np.array([10.0, 28.0, 8.0 / 3.0]), (0, 2), ts,
                           lambda u, i: u - 2.0)
    out["lorenz_T2"] = dict(u0=[1.0, 0.0, 0.0], p=[10.0, 28.0, 8.0 / 3.0], tspan=[0, 2], ts=ts.tolist(),
                            loss="lsq_shift 2.0", du0=du0.tolist(), dp=dp.tolist(), u=us)

    # literal known answers of the reference's own tests
    ts = np.arange(0, 2.0001, 0.05)
    du0, dp, us = gradient(fallmass, [1.0, 0.0], np.array([9.81, 1.0]), (0, 2), ts, lambda u, i: np.array([1.0, 0.0]))
    out["fallmass"] = dict(u0=[1.0, 0.0], p=[9.81, 1.0], tspan=[0, 2], ts=ts.tolist(), loss="sum of u[1]",
                           reference_literal=[-27.675, 0.0], reference_atol=1e-2, du0=du0.tolist(), dp=dp.tolist())
    ts = np.array([1.0])
    du0, dp, us = gradient(lindiag, [1.0, 1.0], np.array([1.0, 2.0]), (0, 1), ts, lambda u, i: np.ones(2))
    out["lindiag"] = dict(u0=[1.0, 1.0], p=[1.0, 2.0], tspan=[0, 1], ts=ts.tolist(), loss="sum(u(T))",
                          reference_literal=np.exp([1.0, 2.0]).tolist(), reference_rtol=1e-3,
                          du0=du0.tolist(), dp=dp.tolist())
    # continuous cost L = int_0^T (u1 + u2)^2 / 2 dt on the time-dependent LV problem (test/Core3/adjoint.jl:910-1127)
    p4 = np.array([1.5, 1.0, 3.0, 1.0])

    def rhs_c(t, z):
        u = z[:2]; S = z[2:14].reshape(2, 6)
        f_, J, P = lvt(u, p4, t)
        dS = J @ S; dS[:, 2:] += P
        return np.concatenate([f_, dS.ravel(), u.sum() * (S[0] + S[1])])
    S0 = np.zeros((2, 6)); S0[:, :2] = np.eye(2)
    solc = solve_ivp(rhs_c, (0, 4.0), np.concatenate([[1.0, 1.0], S0.ravel(), np.zeros(6)]), method="DOP853", rtol=1e-13, atol=1e-13)
    gc = solc.y[14:, -1]
    out["lvt_continuous"] = dict(u0=[1.0, 1.0], p=p4.tolist(), tspan=[0, 4.0], cost="g = (u1+u2)^2/2", du0=gc[:2].tolist(), dp=gc[2:].tolist())

    # mixed continuous cost with a parameter term, G = int_0^10 u1^2 + p1 dt on LV (test/Core7/mixed_costs.jl:13-57:
    # `g(u, p, t) = u[1]^2 + p[1]`, dgdu = [2 u1, 0], dgdp = [1, 0, 0, 0]; the reference compares with ForwardDiff of quadgk)
    def rhs_m(t, z):
        u = z[:2]; S = z[2:14].reshape(2, 6)
        f_, J, P = lv(u, p4, t)
        dS = J @ S; dS[:, 2:] += P
        dG = 2.0 * u[0] * S[0]; dG[2] += 1.0
        return np.concatenate([f_, dS.ravel(), dG])
    solm = solve_ivp(rhs_m, (0, 10.0), np.concatenate([[1.0, 1.0], S0.ravel(), np.zeros(6)]), method="DOP853", rtol=1e-13, atol=1e-13)
    gm = solm.y[14:, -1]
    out["lv_mixed_cost"] = dict(u0=[1.0, 1.0], p=p4.tolist(), tspan=[0, 10.0], cost="g = u1^2 + p1", du0=gm[:2].tolist(), dp=gm[2:].tolist())

    with open(os.path.join(HERE, "gradients.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, "du0", v["du0"], "dp", v["dp"])


if __name__ == "__main__":
    main()
