"""Generate tests/golden/discrete_losses.json — independent gradients for discrete losses that depend on the parameters and on a data block
(dgdu_discrete AND dgdp_discrete, src/adjoint_common.jl:771-779; the discrete and the mixed cost of test/Core7/mixed_costs.jl:199-390, 391-570).

The reference's tests compare adjoint_sensitivities(...; dgdu_discrete, dgdp_discrete) with ForwardDiff through the solver; scipy DOP853 forward
sensitivities at rtol = atol = 1e-13 play ForwardDiff's role here (as in make_golden.py, whose model definitions are reused).

    python tests/golden/make_discrete_losses.py        (needs scipy)

Problem: Lotka-Volterra `fiip`, u0 = [1, 1], p = [1.5, 1, 3, 1], tspan (0, 10), loss times 1:9 (mixed_costs.jl:10-16, save_start = save_end = false).
Losses (l_i at t_i, i 0-based, d_i the data column, u_n the last state):
  u1sq_p1   l_i = u1^2 + p1                     mixed_costs.jl:199-227
  u1sq_p2   l_i = u1^2 + p2                     mixed_costs.jl:404-424 (the discrete part of the mixed cost)
  lsq_data  l_i = sum_j (u_j - d_ij)^2          sum(abs2, sol .- data)
  full      l_i = (i + 1) p1 u1 u_n + sin(t_i) u1 + p2^2 d_i1 u_n     every argument of the callback in use
"""
import json
import os

import numpy as np
from scipy.integrate import solve_ivp

from make_golden import lv

HERE = os.path.dirname(os.path.abspath(__file__))


def gradient(u0, p, tspan, ts, dl):
    """dl(u, p, t, i) -> (dl/du [n], dl/dp [np]); returns dL/du0, dL/dp of L = sum_i l_i."""
    n, npar = len(u0), len(p)
    nth = n + npar

    def rhs(t, z):
        u = z[:n]
        S = z[n:].reshape(n, nth)
        f, J, P = lv(u, p, t)
        dS = J @ S
        dS[:, n:] += P
        return np.concatenate([f, dS.ravel()])

    S0 = np.zeros((n, nth)); S0[:, :n] = np.eye(n)
    sol = solve_ivp(rhs, tspan, np.concatenate([np.asarray(u0, float), S0.ravel()]), method="DOP853", rtol=1e-13, atol=1e-13, t_eval=ts)
    assert sol.success
    g = np.zeros(nth); us = []
    for i in range(len(ts)):
        u = sol.y[:n, i]; S = sol.y[n:, i].reshape(n, nth)
        gu, gp = dl(u, p, ts[i], i)
        g += gu @ S
        g[n:] += gp
        us.append(u.tolist())
    return g[:n], g[n:], us


def main():
    u0, p = [1.0, 1.0], np.array([1.5, 1.0, 3.0, 1.0])
    ts = np.arange(1.0, 9.5, 1.0)
    rng = np.random.default_rng(20240927)
    data = rng.uniform(0.5, 2.0, (len(ts), 2))
    losses = {
        "u1sq_p1": lambda u, p_, t, i: (np.array([2 * u[0], 0.0]), np.array([1.0, 0, 0, 0])),
        "u1sq_p2": lambda u, p_, t, i: (np.array([2 * u[0], 0.0]), np.array([0, 1.0, 0, 0])),
        "lsq_data": lambda u, p_, t, i: (2 * (u - data[i]), np.zeros(4)),
        "full": lambda u, p_, t, i: (np.array([(i + 1) * p_[0] * u[1] + np.sin(t), (i + 1) * p_[0] * u[0] + p_[1] ** 2 * data[i, 0]]),
                                     np.array([(i + 1) * u[0] * u[1], 2 * p_[1] * data[i, 0] * u[1], 0, 0])),
    }
    out = dict(u0=u0, p=p.tolist(), tspan=[0.0, 10.0], ts=ts.tolist(), data=data.tolist(), losses={})
    for name, dl in losses.items():
        du0, dp, us = gradient(u0, p, (0.0, 10.0), ts, dl)
        out["losses"][name] = dict(du0=du0.tolist(), dp=dp.tolist())
        out["u"] = us
    with open(os.path.join(HERE, "discrete_losses.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote discrete_losses.json")


if __name__ == "__main__":
    main()
