#!/usr/bin/env python
"""Golden gradients for the two parameter-dependent continuous-cost problems of /root/reference/test/Core7/adjoint_param.jl:6-95, computed
INDEPENDENTLY of the oracle: scipy DOP853 (rtol = atol = 1e-13) on the state + forward sensitivities S = du/dp + the running integrals
G = int g dt and dG/dp = int (g_u S + g_p) dt — the role ForwardDiff-of-quadgk plays in the reference test (:43-49, :77-83).
Writes tests/golden/adjoint_param.json.  Needs numpy + scipy only."""
import json
import os

import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))


def pendulum():
    p = np.array([1.0, -24.05, -19.137]); x0 = np.array([0.1, 0.0]); T = 10.0   # :12-14

    def rhs(t, z):
        x = z[:2]; S = z[2:8].reshape(2, 3)
        s, c = np.sin(x[0]), np.cos(x[0])
        f = np.array([p[0] * x[1], -s + (-p[0] * s + p[1] * x[1])])
        J = np.array([[0.0, p[0]], [-(1.0 + p[0]) * c, p[1]]])
        fp = np.array([[x[1], 0.0, 0.0], [-s, x[1], 0.0]])
        r = -p[0] * s + p[1] * x[1]
        g = (x[0] - np.pi) ** 2 + x[1] ** 2 + 5.0 * r * r                         # :18
        gu = np.array([2.0 * (x[0] - np.pi) - 10.0 * r * p[0] * c, 2.0 * x[1] + 10.0 * r * p[1]])
        gp = np.array([-10.0 * r * s, 10.0 * r * x[1], 0.0])
        return np.concatenate([f, (J @ S + fp).ravel(), [g], gu @ S + gp])
    z0 = np.zeros(2 + 6 + 1 + 3); z0[:2] = x0
    sol = solve_ivp(rhs, (0.0, T), z0, method="DOP853", rtol=1e-13, atol=1e-13)
    return dict(p=p.tolist(), u0=x0.tolist(), tspan=[0.0, T], G=float(sol.y[8, -1]), dGdp=sol.y[9:12, -1].tolist())


def lin1p():
    p = np.array([2.0, 3.0]); u0 = np.array([2.0]); T = 1.0                       # :52-53, 59

    def rhs(t, z):
        u = z[0]; S = z[1:3]
        f = -u * p[0] - p[1]
        fp = np.array([-u, -1.0])
        g = -u * p[0] - p[1]                                                      # :62
        return np.concatenate([[f], -p[0] * S + fp, [g], -p[0] * S + np.array([-u, -1.0])])
    z0 = np.zeros(1 + 2 + 1 + 2); z0[0] = u0[0]
    sol = solve_ivp(rhs, (0.0, T), z0, method="DOP853", rtol=1e-13, atol=1e-13)
    # closed form: u(t) = (u0 + b/a) e^{-a t} - b/a;  G = int (-a u - b) dt = int u' dt = u(T) - u0
    a, b = p
    uT = (u0[0] + b / a) * np.exp(-a * T) - b / a
    dGda = -T * (u0[0] + b / a) * np.exp(-a * T) + (-b / a ** 2) * np.exp(-a * T) + b / a ** 2
    dGdb = (np.exp(-a * T) - 1.0) / a
    return dict(p=p.tolist(), u0=u0.tolist(), tspan=[0.0, T], G=float(sol.y[3, -1]), dGdp=sol.y[4:6, -1].tolist(), G_closed=float(uT - u0[0]), dGdp_closed=[float(dGda), float(dGdb)])


if __name__ == "__main__":
    out = dict(pendulum=pendulum(), lin1p=lin1p(), source="tests/golden/make_adjoint_param.py: scipy DOP853 forward sensitivities, rtol = atol = 1e-13")
    json.dump(out, open(os.path.join(HERE, "adjoint_param.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
