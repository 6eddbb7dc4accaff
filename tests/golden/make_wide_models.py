"""Generate tests/golden/wide_models.json — independent high-accuracy gradients for the three models of the workgroup-per-trajectory family.

Same role as make_golden.py (the reference is pure Julia and cannot run here): the forward-sensitivity system dS/dt = J S + df/dtheta is integrated with
scipy DOP853 at rtol = atol = 1e-13 and contracted with dl/du at the loss times.  The right-hand sides are restated HERE in numpy matrix form,
from the reference's own definitions, independently of oracle/ and csrc/ (which use explicit loops over a flat parameter vector):

  node    Lux.Chain(x -> x.^3, Dense(2, 50, tanh), Dense(50, 2)), u0 = [2, 0], tspan (0, 1.5), 30 loss times, loss = sum(abs2, data - pred)
          (docs/src/Benchmark.md:62-80); parameters in ComponentArray's flat order: layer_2.weight (50 x 2, column-major), layer_2.bias,
          layer_3.weight (2 x 50, column-major), layer_3.bias
  linear  u' = A u with every entry of A a parameter (column-major), 6 states, loss = sum_i w_i . u(t_i)
  matrix  the reference's 30 x 50 matrix state, df[i, j] = p1 i + p2 j, l = sum(abs2, sol) at saveat 0:0.1:1 (test/Core5/size_handling_adjoint.jl:37-70) —
          here as a 5 x 4 state so that the sensitivity system stays small; its closed form is checked too

    python tests/golden/make_wide_models.py        (needs scipy; a few seconds)
"""
import json
import os

import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))


def gradient(f_J_P, u0, p, tspan, ts, dldu):
    n, npar = len(u0), len(p)

    def rhs(t, z):
        u = z[:n]; S = z[n:].reshape(n, n + npar)
        f, J, P = f_J_P(u, p, t)
        dS = J @ S; dS[:, n:] += P
        return np.concatenate([f, dS.ravel()])
    z0 = np.concatenate([u0, np.hstack([np.eye(n), np.zeros((n, npar))]).ravel()])
    sol = solve_ivp(rhs, tspan, z0, method="DOP853", rtol=1e-13, atol=1e-13, t_eval=ts)
    g = np.zeros(n + npar); out = []
    for i in range(len(ts)):
        u = sol.y[:n, i]; S = sol.y[n:, i].reshape(n, n + npar)
        out.append(u.tolist()); g += dldu(u, i) @ S
    return g[:n], g[n:], out


def node(d, H):
    def f_J_P(u, p, t):
        W1 = p[:H * d].reshape(H, d, order="F"); b1 = p[H * d:H * d + H]
        W2 = p[H * d + H:H * d + H + d * H].reshape(d, H, order="F"); b2 = p[H * d + H + d * H:]
        x = u ** 3; z = W1 @ x + b1; h = np.tanh(z); s = 1.0 - h * h
        f = W2 @ h + b2
        J = W2 @ (s[:, None] * W1) @ np.diag(3.0 * u * u)
        # df/dW1[i, j] = W2[:, i] s_i x_j (column-major flat index i + j H); df/db1 = W2 diag(s); df/dW2[i, j] = e_i h_j (index i + j d); df/db2 = I
        P = np.zeros((d, len(p)))
        for j in range(d):
            P[:, j * H:(j + 1) * H] = W2 * (s * x[j])[None, :]
        P[:, H * d:H * d + H] = W2 * s[None, :]
        for j in range(H):
            P[:, H * d + H + j * d:H * d + H + (j + 1) * d] = np.eye(d) * h[j]
        P[:, H * d + H + d * H:] = np.eye(d)
        return f, J, P
    return f_J_P


def linear(n):
    def f_J_P(u, p, t):
        A = p.reshape(n, n, order="F")
        P = np.zeros((n, n * n))
        for j in range(n):
            P[:, j * n:(j + 1) * n] = np.eye(n) * u[j]
        return A @ u, A, P
    return f_J_P


def matrix_state(R, Cc):
    ii = np.tile(np.arange(1, R + 1), Cc).astype(float); jj = np.repeat(np.arange(1, Cc + 1), R).astype(float)   # column-major (i, j) of component c

    def f_J_P(u, p, t):
        return p[0] * ii + p[1] * jj, np.zeros((R * Cc, R * Cc)), np.stack([ii, jj], axis=1)
    return f_J_P, ii, jj


def main():
    rng = np.random.default_rng(20260926)
    res = {}
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, 0.05 * rng.standard_normal(H), rng.standard_normal(d * H) * 0.07, 0.05 * rng.standard_normal(d)])
    u0 = np.array([2.0, 0.0]); data = rng.standard_normal((len(ts), d))
    du0, dp, out = gradient(node(d, H), u0, p, (0.0, T), ts, lambda u, i: 2.0 * (u - data[i]))
    res["node"] = dict(dims=[d, H], u0=u0.tolist(), p=p.tolist(), T=T, ts=ts.tolist(), data=data.tolist(), du0=du0.tolist(), dp=dp.tolist(), out=out)
    n, T = 6, 1.0
    ts = np.array([0.0, 0.13, 0.37, 0.5, 0.81, 1.0])
    A = rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n); p = A.flatten(order="F")
    u0 = rng.standard_normal(n); w = rng.standard_normal((len(ts), n))
    du0, dp, out = gradient(linear(n), u0, p, (0.0, T), ts, lambda u, i: w[i])
    res["linear"] = dict(n=n, u0=u0.tolist(), p=p.tolist(), T=T, ts=ts.tolist(), w=w.tolist(), du0=du0.tolist(), dp=dp.tolist(), out=out)
    R, Cc, T = 5, 4, 1.0
    ts = np.linspace(0.0, T, 11)
    fjp, ii, jj = matrix_state(R, Cc)
    u0 = rng.standard_normal(R * Cc); p = rng.random(2)
    du0, dp, out = gradient(fjp, u0, p, (0.0, T), ts, lambda u, i: 2.0 * u)
    ex = np.zeros(2); exu = np.zeros(R * Cc)
    for t in ts:                                        # closed form: u_c(t) = u0_c + t (p1 i + p2 j)
        u = u0 + t * (p[0] * ii + p[1] * jj); ex += [np.sum(2 * u * t * ii), np.sum(2 * u * t * jj)]; exu += 2 * u
    assert np.max(np.abs(dp - ex)) < 1e-9 * np.max(np.abs(ex)) and np.max(np.abs(du0 - exu)) < 1e-9 * np.max(np.abs(exu))
    res["matrix"] = dict(dims=[R, Cc], u0=u0.tolist(), p=p.tolist(), T=T, ts=ts.tolist(), du0=du0.tolist(), dp=dp.tolist(), out=out)
    with open(os.path.join(HERE, "wide_models.json"), "w") as f:
        json.dump(res, f)
    print("wrote wide_models.json:", {k: (len(v["du0"]), len(v["dp"])) for k, v in res.items()})


if __name__ == "__main__":
    main()
