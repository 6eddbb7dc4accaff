"""Generate tests/golden/bruss_horizon.json — the 2-D Brusselator of docs/src/examples/pde/brusselator.md:73-112 on the 32 x 32 grid over its
DOCUMENTED horizon tspan = (0, 11.5), solved with an implicit method, independently of oracle/ and csrc/: scipy BDF (the reference docs use
FBDF, :115) with the 5-point-stencil sparsity pattern, rtol = atol = 1e-10.  The device integrates the same horizon with explicit RK4 at the
stability limit (460 000 steps of dt = 2.5e-5; the 14.7 GB of interpolant knots fit the 288 GB of an MI355X); this fixture checks that
forward solution at a sample of grid points and save times.

    python tests/golden/make_bruss_horizon.py        (scipy; a few minutes)
"""
import json
import os

import numpy as np
from scipy.integrate import solve_ivp
from scipy.sparse import lil_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
G = 32
A, B, ALPHA = 3.4, 1.0, 10.0


def u0_doc():
    xs = np.linspace(0.0, 1.0, G)
    U = np.zeros((G, G)); V = np.zeros((G, G))
    for i in range(G):
        for j in range(G):
            U[i, j] = 22.0 * (xs[j] * (1 - xs[j])) ** 1.5      # docs/src/examples/pde/brusselator.md:88-96
            V[i, j] = 27.0 * (xs[i] * (1 - xs[i])) ** 1.5
    return np.concatenate([U.ravel(order="F"), V.ravel(order="F")])


def rhs(t, z):
    dx = 1.0 / (G - 1); adx = ALPHA / (dx * dx)
    U = z[:G * G].reshape(G, G, order="F"); V = z[G * G:].reshape(G, G, order="F")      # [i, j], i fastest
    lap = lambda W: np.roll(W, 1, 0) + np.roll(W, -1, 0) + np.roll(W, 1, 1) + np.roll(W, -1, 1) - 4.0 * W      # periodic (mod G), :98-112
    x = (np.arange(G) * dx)[:, None]; y = (np.arange(G) * dx)[None, :]
    force = np.where(((x - 0.3) ** 2 + (y - 0.6) ** 2 <= 0.01) & (t >= 1.1), 5.0, 0.0)
    dU = adx * lap(U) + B + U * U * V - (A + 1.0) * U + force
    dV = adx * lap(V) + A * U - U * U * V
    return np.concatenate([dU.ravel(order="F"), dV.ravel(order="F")])


def sparsity():
    n = G * G
    S = lil_matrix((2 * n, 2 * n))
    idx = lambda i, j: (i % G) + (j % G) * G
    for j in range(G):
        for i in range(G):
            c = idx(i, j)
            for (a, b) in ((i, j), (i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)):
                S[c, idx(a, b)] = 1; S[n + c, n + idx(a, b)] = 1
            S[c, n + c] = 1; S[n + c, c] = 1
    return S.tocsr()


def main():
    ts = np.arange(0.0, 11.5001, 0.5)
    z0 = u0_doc()
    # the forcing switches on at t = 1.1: integrate the two smooth pieces separately
    s1 = solve_ivp(rhs, (0.0, 1.1), z0, method="BDF", jac_sparsity=sparsity(), rtol=1e-10, atol=1e-10, t_eval=ts[ts <= 1.1])
    s2 = solve_ivp(rhs, (1.1, 11.5), s1.y[:, -1] if abs(s1.t[-1] - 1.1) < 1e-12 else solve_ivp(rhs, (s1.t[-1], 1.1), s1.y[:, -1], method="BDF", jac_sparsity=sparsity(), rtol=1e-10, atol=1e-10).y[:, -1],
                   method="BDF", jac_sparsity=sparsity(), rtol=1e-10, atol=1e-10, t_eval=ts[ts > 1.1])
    assert s1.success and s2.success
    Y = np.concatenate([s1.y, s2.y], axis=1)          # [2048][len(ts)]
    sample = [0, 5, 9 + 19 * G, 10 + 19 * G, 17 + 3 * G, 31 + 31 * G, G * G + 0, G * G + 9 + 19 * G, G * G + 20 + 20 * G, 2 * G * G - 1]
    out = dict(G=G, p=[A, B, ALPHA], tspan=[0.0, 11.5], ts=ts.tolist(), sample_indices=sample, u=Y[sample, :].tolist(),
               norm_per_time=np.linalg.norm(Y, axis=0).tolist(), method="scipy BDF rtol=atol=1e-10, 5-point sparsity, pieces split at t = 1.1",
               nfev=int(s1.nfev + s2.nfev), steps=int(len(s1.t) + len(s2.t)))
    with open(os.path.join(HERE, "bruss_horizon.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("steps", out["steps"], "nfev", out["nfev"], "final norm", out["norm_per_time"][-1])


if __name__ == "__main__":
    main()
