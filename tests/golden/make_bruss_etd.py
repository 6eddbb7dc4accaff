"""Golden numbers for the exponential stepper of the PDE family (ORC_STEPPER_ETDRK4 / HIPADJ_STEPPER_ETDRK4_FIXED), independent of the oracle and of the device:
the Brusselator of docs/src/examples/pde/brusselator.md:85-112 on an 8 x 8 grid over (0, 2.2) — across the switch of the forcing at t = 1.1 — integrated by scipy's
Radau (rtol 1e-12) in two legs (forcing off / on), the loss  L = 1/2 sum_i |u(t_i) - 1|^2  over t = 0.55, 1.1, 1.65, 2.2, and its gradient by central differences
of that loss (all three parameters, four components of u0).  Run:  python tests/golden/make_bruss_etd.py  ->  tests/golden/bruss_etd.json"""
import json, os
import numpy as np
from scipy.integrate import solve_ivp

G = 8; p0 = np.array([3.4, 1.0, 10.0]); T = 2.2
xs = np.linspace(0, 1, G); X, Y = np.meshgrid(xs, xs, indexing="ij")
mask = ((X - 0.3) ** 2 + (Y - 0.6) ** 2 <= 0.01)
U0 = 22.0 * (Y * (1 - Y)) ** 1.5; V0 = 27.0 * (X * (1 - X)) ** 1.5
z0 = np.concatenate([U0.ravel(order="F"), V0.ravel(order="F")])
ts = [0.55, 1.1, 1.65, 2.2]


def lap(W):
    return np.roll(W, 1, 0) + np.roll(W, -1, 0) + np.roll(W, 1, 1) + np.roll(W, -1, 1) - 4 * W


def solution(z0, p):
    A, B, alpha = p; adx = alpha * (G - 1) ** 2

    def rhs(t, z, on):
        U = z[:G * G].reshape(G, G, order="F"); V = z[G * G:].reshape(G, G, order="F")
        f = np.where(mask & on, 5.0, 0.0)
        return np.concatenate([(adx * lap(U) + B + U * U * V - (A + 1) * U + f).ravel(order="F"), (adx * lap(V) + A * U - U * U * V).ravel(order="F")])
    s1 = solve_ivp(lambda t, z: rhs(t, z, False), (0, 1.1), z0, method="Radau", rtol=1e-12, atol=1e-14, t_eval=ts[:2])
    s2 = solve_ivp(lambda t, z: rhs(t, z, True), (1.1, T), s1.y[:, -1], method="Radau", rtol=1e-12, atol=1e-14, t_eval=ts[2:])
    return np.concatenate([s1.y, s2.y], axis=1).T


def loss(z0, p):
    return 0.5 * np.sum((solution(z0, p) - 1.0) ** 2)


gp = []
for j in range(3):
    e = np.zeros(3); e[j] = 1e-5 * p0[j]
    gp.append((loss(z0, p0 + e) - loss(z0, p0 - e)) / (2 * e[j]))
idx = [3, 40, 70, 100]
gu = []
for c in idx:
    e = np.zeros_like(z0); e[c] = 1e-5
    gu.append((loss(z0 + e, p0) - loss(z0 - e, p0)) / 2e-5)
out = dict(G=G, p=p0.tolist(), t1=T, ts=ts, u0=z0.tolist(), sol=solution(z0, p0).tolist(), dp=gp, du0_index=idx, du0=gu,
           note="scipy Radau rtol 1e-12, two legs around the forcing switch at t = 1.1; gradient of 1/2 sum |u(t_i) - 1|^2 by central differences (relative accuracy ~1e-7)")
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bruss_etd.json"), "w"))
print(gp, gu)
