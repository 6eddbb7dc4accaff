"""Exact answer of the reference's mass-matrix test problem (test/Core3/adjoint.jl:1315-1337), from matrix exponentials (scipy only).

    M u' = A u + p + e2 sum(p),  A = [1 2 3; 4 5 6; 7 8 9],  M = -[1 2 4; 2 3 7; 1 3 41],  u0 = [1, 2, 3],  p = [1, 2, 3],  t in (0, 1)
    G = sum over ts = 0:0.01:1 of sum(u(t))                       (dg(out, u, p, t, i) = out .= 1)

The reference asserts  adjoint_sensitivities(...)[2]' ≈ ForwardDiff.gradient(G)  rtol 1e-11  for Quadrature / Gauss / GaussKronrod /
Interpolating / Backsolve.  The system is linear, so dG/dp and dG/du0 have closed forms:
    u(t) = E(t) u0 + Phi(t) M^{-1} B p,   E = exp(C t),  Phi = int_0^t exp(C s) ds,  C = M^{-1} A,  B = I + e2 1'
`lam0` is what the reference returns as du0: lam(t0) of M' lam' = -A' lam (src/sensitivity_interface.jl:500) = M^{-T} dG/du0.
Run:  python tests/golden/make_mass_matrix.py   -> tests/golden/mass_matrix.json"""
import json
import os
import numpy as np
import scipy.linalg as sl

A = np.array([[1.0, 2, 3], [4, 5, 6], [7, 8, 9]])
M = -np.array([[1.0, 2, 4], [2, 3, 7], [1, 3, 41]])
u0 = np.array([1.0, 2.0, 3.0]); p = np.array([1.0, 2.0, 3.0])
B = np.eye(3); B[1, :] += 1.0
C = np.linalg.solve(M, A)
ts = np.arange(101) * 0.01
dGdp = np.zeros(3); dGdu0 = np.zeros(3); G = 0.0; u_ts = []
for t in ts:
    X = sl.expm(np.block([[C, np.eye(3)], [np.zeros((3, 6))]]) * t)
    E, Phi = X[:3, :3], X[:3, 3:]
    u = E @ u0 + Phi @ np.linalg.solve(M, B @ p)
    u_ts.append(u.tolist()); G += u.sum()
    dGdp += np.ones(3) @ Phi @ np.linalg.solve(M, B)
    dGdu0 += np.ones(3) @ E
out = dict(A=A.tolist(), M=M.tolist(), u0=u0.tolist(), p=p.tolist(), ts=ts.tolist(), G=G, dGdp=dGdp.tolist(), dGdu0=dGdu0.tolist(),
           lam0=np.linalg.solve(M.T, dGdu0).tolist(), u_end=u_ts[-1])
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mass_matrix.json"), "w") as f:
    json.dump(out, f, indent=1)
print(out["dGdp"], out["lam0"])
