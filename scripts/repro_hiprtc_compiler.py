"""Reproducer / A-B for the runtime-model compiler choice: a 5-state ring behind a dense mass matrix, GaussAdjoint + RK4, device vs
oracle.  Default: the library binds the build toolkit's hiprtc (ROCm 7.2 here) -> agreement to roundoff.  With HIPADJ_HIPRTC=libhiprtc.so the
hiprtc already in the process is used — inside python that is the ROCm 7.0 pair bundled with the torch wheel — and the same translation unit
returns parameter gradients wrong from the 6th digit (non-finite for some shapes).  profiles/r2_hiprtc_compiler_ab.log is the output of both.
    DBG_NTRAJ=54 python scripts/repro_hiprtc_compiler.py
    DBG_NTRAJ=54 HIPADJ_HIPRTC=libhiprtc.so python scripts/repro_hiprtc_compiler.py
Knobs: DBG_N (ring size), DBG_M (rand | diag | one | emu), DBG_ALGS (comma list), DBG_AUTO=1 (dual-number VJPs), HIPADJ_RTC_FLAGS."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, scimlsensitivity_jl_amd as sa, oracle as O, user_models as UM
np.set_printoptions(precision=6, linewidth=250)
def rel(a, b): return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
nring = int(os.environ.get("DBG_N", "5")); mkind = os.environ.get("DBG_M", "rand"); algs = os.environ.get("DBG_ALGS", "gauss").split(","); auto = os.environ.get("DBG_AUTO") == "1"
rng = np.random.default_rng(5)
m = UM.ring(nring); n, npar = m["n"], m["np"]
f = sa.DeviceFunction(f"ring{nring}_dbg3", n, npar, m["f"]) if auto else sa.DeviceFunction(f"ring{nring}_dbg3", n, npar, m["f"], m["vjp"], m["vjp_p"])
N, T = int(os.environ.get("DBG_NTRAJ", "3")), 2.0
u0 = rng.uniform(0.3, 1.0, (N, n)); ts = np.array([0.4, 1.1, 2.0]); delta = rng.standard_normal((N, 3, n))
M = np.eye(n) * 1.5 + 0.3 * rng.standard_normal((n, n))
if mkind == "diag": M = np.diag(1.0 + 0.2 * np.arange(n))
if mkind == "one": M = np.eye(n); M[1, 3] = 0.7
if mkind == "emu":
    import emu as E
    M = np.linalg.inv(E.ring_mm_inverse(n))
f.set_mass_matrix(M)
pp = rng.uniform(0.4, 1.2, (N, npar))
for alg in algs:
    sens = dict(gauss=sa.GaussAdjoint, gausskronrod=sa.GaussKronrodAdjoint, interpolating=sa.InterpolatingAdjoint, backsolve=sa.BacksolveAdjoint, quadrature=sa.QuadratureAdjoint)[alg]()
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), pp[0], (nring, 0, 0, 0)), u0, pp), sa.RK4(), dt=0.01, saveat=ts, sensealg=sens)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
    with O.mass_matrix(M):
        ref = O.Problem("RING", alg="GAUSS" if alg == "gausskronrod" else alg.upper(), t0=0.0, t1=T, save_times=ts, loss="COTANGENT", dims=(nring, 0, 0, 0), stepper="RK4", dt=0.01, quad_abstol=1e-12, quad_reltol=1e-12)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    print(os.environ.get("DBG_TAG", ""), "n", nring, "M", mkind, alg, "flags", os.environ.get("HIPADJ_RTC_FLAGS"), "du0 %.1e dp %.1e" % (rel(du0, rdu0), rel(dp, rdp)), flush=True)
    if rel(dp, rdp) > 1e-9:
        print("   dev", dp[0]); print("   orc", rdp[0])
    sol.engine.close()
