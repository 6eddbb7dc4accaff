"""One adaptive case for counter passes: Lorenz, 10^4 trajectories, Tsit5 at the default tolerances, one sensealg (argv[1]), a few reverse passes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scimlsensitivity_jl_amd as sa

alg = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint(), "quadrature": sa.QuadratureAdjoint()}[sys.argv[1] if len(sys.argv) > 1 else "interpolating"]
N = 10000
rng = np.random.default_rng(5)
u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3]); ts = np.linspace(0, 10, 101)
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, 10.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg, dgdu_discrete=sa.LsqShift(2.0), abstol=1e-6, reltol=1e-3, max_steps=0)
eng = sol.engine
eng.forward(u0, p, want_out=False)
for _ in range(3):
    du0, dp = eng.adjoint(None)
st = eng.stats()
print("forward_ms", st["forward_ms_last"], "adjoint_kernel_ms", st["adjoint_main_kernel_ms_last"], "dp", dp)
if "--steps" in sys.argv:   # step-count statistics of the forward solve per wave (64 consecutive trajectories)
    import ctypes as C
    ns = eng.forward_step_counts() if hasattr(eng, "forward_step_counts") else None
    if ns is not None:
        ns = np.asarray(ns)[:N]; w = ns[: N // 64 * 64].reshape(-1, 64)
        print("forward steps: mean", ns.mean(), "max", ns.max(), "| per wave: mean of max", w.max(axis=1).mean(), " mean of mean", w.mean(axis=1).mean())
eng.close()
