"""QuadratureAdjoint on the adaptive path: forward / reverse ms (Lorenz, N = 10^4)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scimlsensitivity_jl_amd as sa
N = 10000
rng = np.random.default_rng(5)
u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
ts = np.linspace(0, 10, 101)
for tol in ((1e-6, 1e-3), (1e-8, 1e-8)):
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, 10.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.QuadratureAdjoint(),
                   dgdu_discrete=sa.LsqShift(2.0), abstol=tol[0], reltol=tol[1])
    for _ in range(3):
        du0, dp = sol.engine.adjoint(None)
    st = sol.engine.stats()
    print(json.dumps(dict(case="lorenz tsit5 quadrature", abstol=tol[0], reltol=tol[1], forward_ms=st["forward_ms_last"], adjoint_ms=st["adjoint_ms_last"],
                          lambda_pass_ms=st["adjoint_main_kernel_ms_last"], workspace_GB=st["workspace_bytes"] / 1e9)))
    sol.engine.close()
