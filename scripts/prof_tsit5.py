"""One adaptive Tsit5 forward + reverse pass of the Lorenz ensemble (for rocprofv3 counter passes)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scimlsensitivity_jl_amd as sa
N = 10000
rng = np.random.default_rng(5)
u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
ts = np.linspace(0, 10, 101)
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, 10.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(),
               dgdu_discrete=sa.LsqShift(2.0), abstol=1e-8, reltol=1e-8, max_steps=4096)
for _ in range(3):
    du0, dp = sol.engine.adjoint(None)
print(dp)
