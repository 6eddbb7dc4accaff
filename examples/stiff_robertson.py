#!/usr/bin/env python
"""examples/stiff_robertson.py — a stiff ensemble and a semi-explicit DAE through the host mirror of the reference's operator surface (`import scimlsensitivity_jl_amd as sa`).

  (1) Robertson kinetics at the classic stiff rates (0.04, 3e7, 1e4) +- 5 %, tspan (0, 100), G = y3(50) + y3(100), as a runtime model (text -> hiprtc): Rosenbrock23, GaussAdjoint —
      a problem adaptive Tsit5 needs ~1e6 steps per trajectory for.
  (2) The same chemistry as the reference writes it in test/Core3/adjoint.jl:1434-1454: the third row is the conservation constraint and the mass matrix diag(1, 1, 0) — a DAE,
      started from the test's inconsistent state [1, 0, 1]; InterpolatingAdjoint.  dG/dp of (2) equals dG/dp of (1) for trajectory 0 (the same trajectory, sum(u0) = 1).

    python examples/stiff_robertson.py [ntraj = 1024]          (needs an MI355X; without one the first solve fails loudly: there is no CPU fallback)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repository root (scimlsensitivity_jl_amd.py forwards to the package directory)
import scimlsensitivity_jl_amd as sa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
p = np.array([0.04, 3.0e7, 1.0e4]) * (1 + 0.05 * rng.uniform(-1, 1, (N, 3))); p[0] = [0.04, 3.0e7, 1.0e4]
ts = np.array([50.0, 100.0])
dg = np.zeros((N, len(ts), 3)); dg[:, :, 2] = 1.0                     # dg/du = e_3 at both loss times

ode = sa.DeviceFunction("rober_example", 3, 3,
                        "du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = p[1]*u[1]*u[1];")      # only f: VJPs by dual numbers
u0 = np.tile([1.0, 0.0, 0.0], (N, 1))
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(ode, u0[0], (0.0, 100.0), p[0]), u0, p), sa.Rosenbrock23(), saveat=ts, sensealg=sa.GaussAdjoint(), abstol=1e-10, reltol=1e-8)
du0, dp = sa.adjoint_sensitivities(sol, sa.Rosenbrock23(), t=ts, dgdu_discrete=dg)
st = sol.engine.stats()
print(f"ODE  : y(100) = {sol.u[0, 1]}, dG/dp = {dp[0]}   ({N} trajectories: forward {st['forward_ms_last']:.2f} ms, reverse {st['adjoint_ms_last']:.2f} ms)")
sol.engine.close()

dae = sa.DeviceFunction("rober_dae_example", 3, 3,
                        "du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = u[0] + u[1] + u[2] - 1.0;",
                        mass_matrix=np.diag([1.0, 1.0, 0.0]))
u0 = np.tile([1.0, 0.0, 1.0], (N, 1))                                  # inconsistent: the initialisation moves y3 to 0
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(dae, u0[0], (0.0, 100.0), p[0]), u0, p), sa.Rosenbrock23(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-10, reltol=1e-8)
du0d, dpd = sa.adjoint_sensitivities(sol, sa.Rosenbrock23(), t=ts, dgdu_discrete=dg)
print(f"DAE  : y(100) = {sol.u[0, 1]}, dG/dp = {dpd[0]}   (constraint residual {np.max(np.abs(sol.u.sum(axis=2) - 1.0)):.1e})")
sol.engine.close()
print("max relative difference of dG/dp between the two formulations:", float(np.max(np.abs(dpd - dp) / np.abs(dp))))
