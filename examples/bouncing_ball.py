#!/usr/bin/env python
"""examples/bouncing_ball.py — the reference's ContinuousCallback test problem (test/Callbacks2/continuous_callbacks.jl:10-24, 212-217) through the host mirror of its operator surface:

    fiip(du, u, p, t):  du1 = u2;  du2 = -p1            u0 = [5, 0], tspan (0, 2.5), p = [9.8, 0.8], saveat 0.5, abstol = reltol = 1e-12
    condition(u, t, integrator) = u[1]                  affect!(integrator) = (integrator.u[2] = -integrator.p[2] * integrator.u[2])
    g(sol) = sum(sol)

The model, the condition and the affect are text compiled with hiprtc; every trajectory locates its own bounces on the dense output, and the adjoint carries the sensitivity of the
bounce TIMES.  Trajectory 0 is the reference's; trajectory i > 0 is dropped from 5 + 0.01 i.  Printed next to the closed-form gradient
(tests/golden/continuous_callbacks.json).

    python examples/bouncing_ball.py [ntraj = 4096]          (needs an MI355X; without one the first solve fails loudly: there is no CPU fallback)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)      # the repository root (scimlsensitivity_jl_amd.py forwards to the package directory)
import scimlsensitivity_jl_amd as sa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ts = np.arange(0.0, 2.5 + 1e-12, 0.5)
u0 = np.tile([5.0, 0.0], (N, 1)); u0[:, 0] += 0.01 * np.arange(N)
p = np.array([9.8, 0.8])

ball = sa.DeviceFunction("bouncing_ball_example", 2, 2, "du[0] = u[1]; du[1] = -p[0];")                       # only f: VJPs by dual numbers
cb = sa.ContinuousCallback(condition="c = u[0];", affect="un[1] = -p[1] * u[1];")
for alg in (sa.InterpolatingAdjoint(), sa.GaussAdjoint()):
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(ball, u0[0], (0.0, 2.5), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg, abstol=1e-12, reltol=1e-12, callback=cb)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((N, len(ts), 2)))      # dg = 1: g = sum(sol); dp: the sum over the ensemble (shared p)
    ne = sol.engine.event_counts(); st = sol.engine.stats()
    print(f"{alg.name:14s} du0[0] = {du0[0]}  bounces of trajectory 0: {ne[0]}   ({N} trajectories, {int(ne.sum())} events: forward {st['forward_ms_last']:.3f} ms, reverse {st['adjoint_ms_last']:.3f} ms)")
    sol.engine.close()

# one trajectory alone, so that dp is the reference's dp
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(ball, u0[0], (0.0, 2.5), p), u0[:1]), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-12, reltol=1e-12, callback=cb)
du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), 2)))
sol.engine.close()
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "continuous_callbacks.json")))["ball"]
print(f"device       : du0 = {du0[0]}, dp = {dp}")
print(f"closed form  : du0 = {np.array(gold['du0'])}, dp = {np.array(gold['dp'])}   (bounce at t = {gold['event_times'][0]:.12f})")
