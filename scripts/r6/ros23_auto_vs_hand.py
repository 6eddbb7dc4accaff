#!/usr/bin/env python
"""Rosenbrock23 on a runtime ring (n = 6) with hand-written VJP bodies against dual-number VJPs: the Jacobian of every W comes from n unit-vector VJP calls — does the dual-number
form pay n evaluations of the dual Jacobian, or does the compiler merge them?  One JSON line per variant."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
import user_models as UM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m = UM.ring(n)
rng = np.random.default_rng(3)
N = 8192
u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.4, 1.2, n + 1); ts = np.linspace(0, 2.0, 11)
for name, args in (("hand", (m["vjp"], m["vjp_p"])), ("dual", ())):
    f = sa.DeviceFunction(f"ring{n}_ros_{name}", m["n"], m["np"], m["f"], *args)
    for stepper, sname in ((sa.Rosenbrock23(), "Rosenbrock23"), (sa.Tsit5(), "Tsit5")):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 2.0), p), u0), stepper, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(1.5), abstol=1e-8, reltol=1e-8)
        eng = sol.engine
        eng.forward(u0, p, want_out=False); eng.adjoint(None); eng.adjoint(None)
        st = eng.stats()
        print(json.dumps(dict(n=n, vjp=name, stepper=sname, forward_ms=st["forward_ms_last"], adjoint_ms=st["adjoint_ms_last"])), flush=True)
        eng.close()
