"""Adaptive Tsit5 with checkpointing = true (Interpolating / Gauss: per-lane interval re-solve), Lorenz and LV ensembles: forward / reverse times.
usage: python scripts/r3/bench_tsit5_ckpt.py [N]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scimlsensitivity_jl_amd as sa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(5)
cases = [("lorenz", np.array([1.0, 0.0, 0.0]), np.array([10.0, 28.0, 8 / 3]), 10.0, np.linspace(0, 10, 101), 0.1),
         ("lv", np.array([1.0, 1.0]), np.array([1.5, 1.0, 3.0, 1.0]), 10.0, np.linspace(0, 10, 21), 0.05)]
for model, u0c, p, T, ts, sig in cases:
    u0 = u0c + sig * rng.standard_normal((N, len(u0c)))
    for alg in (sa.InterpolatingAdjoint(checkpointing=True), sa.GaussAdjoint(checkpointing=True)):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model, u0[0], (0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg, dgdu_discrete=sa.LsqShift(2.0), abstol=1e-6, reltol=1e-3, max_steps=0)
        eng = sol.engine
        eng.forward(u0, p, want_out=False)
        for _ in range(3):
            du0, dp = eng.adjoint(None)
        st = eng.stats()
        print(json.dumps(dict(model=model, N=N, alg=alg.name, checkpointing=True, forward_ms=st["forward_ms_last"], adjoint_kernel_ms=st["adjoint_main_kernel_ms_last"], dp0=float(dp[0]))), flush=True)
        eng.close()
