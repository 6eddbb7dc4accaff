"""Adaptive Tsit5 on the device: forward / reverse time for the Lorenz ensemble and the reference's LV test setup.
usage: python scripts/bench_tsit5.py [N]"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import scimlsensitivity_jl_amd as sa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(5)
cases = [("lorenz", np.array([1.0, 0.0, 0.0]), np.array([10.0, 28.0, 8 / 3]), 10.0, np.linspace(0, 10, 101), 0.1),
         ("lv", np.array([1.0, 1.0]), np.array([1.5, 1.0, 3.0, 1.0]), 10.0, np.linspace(0, 10, 21), 0.05)]
for model, u0c, p, T, ts, sig in cases:
    u0 = u0c + sig * rng.standard_normal((N, len(u0c)))
    for tol in ((1e-6, 1e-3), (1e-8, 1e-8)):
        for alg in (sa.InterpolatingAdjoint(), sa.BacksolveAdjoint(), sa.GaussAdjoint()):
            try:
                sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model, u0[0], (0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg,
                               dgdu_discrete=sa.LsqShift(2.0), abstol=tol[0], reltol=tol[1], max_steps=0)
                eng = sol.engine
                first_fwd = eng.stats()["forward_ms_last"]      # the handle's FIRST forward solve: code load and, when the record start was too small, the regrow-and-repeat round
                eng.forward(u0, p, want_out=False)
                eng.forward(u0, p, want_out=False)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); du0, dp = eng.adjoint(None); best = min(best, time.perf_counter() - t0)
                st = eng.stats()
                print(json.dumps(dict(model=model, N=N, alg=alg.name, abstol=tol[0], reltol=tol[1], forward_ms=st["forward_ms_last"], forward_first_call_ms=first_fwd,
                                      adjoint_kernel_ms=st["adjoint_main_kernel_ms_last"], adjoint_ms=st["adjoint_ms_last"],
                                      host_call_ms=best * 1e3, traj_per_s=N / (st["adjoint_ms_last"] * 1e-3),
                                      workspace_GB=st["workspace_bytes"] / 1e9, dp=[float(x) for x in dp])), flush=True)
                eng.close()
            except Exception as e:
                print(json.dumps(dict(model=model, alg=alg.name, tol=tol, error=str(e))), flush=True)
