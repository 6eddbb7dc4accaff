"""Adaptive Tsit5 on RUNTIME-registered lane models (hiprtc): LV (2 states, 4 parameters) and Robertson (3, 3), 10^4 trajectories, default tolerances.
HIPADJ_TS5_REGS_USER=0/1 selects the LDS / register stage rows (default: registers when the toolkit's hiprtc is bound).  usage: python scripts/r3/bench_tsit5_user.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
import user_models as UM

N = 10000
rng = np.random.default_rng(5)
for name, m, u0c, p, T, ts in (("lv", UM.LV, np.array([1.0, 1.0]), np.array([1.5, 1.0, 3.0, 1.0]), 10.0, np.linspace(0, 10, 21)),
                               ("rober", UM.ROBER, np.array([1.0, 0.0, 0.0]), np.array([0.04, 3e2, 1e1]), 2.0, np.linspace(0, 2, 11))):
    f = sa.DeviceFunction("bt5u_" + name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    u0 = np.abs(u0c + 0.05 * rng.standard_normal((N, len(u0c))))
    for alg in (sa.InterpolatingAdjoint(), sa.BacksolveAdjoint(), sa.GaussAdjoint()):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg, dgdu_discrete=sa.LsqShift(0.5), abstol=1e-6, reltol=1e-3, max_steps=0)
        eng = sol.engine
        eng.forward(u0, p, want_out=False)
        for _ in range(3):
            du0, dp = eng.adjoint(None)
        st = eng.stats()
        print(json.dumps(dict(model=name, alg=alg.name, regs=os.environ.get("HIPADJ_TS5_REGS_USER", "default"), forward_ms=round(st["forward_ms_last"], 4),
                              adjoint_kernel_ms=round(st["adjoint_main_kernel_ms_last"], 4), dp0=float(dp[0]))), flush=True)
        eng.close()
