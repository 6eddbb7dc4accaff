"""Reproducer for the heavy-kernel self-test: 8-state ring, dual-number VJPs, GaussAdjoint + RK4 through the per-column segment kernel
(1232 spilled registers, 2860 B of scratch per lane).  With HIPADJ_RTC_SELFTEST=0 the -O3 build of the ROCm 7.2 hiprtc returns non-finite gradients
(status -4); HIPADJ_RTC_FLAGS=-O1 / -O0, DBG_N=7 or HIPADJ_SEG_CAP=64 (the one-column kernel) are fine: profiles/r2_ring8_auto_gauss_o3_vs_o1.log.
    HIPADJ_RTC_SELFTEST=0 HIPADJ_USER_COLS=0 python scripts/repro_ring8_auto_gauss.py"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, scimlsensitivity_jl_amd as sa, oracle as O, user_models as UM
from scimlsensitivity_jl_amd import _lib
n = int(os.environ.get("DBG_N", "8")); m = UM.ring(n); npar = n + 1
f = sa.DeviceFunction(f"ring{n}_auto_dbg", n, npar, m["f"])
S, dt = 1000, 0.01; ts = dt * np.arange(10, S + 1, 10)
for N, segs in ((640, 13), (640, 1)):
    rng = np.random.default_rng(0)
    u0 = rng.uniform(0.3, 1.0, (N, n)); pp = rng.uniform(0.4, 1.2, (N, npar)); delta = rng.standard_normal((N, len(ts), n))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, S * dt), pp[0], (n, 0, 0, 0)), u0, pp), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.GaussAdjoint(), time_segments=segs)
    eng = sol.engine
    du0 = np.empty((N, n)); dp = np.empty((N, npar))
    try:
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
        bad = np.argwhere(~np.isfinite(dp).all(axis=1)).ravel()
        print("N", N, "segs", eng.stats()["time_segments"], "ok; non-finite rows:", len(bad), flush=True)
    except Exception as e:
        print("N", N, "segs", eng.stats()["time_segments"], "ERROR", str(e)[:120], flush=True)
        # look at the raw device output without the status check
        L = _lib.load()
    eng.close()
