#!/usr/bin/env python
"""Adaptive Tsit5 on the 10^4-trajectory Lorenz ensemble (default tolerances 1e-6 / 1e-3 and 1e-8 / 1e-8): forward and reverse device times per sensealg, dp vs the oracle on a sample."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scimlsensitivity_jl_amd as sa
import oracle as O
import bench
N = int(os.environ.get("TS5_N", "10000"))
u0, p = bench.inputs(N)
ts = bench.save_times()
TOLS = ((1e-6, 1e-3),) if os.environ.get("TS5_TOLS") == "default" else ((1e-6, 1e-3), (1e-8, 1e-8))
for tol in TOLS:
    for alg, oalg in (("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS")):
        eng = sa.Engine("lorenz", alg, N, 0.0, bench.T_FINAL, 0.0, save_times=ts, loss_kind=1, loss_shift=bench.LOSS_SHIFT, p_shared=True, stepper=1, abstol=tol[0], reltol=tol[1],
                        checkpointing=(alg == "backsolve"))
        eng.forward(u0, p, want_out=False); eng.forward(u0, p, want_out=False)
        du0, dp = eng.adjoint(None)
        s0 = eng.stats()
        for _ in range(5):
            eng.adjoint(None)
        s1 = eng.stats()
        row = dict(alg=alg, abstol=tol[0], reltol=tol[1], quad=os.environ.get("HIPADJ_QUAD", "1"), forward_ms=s1["forward_ms_last"], reverse_ms=(s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / 5,
                   sweep_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 5)
        if tol[1] == 1e-8:
            n_chk = 256
            ref = O.Problem("LORENZ", alg=oalg, stepper="TSIT5", t0=0.0, t1=bench.T_FINAL, dt=0.0, abstol=tol[0], reltol=tol[1], save_times=ts, loss="LSQ_SHIFT", loss_shift=bench.LOSS_SHIFT,
                            checkpointing=(alg == "backsolve"))
            rdu0, rdp, _, _ = ref.adjoint_ensemble(u0[:n_chk], np.tile(p, (n_chk, 1)))
            row["du0_rel_first256"] = float(np.max(np.abs(du0[:n_chk] - rdu0)) / np.max(np.abs(rdu0)))
        print(json.dumps(row), flush=True)
        eng.close()
