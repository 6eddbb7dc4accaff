#!/usr/bin/env python
"""Lotka-Volterra (the reference's test problem) as a 10^4-trajectory ensemble on adaptive Tsit5, default tolerances: reverse-sweep times with four lanes per trajectory
(HIPADJ_QUAD=2) and with one (HIPADJ_QUAD=0)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scimlsensitivity_jl_amd as sa
N = int(os.environ.get("LV_N", "10000"))
rng = np.random.default_rng(1)
u0 = np.array([1.0, 1.0]) + 0.2 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0]); ts = np.linspace(0.0, 10.0, 21)
for model in ("lv", "lvt"):
    for alg in ("interpolating", "backsolve", "gauss"):
        eng = sa.Engine(model, alg, N, 0.0, 10.0, 0.0, save_times=ts, loss_kind=1, loss_shift=2.0, p_shared=True, stepper=1, abstol=1e-6, reltol=1e-3, checkpointing=(alg == "backsolve"))
        eng.forward(u0, p, want_out=False); eng.forward(u0, p, want_out=False)
        eng.adjoint(None); s0 = eng.stats()
        for _ in range(5):
            eng.adjoint(None)
        s1 = eng.stats()
        print(json.dumps(dict(model=model, alg=alg, quad=os.environ.get("HIPADJ_QUAD", "auto"), forward_ms=s1["forward_ms_last"], sweep_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 5)), flush=True)
        eng.close()
