#!/usr/bin/env python
"""The published 2-50-2 neural ODE (docs/src/Benchmark.md:62) as a wide runtime model, N = 4096: reverse-pass time on RK4 (232 steps) and on adaptive Tsit5, parity vs the oracle on a sample."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scimlsensitivity_jl_amd as sa
import oracle as O
rng = np.random.default_rng(11)
d, H, T, N = 2, 50, 1.5, int(os.environ.get("NODE_N", "4096"))
ts = np.linspace(0.0, T, 30); Sn = 29 * 8; dtn = T / Sn
p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d))
delta = rng.standard_normal((N, len(ts), d))
fun = sa.WideDeviceFunction.dense_chain("node_bench", (d, H, d), input_power=3)
for alg, oalg in (("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("backsolve", "BACKSOLVE")):
    for stepper in (0, 1):
        eng = sa.Engine(fun.name, alg, N, 0.0, T, dtn if stepper == 0 else 0.0, save_times=ts, checkpointing=(alg == "backsolve"), stepper=stepper, abstol=1e-6, reltol=1e-3)
        eng.forward(u0, p, want_out=False)
        du0, dp = eng.adjoint(delta)
        s0 = eng.stats()
        for _ in range(5):
            eng.adjoint(delta)
        s1 = eng.stats()
        row = dict(alg=alg, stepper="tsit5" if stepper else "rk4", N=N, forward_ms=s1["forward_ms_last"], reverse_ms=(s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / 5,
                   sweep_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 5)
        if stepper == 0:
            n_chk = 64
            ref = O.Problem("MLP1", alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dtn, save_times=ts, checkpointing=(alg == "backsolve"), dims=(d, H, 0, 0))
            rdu0, rdp, _, _ = ref.adjoint_ensemble(u0[:n_chk], np.tile(p, (n_chk, 1)), delta[:n_chk])
            row["parity_du0_first64"] = float(np.max(np.abs(du0[:n_chk] - rdu0)) / np.max(np.abs(rdu0)))
        print(json.dumps(row))
        eng.close()
