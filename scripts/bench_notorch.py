#!/usr/bin/env python
"""Torch-free timing of the headline workload and of the off-grid sweep through the host-pointer C ABI (HIPADJ_NO_TORCH=1:
no torch import, the process runs on the HIP runtime of the ROCm installation alone — what a Julia host sees).  Prints one JSON
line per case: the library's own HIP events around the dominant kernel (hipadj_stats) and the wall time of the synchronous
host call (PCIe-inclusive: u0 in, du0 / dp out).  scripts/archive/gpu_quick2.sh runs it plain and under rocprofv3 --kernel-trace --stats."""
import json
import os
import sys
import time

os.environ.setdefault("HIPADJ_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scimlsensitivity_jl_amd as sa  # noqa: E402

N, T, DT, REPS = 10000, 10.0, 0.01, 20
rng = np.random.default_rng(20240601)
u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3))
p = np.array([10.0, 28.0, 8.0 / 3.0])
prob = sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0.0, T), p), u0)

CASES = [
    ("on-grid saveat=0.1 (BASELINE configs[1]), time-segmented k_interp", dict(saveat=0.1)),
    ("off-grid saveat=0.1003 (100 stops off the step grid), time-segmented k_offgrid_seg", dict(saveat=0.1003)),
    ("off-grid saveat=0.1003, sequential in time (time_segments = 1), k_interp_offgrid", dict(saveat=0.1003, time_segments=1)),
    ("off-grid saveat=0.1003, GaussAdjoint, time-segmented", dict(saveat=0.1003, sensealg=sa.GaussAdjoint())),
]
for label, kw in CASES:
    kw.setdefault("sensealg", sa.InterpolatingAdjoint())
    sol = sa.solve(prob, sa.RK4(), dt=DT, dgdu_discrete=sa.LsqShift(2.0), want_out=False, **kw)
    eng = sol.engine
    eng.adjoint(None)                                    # warm-up (code load)
    st0 = eng.stats()
    t0 = time.perf_counter()
    for _ in range(REPS):
        du0, dp = eng.adjoint(None)
    wall = (time.perf_counter() - t0) / REPS
    st1 = eng.stats()
    k_ms = (st1["adjoint_main_kernel_ms_total"] - st0["adjoint_main_kernel_ms_total"]) / REPS
    a_ms = (st1["adjoint_ms_total"] - st0["adjoint_ms_total"]) / REPS
    print(json.dumps({"case": label, "ntraj": N, "loss_times": int(len(sol.t)), "time_segments": st1["time_segments"],
                      "main_kernel_ms": k_ms, "reverse_pass_device_ms": a_ms, "host_call_ms": wall * 1e3,
                      "algorithmic_GBps": st1["adjoint_algorithmic_bytes"] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                      "dp": dp.tolist(), "torch_in_process": "torch" in sys.modules}))
    eng.close()
