#!/usr/bin/env python
"""Secondary measurements for BASELINE configs[3] (MLP, GaussAdjoint, MFMA) and configs[4] (Brusselator 32x32,
QuadratureAdjoint): forward and reverse device times through the C ABI.  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(eng, u0, p, delta, reps):
    import torch
    eng.forward(u0, p, want_out=False)
    eng.adjoint(delta)
    s0 = eng.stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        du0, dp = eng.adjoint(delta)
    wall = (time.perf_counter() - t0) / reps
    s1 = eng.stats()
    return dict(forward_ms=s1["forward_ms_last"], adjoint_ms=(s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / reps,
                main_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / reps, host_wall_ms=wall * 1e3,
                workspace_GB=s1["workspace_bytes"] / 1e9), du0, dp


def main():
    import scimlsensitivity_jl_amd as sa
    from test_gpu_parity import mlp_params, bruss_u0
    rng = np.random.default_rng(0)
    # ---- configs[3]: MLP 2 -> 128 -> 128 -> 2, batch 4096, tspan (0, 1.5), RK4 dt = 0.01, 30 loss times, GaussAdjoint
    d, H, B = 2, 128, 4096
    S, dt = 150, 0.01
    ts = dt * np.arange(5, S + 1, 5)
    u0 = rng.standard_normal((1, d * B)); p = mlp_params(d, H)
    delta = rng.standard_normal((1, len(ts), d * B))
    for alg in ("gauss", "interpolating", "backsolve", "quadrature"):
        eng = sa.Engine("mlp", alg, 1, 0.0, S * dt, dt, save_times=ts, dims=(d, H, B, 0))
        r, du0, dp = run(eng, u0, p, delta, 3)
        nq = 2 if alg == "gauss" else 4
        # flops of the sweep: per step (3 fwd + 4 bwd [+1 fsal + 2x(fwd+bwd) for Gauss]) H x H x B contractions, 2 flop/MAC
        gemms = (3 + 4 + (1 + 4 if alg == "gauss" else 0)) * S          # nominal count of the Gauss / Interpolating stage algebra (Backsolve: 8, Quadrature: sweep 8 + 2 x 22 nodes per interval)
        sweep_flops = gemms * 2.0 * H * H * B
        wgrad_flops = nq * S * 2.0 * B * (H * (H + 16) + H * 16 + 16 * (H + 16))
        r.update(case=f"mlp H={H} B={B} S={S} {alg}", sweep_TFLOPs=sweep_flops / (r["main_kernel_ms"] * 1e-3) / 1e12,
                 wgrad_ms=r["adjoint_ms"] - r["main_kernel_ms"], wgrad_TFLOPs=wgrad_flops / max((r["adjoint_ms"] - r["main_kernel_ms"]) * 1e-3, 1e-9) / 1e12,
                 dp_norm=float(np.linalg.norm(dp)))
        print(json.dumps(r))
        eng.close()
    # ---- configs[4]: Brusselator 32 x 32, QuadratureAdjoint, explicit RK4 at dt = 2.5e-5, 400 steps, 5 loss times
    G, dtb, Sb = 32, 2.5e-5, 400
    tsb = dtb * np.arange(0, Sb + 1, 100)
    for N in (1, 256):
        u0b = bruss_u0(G, N); pb = np.array([3.4, 1.0, 10.0])
        db = rng.standard_normal((N, len(tsb), 2 * G * G))
        for alg in ("quadrature", "interpolating"):
            eng = sa.Engine("bruss", alg, N, 0.0, Sb * dtb, dtb, save_times=tsb, dims=(G, 0, 0, 0))
            r, du0, dp = run(eng, u0b, pb, db, 3)
            r.update(case=f"bruss G={G} N={N} S={Sb} {alg}", us_per_step=r["main_kernel_ms"] * 1e3 / Sb,
                     knot_GBps=N * (Sb + 1) * 16.0 * 2 * G * G / (r["main_kernel_ms"] * 1e-3) / 1e9, dp=dp.tolist())
            print(json.dumps(r))
            eng.close()


if __name__ == "__main__":
    main()
