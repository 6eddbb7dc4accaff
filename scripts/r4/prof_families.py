#!/usr/bin/env python
"""The secondary kernels VERDICT r3 asks rocprofv3 evidence for, one short run each (forward once, then REPS reverse passes):
  k_wide_adjoint      30 x 50 index-affine state, N = 2048, InterpolatingAdjoint, 100 RK4 steps            (HBM)
  k_wide_adjoint_ts5  2-50-2 neural ODE of docs/src/Benchmark.md, N = 4096, adaptive Tsit5, Interpolating    (latency / VALU)
  k_wide_adjoint      the same network on fixed-step RK4 (232 steps), N = 4096                               (FP64 VALU)
  k_mlp_adjoint_grad<128>  configs[3]: MLP 2-128-128-2, 4096 columns, 150 RK4 steps, GaussAdjoint           (FP64 MFMA)
  k_bruss_quad_adj<32>     configs[4]: Brusselator 32 x 32, N = 256, QuadratureAdjoint, 400 RK4 steps        (HBM)
Prints one JSON line per case with the ALGORITHMIC bytes / flops of one launch (bench.py's formulas), so that scripts/r4/join_prof.py can put the
rocprofv3 duration and the PMC bytes next to them.  Run under `rocprofv3 --kernel-trace --stats` and under `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REPS = int(os.environ.get("PROF_REPS", "4"))
ONLY = os.environ.get("PROF_ONLY", "")


def run(eng, u0, p, delta):
    eng.set_timing(1)
    eng.forward(u0, p, want_out=False)
    eng.adjoint(delta)
    s0 = eng.stats()
    for _ in range(REPS):
        eng.adjoint(delta)
    s1 = eng.stats()
    return (s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / REPS, (s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / REPS


def main():
    import scimlsensitivity_jl_amd as sa
    from test_gpu_parity import mlp_params, bruss_u0
    rng = np.random.default_rng(11)

    def want(k):
        return not ONLY or k in ONLY.split(",")
    if want("idx"):
        R, Cc, S, dt, N = 30, 50, 100, 0.01, 2048
        n = R * Cc
        ts = np.linspace(0.0, S * dt, 11)
        fun = sa.WideDeviceFunction.index_affine("prof_idxaff", R, Cc)
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, S * dt, dt, save_times=ts)
        k, a = run(eng, rng.standard_normal((N, n)), rng.random(2), rng.standard_normal((N, len(ts), n)))
        print(json.dumps(dict(case="idx", kernel="k_wide_adjoint", match="k_wide_adjoint<", bound="hbm", alg_bytes=N * (S + 1) * 16.0 * n + N * len(ts) * 8.0 * n, event_kernel_ms=k, reverse_ms=a,
                              what=f"30 x 50 state, N = {N}, InterpolatingAdjoint, {S} RK4 steps")))
        eng.close()
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    flop_vjp = 10.0 * H * d + 2.0 * H + 2.0 * d
    if want("node_ts5"):
        N = 4096
        fun = sa.WideDeviceFunction.dense_chain("prof_node", (d, H, d), input_power=3)
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=1e-6, reltol=1e-3)
        k, a = run(eng, np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d)), p, rng.standard_normal((N, len(ts), d)))
        print(json.dumps(dict(case="node_ts5", kernel="k_wide_adjoint_ts5", match="k_wide_adjoint_ts5<", bound="latency", event_kernel_ms=k, reverse_ms=a,
                              what=f"2-50-2 neural ODE as published (adaptive Tsit5, 1e-6 / 1e-3), N = {N}, InterpolatingAdjoint")))
        eng.close()
    if want("node_rk4"):
        N, Sn = 4096, 29 * 8
        fun = sa.WideDeviceFunction.dense_chain("prof_node_rk4", (d, H, d), input_power=3)
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, T / Sn, save_times=ts)
        k, a = run(eng, np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d)), p, rng.standard_normal((N, len(ts), d)))
        print(json.dumps(dict(case="node_rk4", kernel="k_wide_adjoint", match="k_wide_adjoint<", bound="fp64_valu", alg_flops=N * Sn * 4.0 * flop_vjp, event_kernel_ms=k, reverse_ms=a,
                              what=f"2-50-2 neural ODE, {Sn} RK4 steps, N = {N}, InterpolatingAdjoint")))
        eng.close()
    if want("mlp"):
        dm, Hm, B, S = 2, 128, 4096, 150
        tsm = 0.01 * np.arange(5, S + 1, 5)
        eng = sa.Engine("mlp", "gauss", 1, 0.0, S * 0.01, 0.01, save_times=tsm, dims=(dm, Hm, B, 0))
        k, a = run(eng, rng.standard_normal((1, dm * B)), mlp_params(dm, Hm), rng.standard_normal((1, len(tsm), dm * B)))
        executed = ((4 + 6 + 2) * S + len(tsm)) * 2.0 * Hm * Hm * B
        print(json.dumps(dict(case="mlp", kernel="k_mlp_adjoint_grad", match="k_mlp_adjoint_grad<128", bound="mfma", alg_flops=executed, event_kernel_ms=k, reverse_ms=a,
                              what="configs[3]: MLP 2-128-128-2, 4096 columns, 150 RK4 steps, GaussAdjoint (executed contractions: 12 per step + 1 per loss jump, 2 H^2 B flop each)")))
        eng.close()
    if want("bruss"):
        G, dtb, Sb, N = 32, 2.5e-5, 400, 256
        tsb = dtb * np.arange(0, Sb + 1, 100)
        n = 2 * G * G
        eng = sa.Engine("bruss", "quadrature", N, 0.0, Sb * dtb, dtb, save_times=tsb, dims=(G, 0, 0, 0))
        k, a = run(eng, bruss_u0(G, N), np.array([3.4, 1.0, 10.0]), rng.standard_normal((N, len(tsb), n)))
        print(json.dumps(dict(case="bruss", kernel="k_bruss_quad_adj", match="k_bruss_quad_adj<32", bound="hbm", alg_bytes=N * (Sb + 1) * 16.0 * n + N * Sb * 32.0 * n, event_kernel_ms=k, reverse_ms=a,
                              what=f"configs[4]: Brusselator 32 x 32, N = {N}, QuadratureAdjoint lambda pass, {Sb} RK4 steps")))
        eng.close()


if __name__ == "__main__":
    main()
