"""The 30 x 50 matrix state (HBM-bound wide model) with different workgroup sizes: reverse-pass time and fraction of the HBM peak.  usage: python scripts/r3/wide_threads_sweep.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scimlsensitivity_jl_amd as sa
R, Cc, S, dt = 30, 50, 100, 0.01
n = R * Cc; ts = np.linspace(0.0, S * dt, 11); rng = np.random.default_rng(11)
for N in (512, 2048):
    for T in (0, 64, 128, 192, 256, 512, 768):
      try:
        fun = sa.WideDeviceFunction.index_affine(f"sweep_idx_{T}", R, Cc, threads=T)
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, S * dt, dt, save_times=ts)
        u0 = rng.standard_normal((N, n)); p = rng.random(2); delta = rng.standard_normal((N, len(ts), n))
        eng.forward(u0, p, want_out=False); eng.adjoint(delta)
        s0 = eng.stats()
        for _ in range(5): eng.adjoint(delta)
        s1 = eng.stats()
        kms = (s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 5
        by = N * (S + 1) * 16.0 * n + N * len(ts) * 8.0 * n
        print(f"N {N:5d} threads {T:4d}  sweep kernel {kms:.4f} ms  {by / (kms * 1e-3) / 1e12:.2f} TB/s = {by / (kms * 1e-3) / 8e12:.3f}  forward {s1['forward_ms_last']:.3f} ms", flush=True)
        eng.close()
      except Exception as e:
        print(f"N {N} threads {T}: {str(e)[:300]}", flush=True)
