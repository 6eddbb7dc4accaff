#!/usr/bin/env python
"""Brusselator 32 x 32 (BASELINE configs[4]) timings for A/B runs of library builds (HIPADJ_LIBRARY): N = 1 and N = 256, Quadrature / Interpolating / Gauss."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_families import run


def main():
    import scimlsensitivity_jl_amd as sa
    from test_gpu_parity import bruss_u0
    rng = np.random.default_rng(0)
    G, dtb, Sb = 32, 2.5e-5, 400
    tsb = dtb * np.arange(0, Sb + 1, 100)
    for N in (1, 256):
        u0b = bruss_u0(G, N); pb = np.array([3.4, 1.0, 10.0]); db = rng.standard_normal((N, len(tsb), 2 * G * G))
        for alg in ("quadrature", "interpolating", "gauss"):
            eng = sa.Engine("bruss", alg, N, 0.0, Sb * dtb, dtb, save_times=tsb, dims=(G, 0, 0, 0))
            r, du0, dp = run(eng, u0b, pb, db, 3)
            print(json.dumps(dict(lib=os.path.basename(os.environ.get("HIPADJ_LIBRARY", "default")), case=f"bruss N={N} {alg}", forward_ms=round(r["forward_ms"], 4), adjoint_ms=round(r["adjoint_ms"], 4),
                                  us_per_step=round(r["main_kernel_ms"] * 1e3 / Sb, 3), dp=[float(x) for x in np.ravel(dp)[:3]])))
            eng.close()


if __name__ == "__main__":
    main()
