#!/usr/bin/env python
"""Tuning study (not the headline): k_interp of the Lorenz workload compiled through the runtime-model path with different
prefetch depths (HIPADJ_USER_PF -> VGPRs -> waves per SIMD) and numbers of time segments.  One JSON line per setting."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LZ = dict(f="du[0] = p[0]*(u[1]-u[0]); du[1] = u[0]*(p[1]-u[2]) - u[1]; du[2] = u[0]*u[1] - p[2]*u[2];",
          vjp="out[0] = -p[0]*lam[0] + (p[1]-u[2])*lam[1] + u[1]*lam[2]; out[1] = p[0]*lam[0] - lam[1] + u[0]*lam[2]; out[2] = -u[0]*lam[1] - p[2]*lam[2];",
          vjp_p="out[0] = (u[1]-u[0])*lam[0]; out[1] = u[0]*lam[1]; out[2] = -u[2]*lam[2];")


def main():
    import torch
    import scimlsensitivity_jl_amd as sa
    N, steps = 10000, 30
    rng = np.random.default_rng(20240601)
    u0n = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3))
    ts = np.linspace(0.0, 10.0, 101)
    dev = torch.device("cuda", 0)
    u0 = torch.tensor(u0n, device=dev); p = torch.tensor([10.0, 28.0, 8.0 / 3.0], device=dev, dtype=torch.float64)
    du0 = torch.empty((N, 3), device=dev, dtype=torch.float64); dp = torch.empty(3, device=dev, dtype=torch.float64)
    f = sa.DeviceFunction("lorenz_rt", 3, 3, LZ["f"], LZ["vjp"], LZ["vjp_p"])
    settings = [("lorenz", 0, 13)] + [("lorenz_rt", pf, seg) for pf, seg in ((8, 13), (4, 13), (2, 13), (2, 19), (2, 16), (3, 13), (4, 16), (2, 22), (1, 19))]
    for model, pf, seg in settings:
        if pf:
            os.environ["HIPADJ_USER_PF"] = str(pf)
        eng = sa.Engine(model if model == "lorenz" else f.name, "interpolating", N, 0.0, 10.0, 0.01, save_times=ts, loss_kind=1, loss_shift=2.0, time_segments=seg)
        eng.use_torch_stream()
        eng.forward_dev(u0, p, None)
        for _ in range(3):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize(); eng.synchronize()
        s0 = eng.stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        eng.synchronize()
        s1 = eng.stats()
        k = (s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / steps
        print(json.dumps(dict(model=model, pf=pf, time_segments=s1["time_segments"], wall_ms=round(wall * 1e3, 4), main_kernel_ms=round(k, 4), dp0=dp.cpu().tolist()[0])), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
