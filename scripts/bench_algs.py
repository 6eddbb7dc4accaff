#!/usr/bin/env python
"""Secondary measurements (not the headline): reverse-pass time of each sensealg on the Lorenz-63 ensemble of
BASELINE configs[1]/[2] shapes, through the device-pointer API.  Prints one JSON line per algorithm."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import scimlsensitivity_jl_amd as sa
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    steps = 20
    rng = np.random.default_rng(20240601)
    u0n = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3))
    ts = np.linspace(0.0, 10.0, 101)
    dev = torch.device("cuda", 0)
    u0 = torch.tensor(u0n, device=dev); p = torch.tensor([10.0, 28.0, 8.0 / 3.0], device=dev, dtype=torch.float64)
    du0 = torch.empty((N, 3), device=dev, dtype=torch.float64); dp = torch.empty(3, device=dev, dtype=torch.float64)
    for alg, kw in (("interpolating", {}), ("interpolating", dict(time_segments=1)), ("interpolating", dict(checkpointing=True)),
                    ("backsolve", dict(checkpointing=True)), ("backsolve", dict(checkpointing=True, time_segments=1)),
                    ("gauss", {}), ("gauss", dict(time_segments=1)), ("gauss", dict(checkpointing=True)), ("quadrature", {})):
        eng = sa.Engine("lorenz", alg, N, 0.0, 10.0, 0.01, save_times=ts, loss_kind=1, loss_shift=2.0, **kw)
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
        print(json.dumps(dict(alg=alg, kw=kw, ntraj=N, time_segments=s1["time_segments"], wall_ms=wall * 1e3, main_kernel_ms=k,
                              traj_per_s=N / wall, forward_ms=s1["forward_ms_last"], alg_bytes=s1["adjoint_algorithmic_bytes"],
                              alg_GBps=s1["adjoint_algorithmic_bytes"] / (k * 1e-3) / 1e9 if k > 0 else None, dp=dp.cpu().tolist())))
        eng.close()


if __name__ == "__main__":
    main()
