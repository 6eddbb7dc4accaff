#!/usr/bin/env python
"""Where does the reverse pass of a SMALL shard spend its time?  Sustained ms per pass of the Lorenz InterpolatingAdjoint sweep at a shard size for
a sweep of time-segment counts, in the one-launch form and in the three-launch form with 256-thread workgroups (HIPADJ_WPB=4: the four waves of a workgroup land on the four
SIMDs of one CU by construction), each configuration in its own process.
    python scripts/r5/shard_study.py [ntraj ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(n, seg):
    import numpy as np
    import torch
    import scimlsensitivity_jl_amd as sa
    rng = np.random.default_rng(20240601)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    eng = sa.Engine("lorenz", "interpolating", n, 0.0, 10.0, 0.01, save_times=np.linspace(0.0, 10.0, 101), loss_kind=1, loss_shift=2.0, time_segments=seg)
    dev = torch.device("cuda:0")
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((n, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, None)
        for _ in range(min(2000, int(4e6 / n))):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(200):
                eng.adjoint_dev(None, du0, dp)
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200)
        # the sweep kernel alone (events on its dispatch packet)
        eng.set_timing(1)
        for _ in range(3):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize(); eng.synchronize()
        s0 = eng.stats()
        for _ in range(50):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize(); eng.synchronize()
        s1 = eng.stats()
    print(json.dumps(dict(n=n, segments_asked=seg, segments=s1["time_segments"], launches=s1["launches_per_pass"], ms_per_pass=best,
                          main_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 50,
                          fused=os.environ.get("HIPADJ_FUSED", "1"), wpb=os.environ.get("HIPADJ_WPB", "1"), radix=os.environ.get("HIPADJ_TREE_RADIX", "4"))))
    eng.close()


if __name__ == "__main__":
    if sys.argv[1] == "_child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for n in [int(x) for x in sys.argv[1:]] or [1250]:
            segs = (0, 13, 26, 39, 51, 64, 80, 100) if n <= 2500 else (0, 6, 13, 20, 26, 39, 52)
            for env in ({}, {"HIPADJ_TREE_RADIX": "8"}, {"HIPADJ_FUSED": "0"}, {"HIPADJ_FUSED": "0", "HIPADJ_WPB": "4"}):
                for seg in segs:
                    if env.get("HIPADJ_TREE_RADIX") and seg not in (0, 51, 64, 26):
                        continue
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "_child", str(n), str(seg)], env=dict(os.environ, **env), capture_output=True, text=True)
                    line = [l for l in r.stdout.split("\n") if l.startswith("{")]
                    print(line[-1] if line else json.dumps(dict(n=n, seg=seg, env=env, error=r.stderr[-300:])), flush=True)
