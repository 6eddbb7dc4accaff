#!/usr/bin/env python
"""Reverse-pass time of the BASELINE configs[1] ensemble at the shard sizes of the 8 / 4 / 2 / 1-GPU layouts, for the library HIPADJ_LIBRARY points at (A/B builds):
sustained loop (1500 untimed passes, best of five 200-pass bursts) — one JSON line per size.   python scripts/r6/shard_time.py [label] [sizes...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import scimlsensitivity_jl_amd as sa

label = sys.argv[1] if len(sys.argv) > 1 else "default"
sizes = [int(x) for x in sys.argv[2:]] or [1250, 2500, 5000, 10000]
seg = int(os.environ.get("SHARD_SEGMENTS", "0"))
dev = torch.device("cuda:0")
for n in sizes:
    rng = np.random.default_rng(20240601)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    eng = sa.Engine("lorenz", "interpolating", n, 0.0, 10.0, 0.01, save_times=np.linspace(0.0, 10.0, 101), loss_kind=1, loss_shift=2.0, time_segments=seg)
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((n, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, None)
        for _ in range(1500):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(200):
                eng.adjoint_dev(None, du0, dp)
            e1.record(st)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 200)
    s1 = eng.stats()
    print(json.dumps(dict(label=label, n=n, segments=s1["time_segments"], launches=s1["launches_per_pass"], ms_per_pass_best=min(times), ms_per_pass_median=float(np.median(times)),
                          dp=[float(x) for x in dp.cpu().numpy()], du0_sum=float(du0.sum().item()))), flush=True)
    eng.close()
