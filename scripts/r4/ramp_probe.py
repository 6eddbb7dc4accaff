#!/usr/bin/env python
"""How long does the chip take to reach its sustained rate after an idle gap?  Per-pass duration (HIP events between consecutive reverse passes of bench.py's workload)
for 400 back-to-back passes after 100 ms of idle, twice, with the shader clock / power of THIS device (sysfs, found through the PCI bus id) sampled alongside."""
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import scimlsensitivity_jl_amd as sa
import bench


def my_card():
    try:
        bus = torch.cuda.get_device_properties(0).pci_bus_id.lower()
    except Exception:
        bus = None
    out = {}
    for dev in glob.glob("/sys/class/drm/card*/device"):
        real = os.path.realpath(dev).lower()
        if bus and bus in real:
            for f in glob.glob(dev + "/hwmon/hwmon*/freq1_input") + glob.glob(dev + "/hwmon/hwmon*/power1_input") + glob.glob(dev + "/hwmon/hwmon*/power1_average"):
                out[os.path.basename(f)] = f
    return bus, out


def main():
    N = 10000
    u0, p = bench.inputs(N)
    eng = sa.Engine("lorenz", "interpolating", N, 0.0, bench.T_FINAL, bench.DT, save_times=bench.save_times(), loss_kind=1, loss_shift=bench.LOSS_SHIFT, p_shared=True)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((N, 3), device=dev, dtype=torch.float64), torch.empty(3, device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    eng.set_timing(0)
    eng.forward_dev(tu0, tp, None)
    eng.adjoint_dev(None, du0, dp)
    torch.cuda.synchronize()
    bus, src = my_card()
    res = {"pci_bus_id": bus, "sysfs": src, "runs": []}
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            row = [time.perf_counter()]
            for k in sorted(src):
                try:
                    row.append(float(open(src[k]).read()))
                except Exception:
                    row.append(None)
            samples.append(row)
            time.sleep(0.0005)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    for rep in range(3):
        time.sleep(0.1)
        K = 400
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        t_start = time.perf_counter()
        evs[0].record(st)
        for i in range(K):
            eng.adjoint_dev(None, du0, dp)
            evs[i + 1].record(st)
        evs[K].synchronize()
        t_end = time.perf_counter()
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(K)]
        cum = np.cumsum(per)
        win = [(i, float(np.mean(per[i:i + 10]))) for i in range(0, K, 10)]
        s = [r for r in samples if t_start <= r[0] <= t_end]
        res["runs"].append({"per_pass_ms_first40": per[:40], "mean_of_10_windows": win, "ms_at_which_rate_within_3pct_of_last100": float(cum[next((i for i in range(K) if np.mean(per[i:i + 10]) < 1.03 * np.mean(per[-100:])), K - 1)]),
                            "last100_mean": float(np.mean(per[-100:])), "first25_mean": float(np.mean(per[:25])), "passes_5_to_25_mean": float(np.mean(per[5:25])),
                            "clock_samples": [[round((r[0] - t_start) * 1e3, 2)] + r[1:] for r in s][:200], "sample_keys": sorted(src)})
    stop[0] = True
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
