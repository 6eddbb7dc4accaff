#!/usr/bin/env python
"""Shader clock and board power while the headline reverse pass runs back to back (is k_interp_fused clock / power bound?).

Three phases on one handle (Lorenz, 10^4 trajectories, InterpolatingAdjoint, RK4 — bench.py's workload):
  busy   : reverse passes back to back for ~4 s (chunks of 100 passes bracketed by HIP events)
  spaced : one pass, host synchronise, 0.5 ms pause — the chip cools between launches
  idle   : nothing for 1 s
A poller thread samples every readable clock / power source it finds under /sys/class/drm/card*/device (hwmon freq*_input, power*_average / _input,
pp_dpm_sclk) every ~2 ms; `amd-smi metric` / `rocm-smi` snapshots are taken once per phase as a cross-check.  Prints one JSON object."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import scimlsensitivity_jl_amd as sa
import bench


def sources():
    src = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in sorted(glob.glob(dev + "/hwmon/hwmon*/freq*_input")) + sorted(glob.glob(dev + "/hwmon/hwmon*/power*_average")) + sorted(glob.glob(dev + "/hwmon/hwmon*/power*_input")):
            lab = f.replace("_input", "_label").replace("_average", "_label")
            name = open(lab).read().strip() if os.path.exists(lab) else ""
            src[f] = name or os.path.basename(f)
        if os.path.exists(dev + "/pp_dpm_sclk"):
            src[dev + "/pp_dpm_sclk"] = "pp_dpm_sclk"
    return src


def read(path):
    try:
        t = open(path).read()
    except Exception:
        return None
    if path.endswith("pp_dpm_sclk"):
        for line in t.splitlines():
            if "*" in line:
                return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")) * 1e6
        return None
    try:
        return float(t.strip())
    except Exception:
        return None


class Poller(threading.Thread):
    def __init__(self, src):
        super().__init__(daemon=True)
        self.src, self.rows, self.stop, self.phase = src, [], False, "warm"

    def run(self):
        while not self.stop:
            self.rows.append((time.perf_counter(), self.phase, [read(p) for p in self.src]))
            time.sleep(0.002)


def smi_snapshot():
    out = {}
    for name, cmd in (("amd-smi", ["amd-smi", "metric", "-c", "-p", "--json"]), ("rocm-smi", ["rocm-smi", "--showclocks", "--showpower", "--json"])):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out[name] = r.stdout[-3000:] if r.returncode == 0 else f"rc {r.returncode}: {r.stderr[-300:]}"
        except Exception as e:
            out[name] = repr(e)
    return out


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
    for _ in range(10):
        eng.adjoint_dev(None, du0, dp)
    torch.cuda.synchronize()
    src = sources()
    pol = Poller(src)
    pol.start()
    res = {"sources": src, "phases": {}}
    snaps = {}

    # busy
    pol.phase = "busy"
    chunks, t_end = [], time.perf_counter() + 4.0
    snap_thread = threading.Thread(target=lambda: snaps.__setitem__("busy", smi_snapshot()), daemon=True)
    snap_thread.start()
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(100):
            eng.adjoint_dev(None, du0, dp)
        e1.record(st)
        e1.synchronize()
        chunks.append(e0.elapsed_time(e1) / 100.0)
    snap_thread.join(30)
    res["phases"]["busy"] = {"ms_per_pass_chunks_of_100": {"n": len(chunks), "first5": chunks[:5], "last5": chunks[-5:], "min": min(chunks), "median": float(np.median(chunks)), "max": max(chunks)}}
    # spaced
    pol.phase = "spaced"
    one, t_end = [], time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        eng.adjoint_dev(None, du0, dp)
        e1.record(st)
        e1.synchronize()
        one.append(e0.elapsed_time(e1))
        time.sleep(0.0005)
    res["phases"]["spaced"] = {"ms_per_pass_single": {"n": len(one), "min": min(one), "median": float(np.median(one)), "max": max(one)}}
    pol.phase = "idle"
    snaps["idle"] = smi_snapshot()
    time.sleep(1.0)
    pol.stop = True
    pol.join()
    keys = list(src)
    for ph in ("busy", "spaced", "idle"):
        rows = [r for r in pol.rows if r[1] == ph]
        d = res["phases"].setdefault(ph, {})
        d["samples"] = len(rows)
        for i, k in enumerate(keys):
            v = [r[2][i] for r in rows if r[2][i] is not None]
            if v:
                d[f"{src[k]} ({k})"] = {"mean": float(np.mean(v)), "min": float(np.min(v)), "max": float(np.max(v)), "p10": float(np.percentile(v, 10)), "p90": float(np.percentile(v, 90))}
    res["smi_snapshots"] = snaps
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
