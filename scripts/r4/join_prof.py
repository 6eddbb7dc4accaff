#!/usr/bin/env python
"""join_prof.py <cases.jsonl> <kernel_trace.csv> [<FETCH counter csv> <WRITE counter csv>]: the rocprofv3 duration (per-dispatch kernel trace) and PMC bytes per launch
of each case's kernel next to its algorithmic bytes / flops -> fraction of the roofline (HBM 8000 GB/s, FP64 MFMA / VALU 78.6 TFLOP/s).  Cases that share a
kernel name (runtime models instantiate the same name expression) are told apart by launch order: every case launches its kernel the same number of times."""
import collections
import csv
import json
import sys

cases = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]


def launches(path, value):
    """kernel name -> list of values in dispatch order"""
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    key = "Start_Timestamp" if rows and "Start_Timestamp" in rows[0] else ("Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None)
    if key:
        rows.sort(key=lambda r: int(r[key]))
    for r in rows:
        v = value(r)
        if v is not None:
            acc[r["Kernel_Name"]].append(v)
    return acc


def slice_for(acc, case):
    """this case's launches of its kernel: the cases sharing `match` split the launch list evenly, in case order"""
    names = [k for k in acc if case["match"] in k]
    if not names:
        return None, []
    sharing = [c for c in cases if c["match"] == case["match"]]
    allv = [v for k in names for v in acc[k]] if len(names) > 1 else acc[names[0]]
    per = len(allv) // len(sharing)
    i = sharing.index(case)
    return names[0], allv[i * per:(i + 1) * per]


dur = launches(sys.argv[2], lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
pmc = {}
for path, name in zip(sys.argv[3:5], ("FETCH_SIZE", "WRITE_SIZE")):
    pmc[name] = launches(path, lambda r, name=name: float(r["Counter_Value"]) if r.get("Counter_Name") == name else None)
for c in cases:
    kname, v = slice_for(dur, c)
    if not v:
        print(json.dumps(dict(case=c["case"], error="no launches found"))); continue
    v = v[1:] if len(v) > 1 else v      # drop the warm-up launch
    avg_ms = sum(v) / len(v)
    out = dict(case=c["case"], kernel=kname[:90], launches=len(v), rocprof_avg_ms=avg_ms, rocprof_min_ms=min(v), rocprof_max_ms=max(v), event_kernel_ms=c["event_kernel_ms"], what=c["what"])
    if "alg_bytes" in c:
        out.update(alg_GB=c["alg_bytes"] / 1e9, achieved_GBs=c["alg_bytes"] / (avg_ms * 1e-3) / 1e9, frac_of_8TBs=c["alg_bytes"] / (avg_ms * 1e-3) / 1e9 / 8000.0)
    if "alg_flops" in c:
        out.update(alg_TFLOP=c["alg_flops"] / 1e12, achieved_TFLOPs=c["alg_flops"] / (avg_ms * 1e-3) / 1e12, frac_of_78_6=c["alg_flops"] / (avg_ms * 1e-3) / 1e12 / 78.6)
    for name, acc in pmc.items():
        _, pv = slice_for(acc, c)
        if pv:
            out[name + "_KB_mean"] = sum(pv) / len(pv)
    if "FETCH_SIZE_KB_mean" in out and "WRITE_SIZE_KB_mean" in out:
        out["hbm_traffic_GB (FETCH x2 + WRITE)"] = (out["FETCH_SIZE_KB_mean"] * 2 + out["WRITE_SIZE_KB_mean"]) * 1024 / 1e9
        if "alg_bytes" in c:
            out["traffic_over_algorithmic"] = out["hbm_traffic_GB (FETCH x2 + WRITE)"] * 1e9 / c["alg_bytes"]
    print(json.dumps(out))
