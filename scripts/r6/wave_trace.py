#!/usr/bin/env python
"""Where does a shard's one-launch reverse pass spend its time?  Per-wave time stamps (100 MHz real-time counter, 10 ns) of ONE launch of k_interp_fused, taken by a
development build of the library (-DHIPADJ_WAVE_TRACE: scripts/r6/libhipadj_trace.so; the shipped library has none of this):

    HIPADJ_LIBRARY=$PWD/scripts/r6/libhipadj_trace.so python scripts/r6/wave_trace.py [ntraj=1250] [segments=0] [steps=1000]

slots of a wave's record: 0 entry | 1 segment bounds loaded | 2 first knot of the ring arrived | 3 sweep done | per tree level l: 4+4l payload stores issued, 5+4l stores drained,
6+4l ticket returned, 7+4l children loaded and folded | 20 root: du0 written, mu reduced over lanes, partial issued | 21 drained | 22 ensemble ticket | 23 dp written.
Prints (JSON lines): the kernel's span, the distribution of every phase over the waves, and the timeline of the critical path (the wave that writes dp, traced back level by level
through the last arrivers)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import scimlsensitivity_jl_amd as sa
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    dt = 0.01
    L = sa.load_library()
    if not hasattr(L, "hipadj_debug_set_trace"):
        raise SystemExit("this library has no trace points: build with -DHIPADJ_WAVE_TRACE and point HIPADJ_LIBRARY at it")
    L.hipadj_debug_set_trace.argtypes = [C.c_void_p]
    rng = np.random.default_rng(20240601)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    T = S * dt
    eng = sa.Engine("lorenz", "interpolating", n, 0.0, T, dt, save_times=np.linspace(0.0, T, S // 10 + 1), loss_kind=1, loss_shift=2.0, time_segments=seg)
    dev = torch.device("cuda:0")
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((n, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, None)
        for _ in range(600):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(200):
            eng.adjoint_dev(None, du0, dp)
        e1.record(st)
        torch.cuda.synchronize()
        ms_untraced = e0.elapsed_time(e1) / 200
        stt = eng.stats()
        Cseg = stt["time_segments"]
        G = int(os.environ.get("HIPADJ_FUSED_GROUP", "0") or 0)      # grouped form: the records are [ceil(C / G) G][waves]; ranks beyond C stay empty
        if G > 1:
            Cseg = (Cseg + G - 1) // G * G
        waves = (n + 63) // 64
        recs = []
        for rep in range(5):      # five traced launches, each in the middle of a burst (the pass before and after are ordinary)
            buf = torch.zeros((Cseg, waves, 32), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            for _ in range(20):
                eng.adjoint_dev(None, du0, dp)
            L.hipadj_debug_set_trace(C.c_void_p(buf.data_ptr()))
            eng.adjoint_dev(None, du0, dp)
            L.hipadj_debug_set_trace(None)
            for _ in range(5):
                eng.adjoint_dev(None, du0, dp)
            torch.cuda.synchronize()
            recs.append(buf.cpu().numpy().astype(np.int64))
    eng.close()
    print(json.dumps(dict(ntraj=n, steps=S, segments=stt["time_segments"], group=G, wtop=os.environ.get("HIPADJ_WTOP"), radix=os.environ.get("HIPADJ_TREE_RADIX"), waves=waves * Cseg, ms_per_pass_untraced_loop=ms_untraced, launches=stt["launches_per_pass"])))
    names = {0: "entry", 1: "bounds", 2: "first_knot", 3: "sweep_done", 20: "root_issued", 21: "root_drained", 22: "ensemble_ticket", 23: "dp_written"}
    for l in range(4):
        names.update({4 + 4 * l: f"L{l}_stores_issued", 5 + 4 * l: f"L{l}_drained", 6 + 4 * l: f"L{l}_ticket", 7 + 4 * l: f"L{l}_folded"})
    for rep, r in enumerate(recs):
        r = r[: stt["time_segments"]]
        Cseg = stt["time_segments"]
        t0 = r[:, :, 0][r[:, :, 0] > 0].min()
        us = np.where(r > 0, (r - t0) * 0.01, np.nan)          # microseconds since the first wave's entry
        span = np.nanmax(us)
        row = dict(rep=rep, span_us=float(span), entry_skew_us=dict(med=float(np.nanmedian(us[:, :, 0])), max=float(np.nanmax(us[:, :, 0]))))

        def stat(a):
            a = a[np.isfinite(a)]
            return None if a.size == 0 else dict(n=int(a.size), med=round(float(np.median(a)), 2), p90=round(float(np.percentile(a, 90)), 2), max=round(float(a.max()), 2))
        row["phase_us"] = {
            "entry->bounds": stat(us[:, :, 1] - us[:, :, 0]), "bounds->first_knot": stat(us[:, :, 2] - us[:, :, 1]), "first_knot->sweep_done": stat(us[:, :, 3] - us[:, :, 2]),
            "sweep_done (absolute)": stat(us[:, :, 3]),
        }
        for l in range(4):
            if np.isfinite(us[:, :, 5 + 4 * l]).any():
                row["phase_us"][f"L{l}: issue->drained"] = stat(us[:, :, 5 + 4 * l] - us[:, :, 4 + 4 * l])
                row["phase_us"][f"L{l}: drained->ticket"] = stat(us[:, :, 6 + 4 * l] - us[:, :, 5 + 4 * l])
                row["phase_us"][f"L{l}: ticket->folded (last arrivers)"] = stat(us[:, :, 7 + 4 * l] - us[:, :, 6 + 4 * l])
        row["phase_us"]["root: issued->drained"] = stat(us[:, :, 21] - us[:, :, 20])
        row["phase_us"]["root: drained->ensemble ticket"] = stat(us[:, :, 22] - us[:, :, 21])
        row["phase_us"]["root: ticket->dp written"] = stat(us[:, :, 23] - us[:, :, 22])
        # the critical path: the wave that wrote dp
        w = np.argwhere(np.isfinite(us[:, :, 23]))
        if len(w):
            y, x = w[0]
            row["dp_writer"] = dict(rank=int(y), block=int(x), timeline_us={names[k]: round(float(us[y, x, k]), 2) for k in sorted(names) if np.isfinite(us[y, x, k])})
        # when did the LAST sweep finish, and how long after it did the kernel end?
        # sweep duration by rank (0 = the top, one-column segment) and by how many waves share the wave's SIMD
        sw = us[:, :, 3] - us[:, :, 2]
        row["sweep_us_by_rank"] = {str(k): round(float(np.nanmedian(sw[k])), 2) for k in sorted(set([0, 1, 2, Cseg // 2, Cseg - 2, Cseg - 1]))}
        row["sweep_done_us_by_rank_max"] = {str(k): round(float(np.nanmax(us[k, :, 3])), 2) for k in sorted(set([0, 1, 2, Cseg // 2, Cseg - 2, Cseg - 1]))}
        hw = r[:, :, 24]
        h32 = hw & 0xffffffff      # HW_ID: SIMD_ID bits 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13
        simd_key = (((hw >> 32) & 0xf) << 16) | (((h32 >> 13) & 7) << 12) | (((h32 >> 12) & 1) << 8) | (((h32 >> 8) & 0xf) << 4) | ((h32 >> 4) & 3)
        keys, counts = np.unique(simd_key, return_counts=True)
        cnt_of = dict(zip(keys.tolist(), counts.tolist()))
        share = np.vectorize(lambda k: cnt_of[k])(simd_key)
        row["simds_used"] = int(len(keys)); row["waves_per_simd_histogram"] = {str(c): int((counts == c).sum()) for c in sorted(set(counts.tolist()))}
        row["sweep_us_by_simd_sharing"] = {str(c): dict(n=int((share == c).sum()), med=round(float(np.nanmedian(sw[share == c])), 2), max=round(float(np.nanmax(sw[share == c])), 2),
                                                         done_max=round(float(np.nanmax(us[:, :, 3][share == c])), 2)) for c in sorted(set(share.ravel().tolist()))}
        xcc = (hw >> 32) & 0xf
        row["waves_per_xcc"] = {str(x): int((xcc == x).sum()) for x in sorted(set(xcc.ravel().tolist()))}
        slow = np.argsort(np.nan_to_num(us[:, :, 3], nan=-1.0).ravel())[-8:][::-1]
        row["slowest_sweeps"] = [dict(rank=int(i // waves), block=int(i % waves), sweep_us=round(float(sw.ravel()[i]), 2), done_us=round(float(us[:, :, 3].ravel()[i]), 2),
                                      shares_simd_with=int(share.ravel()[i]) - 1, xcc=int(xcc.ravel()[i])) for i in slow]
        row["last_sweep_done_us"] = float(np.nanmax(us[:, :, 3]))
        row["tail_after_last_sweep_us"] = float(span - np.nanmax(us[:, :, 3]))
        print(json.dumps(row))


if __name__ == "__main__":
    main()
