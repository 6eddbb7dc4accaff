#!/usr/bin/env python
"""The host-pointer calls (what the Julia binding calls) with and without the pipelined staging of round 6: hipadj_forward (u0 up, out = sol(ts) [N][M][n] down) and hipadj_adjoint
(Delta [N][M][n] up, du0 / dp down) on BASELINE configs[1], pageable numpy arrays, wall clock, median of 15 calls.  One JSON line; the mode comes from the environment
(HIPADJ_HOST_PIPELINE=0: one block, CPU copy and DMA back to back; unset: chunks of ~3 MB, the CPU copy of one under the DMA of the next)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import scimlsensitivity_jl_amd as sa  # noqa: E402


def main():
    N, T, dt = 10000, 10.0, 0.01
    rng = np.random.default_rng(1)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
    ts = np.arange(0, T + 1e-9, 0.1); M = len(ts)
    eng = sa.Engine("lorenz", "interpolating", N, 0.0, T, dt, save_times=ts, loss_kind=0)
    delta = rng.standard_normal((N, M, 3))
    out = eng.forward(u0, p, want_out=True); g0 = eng.adjoint(delta)
    tf, ta = [], []
    for _ in range(15):
        t0 = time.perf_counter(); o = eng.forward(u0, p, want_out=True); tf.append(time.perf_counter() - t0)
        assert np.array_equal(o, out)
    for _ in range(15):
        t0 = time.perf_counter(); g = eng.adjoint(delta); ta.append(time.perf_counter() - t0)
        assert np.array_equal(g[0], g0[0]) and np.array_equal(g[1], g0[1])
    eng.close()
    import hashlib
    print(json.dumps(dict(mode=("one block" if os.environ.get("HIPADJ_HOST_PIPELINE") == "0" else "pipelined chunks"), threads=os.environ.get("HIPADJ_HOST_COPY_THREADS"),
                          forward_ms=float(np.median(tf)) * 1e3, adjoint_ms=float(np.median(ta)) * 1e3, forward_ms_min=min(tf) * 1e3, adjoint_ms_min=min(ta) * 1e3,
                          link_bound_ms=(N * M * 3 * 8 + N * 3 * 8) / 63e9 * 1e3, out_sha=hashlib.sha1(out.tobytes()).hexdigest()[:12], du0_sha=hashlib.sha1(g0[0].tobytes()).hexdigest()[:12])))


if __name__ == "__main__":
    main()
