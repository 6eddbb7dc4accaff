#!/usr/bin/env python
"""ContinuousCallback on the device, measured: an ensemble of bouncing balls (test/Callbacks2/continuous_callbacks.jl:10-14, 212-217 — du = [u2, -p1], condition u1, affect
u2 <- -p2 u2) dropped from 2 .. 9 with restitution 0.8 .. 0.9 over (0, 4): every trajectory its own event times and its own number of events (2 .. 7).  Per sensealg: forward and
reverse time of the library's own event timers, gradients per second (forward + reverse), the CPU oracle's time per trajectory on a sample (one core, same stepper and
tolerances) and the sample's agreement; and the same ensemble WITHOUT the callback (the ball falls through the floor) as the cost reference of the event machinery.

    python scripts/r6/bench_continuous_callback.py [N=65536] > profiles/r6_continuous_callback_bench.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa  # noqa: E402
import oracle as O  # noqa: E402   (the checker, timed beside the device as the CPU baseline — never on the product path)
import user_models as UM  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    rng = np.random.default_rng(5)
    T = 4.0
    u0 = np.stack([rng.uniform(2.0, 9.0, N), rng.uniform(-1.0, 1.0, N)], axis=1)
    p = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    ts = np.array([0.3, 1.0, 1.7, 2.2, 3.1, 4.0]); d = rng.standard_normal((N, len(ts), 2))
    m, cond, aff = UM.EVENTS[1]
    rows = []
    for tol in (1e-6, 1e-10):
        for cc in (True, False):
            f = sa.DeviceFunction(f"cc_bench_{int(cc)}_{tol:g}", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
            if cc:
                f.set_continuous_callback(cond, aff)
            for alg in (sa.InterpolatingAdjoint(), sa.GaussAdjoint()):
                sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), p[0]), u0, p), sa.Tsit5(), saveat=ts, sensealg=alg, abstol=tol, reltol=tol)
                eng = sol.engine
                best = 1e9
                for _ in range(3):
                    eng.forward(u0, p, want_out=False)
                    t0 = time.perf_counter(); du0, dp = eng.adjoint(d); best = min(best, time.perf_counter() - t0)
                st = eng.stats()
                row = dict(case="bouncing_ball", N=N, callback=cc, alg=alg.name, stepper="Tsit5", abstol=tol, reltol=tol, forward_ms=st["forward_ms_last"], adjoint_ms=st["adjoint_ms_last"],
                           adjoint_kernel_ms=st["adjoint_main_kernel_ms_last"], host_call_ms=best * 1e3, gradients_per_s=N / ((st["forward_ms_last"] + st["adjoint_ms_last"]) * 1e-3),
                           workspace_GB=st["workspace_bytes"] / 1e9)
                if cc:
                    ne = eng.event_counts()
                    row.update(events_min=int(ne.min()), events_max=int(ne.max()), events_mean=float(ne.mean()))
                    ref = O.Problem("FALLMASS", alg=alg.name.upper(), stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=tol, reltol=tol, save_times=ts, event_kind=1)
                    k = 64; t0 = time.perf_counter()
                    r = [ref.adjoint(u0[i], p[i], d[i]) for i in range(k)]
                    cpu = (time.perf_counter() - t0) / k
                    rdu0 = np.array([x[0] for x in r]); rdp = np.array([x[1] for x in r])
                    sc = lambda a: np.maximum(np.abs(a), 1e-3 * np.abs(a).max(axis=1, keepdims=True))
                    row.update(cpu_oracle_s_per_trajectory=cpu, cpu_oracle_gradients_per_s_one_core=1.0 / cpu,
                               sample_max_rel_du0=float(np.max(np.abs(du0[:k] - rdu0) / sc(rdu0))), sample_max_rel_dp=float(np.max(np.abs(dp[:k] - rdp) / sc(rdp))))
                eng.close()
                rows.append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(dict(what="scripts/r6/bench_continuous_callback.py", rows=rows), indent=1))


if __name__ == "__main__":
    main()
