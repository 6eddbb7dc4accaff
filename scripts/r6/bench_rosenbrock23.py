#!/usr/bin/env python
"""Rosenbrock23 on the device, measured: (a) an ensemble of Robertson problems at the stiff rates (0.04, 3e7, 1e4) +- 10 %, tspan (0, 100), G = y3(50) + y3(100) — a RUNTIME
model (hiprtc); (b) the Lotka-Volterra fit of test/Core2/stiff_adjoints.jl as an ensemble over initial states; (c) Lorenz, T = 10, next to Tsit5 at the same tolerance (a
non-stiff problem: what the W solves cost where they are not needed).  Per case: forward and reverse time of the library's own event timers, trajectories per second of one
gradient (forward + reverse), and the CPU oracle's time per trajectory on a sample (one core), same stepper and tolerances.

    python scripts/r6/bench_rosenbrock23.py [N=8192] > profiles/r6_rosenbrock23_bench.json
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


def run(model, u0, p, T, ts, loss, alg, stepper, tol, dLdu=None, reps=3):
    pr = sa.EnsembleProblem(sa.ODEProblem(model, u0[0], (0.0, T), p if np.ndim(p) == 1 else p[0]), u0, None if np.ndim(p) == 1 else p)
    sol = sa.solve(pr, stepper, saveat=ts, sensealg=alg, dgdu_discrete=loss, abstol=tol[0], reltol=tol[1])
    eng = sol.engine
    eng.forward(u0, p, want_out=False)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); du0, dp = eng.adjoint(dLdu); best = min(best, time.perf_counter() - t0)
    st = eng.stats()
    eng.close()
    return dict(forward_ms=st["forward_ms_last"], adjoint_ms=st["adjoint_ms_last"], adjoint_kernel_ms=st["adjoint_main_kernel_ms_last"], host_call_ms=best * 1e3,
                gradients_per_s=len(u0) / ((st["forward_ms_last"] + st["adjoint_ms_last"]) * 1e-3), workspace_GB=st["workspace_bytes"] / 1e9), du0, dp


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rng = np.random.default_rng(12)
    rows = []
    # (a) Robertson, stiff
    f = sa.DeviceFunction("rober_ros23_bench", 3, 3, UM.ROBER["f"], UM.ROBER["vjp"], UM.ROBER["vjp_p"])
    pp = np.array([0.04, 3.0e7, 1.0e4]) * (1 + 0.1 * rng.uniform(-1, 1, (N, 3)))
    u0 = np.tile([1.0, 0.0, 0.0], (N, 1))
    ts = np.array([50.0, 100.0]); d = np.zeros((N, 2, 3)); d[:, :, 2] = 1.0
    for tol in ((1e-6, 1e-4), (1e-8, 1e-6)):
        for alg in (sa.InterpolatingAdjoint(), sa.GaussAdjoint()):
            r, du0, dp = run(f, u0, pp, 100.0, ts, None, alg, sa.Rosenbrock23(), tol, dLdu=d)
            ref = O.Problem("ROBER", alg=alg.name.upper(), stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=tol[0], reltol=tol[1], save_times=ts, loss="COTANGENT")
            k = 16; t0 = time.perf_counter()
            rdp = np.array([ref.adjoint(u0[i], pp[i], d[i])[1] for i in range(k)])
            cpu = (time.perf_counter() - t0) / k
            rows.append(dict(case="robertson_stiff", N=N, alg=alg.name, stepper="Rosenbrock23", abstol=tol[0], reltol=tol[1], **r, cpu_oracle_s_per_trajectory=cpu,
                             cpu_oracle_gradients_per_s_one_core=1.0 / cpu, sample_max_rel_dp=float(np.max(np.abs(dp[:k] - rdp) / np.abs(rdp)))))
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    # (b) the reference's stiff-adjoint fit as an ensemble over initial states
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "stiff_adjoints.json")))["lv"]
    u0 = np.asarray(gold["u0"]) + 0.1 * rng.standard_normal((N, 2)); p = np.asarray(gold["p"]); ts = np.asarray(gold["ts"])
    data = np.tile(np.asarray(gold["target"])[None], (N, 1, 1))
    for stepper, sname in ((sa.Rosenbrock23(), "Rosenbrock23"), (sa.Tsit5(), "Tsit5")):
        r, du0, dp = run("lv", u0, p, 10.0, ts, sa.LsqData(data, 2.0), sa.InterpolatingAdjoint(), stepper, (1e-8, 1e-8))
        ref = O.Problem("LV", alg="INTERPOLATING", stepper="ROS23" if sname == "Rosenbrock23" else "TSIT5", t0=0.0, t1=10.0, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="LSQ_DATA", loss_scale=2.0)
        k = 16; t0 = time.perf_counter()
        rdu0 = np.array([ref.adjoint(u0[i], p, data[i])[0] for i in range(k)])
        cpu = (time.perf_counter() - t0) / k
        rows.append(dict(case="lv_fit_stiff_adjoints_jl", N=N, alg="interpolating", stepper=sname, abstol=1e-8, reltol=1e-8, **r, cpu_oracle_s_per_trajectory=cpu,
                         cpu_oracle_gradients_per_s_one_core=1.0 / cpu, sample_max_rel_du0=float(np.max(np.abs(du0[:k] - rdu0)) / np.max(np.abs(rdu0)))))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    # (c) Lorenz, non-stiff, both adaptive steppers
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3]); ts = np.linspace(0, 10, 101)
    for stepper, sname in ((sa.Rosenbrock23(), "Rosenbrock23"), (sa.Tsit5(), "Tsit5")):
        r, du0, dp = run("lorenz", u0, p, 10.0, ts, sa.LsqShift(2.0), sa.InterpolatingAdjoint(), stepper, (1e-6, 1e-6))
        rows.append(dict(case="lorenz_T10", N=N, alg="interpolating", stepper=sname, abstol=1e-6, reltol=1e-6, **r))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    print(json.dumps(dict(what="scripts/r6/bench_rosenbrock23.py", rows=rows), indent=1))


if __name__ == "__main__":
    main()
