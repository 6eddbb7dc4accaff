"""Which build of the one-launch reverse kernels of 5..8-state runtime lane models is right?  (GPU visit 8)

For ring models n = 5..8 and Interpolating / Gauss / Backsolve: the three-launch sequence (HIPADJ_FUSED=0) is the reference (it passed the
whole suite in visit 5 and agrees with the oracle); the fused kernel is run as the -O3 build alone, the -O1 build alone, and with the
self-test.  Prints the relative differences of du0 / dp and the scratch sizes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
import user_models as UM


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run(f, n, npar, alg, N, env, shared):
    keys = ("HIPADJ_FUSED", "HIPADJ_RTC_SELFTEST", "HIPADJ_RTC_FORCE_O1")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        rng = np.random.default_rng(47)
        u0 = rng.uniform(0.3, 1.0, (N, n)); pp = rng.uniform(0.4, 1.2, npar) if shared else rng.uniform(0.4, 1.2, (N, npar))
        ts = np.array([0.37, 0.9, 1.44, 2.0]); delta = rng.standard_normal((N, len(ts), n))
        sens = {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(), "gauss": sa.GaussAdjoint()}[alg]
        prob = sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 2.0), pp if shared else pp[0]), u0, pp)
        sol = sa.solve(prob, sa.RK4(), dt=0.01, saveat=ts, sensealg=sens)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
        du0b, dpb = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)     # second pass: determinism / the build kept after the self-test
        st = sol.engine.stats() if hasattr(sol.engine, "stats") else None
        sol.engine.close()
        return du0, dp, du0b, dpb, st
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


for n in (5, 6, 7, 8):
    m = UM.ring(n)
    f = sa.DeviceFunction(f"probe_ring{n}", n, m["np"], m["f"], m["vjp"], m["vjp_p"])
    for alg in ("interpolating", "gauss", "backsolve"):
        for N, shared in ((61, False), (300, True)):
            ref = run(f, n, m["np"], alg, N, {"HIPADJ_FUSED": "0"}, shared)
            row = [f"ring{n} {alg:13s} N={N:3d} shared={int(shared)}"]
            for label, env in (("O3", {"HIPADJ_RTC_SELFTEST": "0"}), ("O1", {"HIPADJ_RTC_FORCE_O1": "1"}), ("selftest", {})):
                sys.stderr.write(f"-- ring{n} {alg} N={N} {label}\n"); sys.stderr.flush()
                r = run(f, n, m["np"], alg, N, env, shared)
                row.append(f"{label}: du0 {rel(r[0], ref[0]):.1e} dp {rel(r[1], ref[1]):.1e} | 2nd du0 {rel(r[2], ref[0]):.1e} dp {rel(r[3], ref[1]):.1e}")
            print("   ".join(row), flush=True)
