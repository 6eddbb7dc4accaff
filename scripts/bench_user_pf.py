#!/usr/bin/env python
"""Runtime-registered wide models (ring, n = 4..8 states): knot-prefetch depth of the reverse sweeps (HIPADJ_USER_PF for k_interp / k_quad_adj,
HIPADJ_USER_PFG for k_gauss) against the oracle and the clock.  10^4 trajectories, RK4 dt = 0.01, 1000 steps, 100 loss times.
One JSON line per (n, sensealg, depth).  Development tooling (the depths the library ships are in user_kernel_names, csrc/hipadj_api.hip)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def main():
    os.environ.setdefault("HIPADJ_TIMING", "1")
    import scimlsensitivity_jl_amd as sa
    import oracle as O
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    print(json.dumps(dict(compiler=_lib.runtime_compiler())), flush=True)
    depths = [int(x) for x in os.environ.get("PF_DEPTHS", "1,2,4").split(",")]
    N, S, dt = int(os.environ.get("PF_NTRAJ", "10000")), 1000, 0.01
    ts = dt * np.arange(10, S + 1, 10)
    rng = np.random.default_rng(0)
    named = dict(lv=(UM.LV, "LV"), rober=(UM.ROBER, "ROBER"))      # PF_SIZES accepts these names next to ring sizes (polynomial right-hand sides)
    for key in os.environ.get("PF_SIZES", "4,5,6,8").split(","):
        if key in named:
            m, omodel = named[key]; n = m["n"]; odims = (0, 0, 0, 0)
        else:
            n = int(key); m = UM.ring(n); omodel = "RING"; odims = (n, 0, 0, 0)
        npar = m["np"]
        auto = os.environ.get("PF_AUTO") == "1"                       # dual-number VJPs (only f is registered)
        f = sa.DeviceFunction(f"{key}_pf{int(auto)}", n, npar, m["f"]) if auto else sa.DeviceFunction(f"{key}_pf", n, npar, m["f"], m["vjp"], m["vjp_p"])
        u0 = rng.uniform(0.3, 1.0, (N, n)); pp = rng.uniform(0.4, 1.2, (N, npar)); delta = rng.standard_normal((N, len(ts), n))
        nref = 16
        for alg in os.environ.get("PF_ALGS", "interpolating,gauss").split(","):
            ref = O.Problem(omodel, alg=alg.upper(), t0=0.0, t1=S * dt, save_times=ts, loss="COTANGENT", dims=odims, stepper="RK4", dt=dt)
            rdu0, rdp, _, _ = ref.adjoint_ensemble(u0[:nref], pp[:nref], delta[:nref])
            for d in depths:
                if d > 0:
                    os.environ["HIPADJ_USER_PF"] = str(d); os.environ["HIPADJ_USER_PFG"] = str(d)
                else:                                       # depth 0: the library's own choice (user_kernel_names)
                    os.environ.pop("HIPADJ_USER_PF", None); os.environ.pop("HIPADJ_USER_PFG", None)
                sens = dict(interpolating=sa.InterpolatingAdjoint, gauss=sa.GaussAdjoint, quadrature=sa.QuadratureAdjoint, backsolve=sa.BacksolveAdjoint)[alg]()
                t0 = time.perf_counter()
                try:
                    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, S * dt), pp[0], odims), u0, pp), sa.RK4(), dt=dt, saveat=ts, sensealg=sens)
                except Exception as e:
                    print(json.dumps(dict(n=n, alg=alg, depth=d, error=str(e)[:300])), flush=True); continue
                build_s = time.perf_counter() - t0
                eng = sol.engine
                du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
                s0 = eng.stats()
                reps = 5
                for _ in range(reps):
                    du0, dp = eng.adjoint(delta)
                s1 = eng.stats()
                print(json.dumps(dict(model=key, n=n, np=npar, alg=alg, depth=d, N=N, steps=S, time_segments=s1.get("time_segments"), forward_ms=s1.get("forward_ms_last"),
                                      adjoint_ms=(s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / reps,
                                      main_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / reps,
                                      err_du0=rel(du0[:nref], rdu0), err_dp=rel(dp[:nref], rdp), build_s=round(build_s, 1))), flush=True)
                eng.close()


if __name__ == "__main__":
    main()
