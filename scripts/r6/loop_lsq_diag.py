#!/usr/bin/env python
"""Diagnosis of an intermittent mismatch between the two least-squares routes (model body vs HIPADJ_LOSS_LSQ_DATA) seen once in a full 4-worker GPU suite and reproduced only under
concurrent load from other processes: every (configuration, route) is run three times; prints which runs disagree with the majority and in which outputs."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
sa.build_extension()
import test_gpu_device_loss as T

def run(f, loss, u0, p, ts, sens, T_, dt):
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T_), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=loss)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss)
    out = None if sol.u is None else np.array(sol.u)
    sol.engine.close()
    return out, du0, dp

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(9)
    N, T_, dt = 66, 2.0, 0.01
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0])
    cfgs = [(np.linspace(0, T_, 11), "interp_ckpt"), (np.array([0.503, 1.0, 1.777, 2.0]), "interp"), (np.array([0.503, 1.0, 1.777, 2.0]), "gauss"), (np.array([0.503, 1.0, 1.777, 2.0]), "backsolve")]
    mk = {"interp_ckpt": lambda: sa.InterpolatingAdjoint(checkpointing=True), "interp": sa.InterpolatingAdjoint, "gauss": sa.GaussAdjoint, "backsolve": sa.BacksolveAdjoint}
    datas = [rng.standard_normal((N, len(ts), 2)) for ts, _ in cfgs]
    ref = {}
    bad = 0
    for it in range(iters):
        for k, (ts, name) in enumerate(cfgs):
            for route in ("model", "builtin"):
                f, loss = (T._lv_with_loss(sa, "lsq"), sa.ModelLoss(datas[k])) if route == "model" else ("lv", sa.LsqData(datas[k], 2.0))
                out, du0, dp = run(f, loss, u0, p, ts, mk[name](), T_, dt)
                key = (name, route)
                if key not in ref:
                    ref[key] = (out, du0, dp)
                else:
                    r = ref[key]
                    d = [float(np.max(np.abs(a - b))) for a, b in zip((out, du0, dp), r)]
                    if max(d) > 0:
                        bad += 1
                        rows = np.where(np.any(du0 != r[1], axis=1))[0]
                        print(json.dumps(dict(iteration=it, config=name, route=route, max_abs_diff_out=d[0], max_abs_diff_du0=d[1], max_abs_diff_dp=d[2], rows_differing=int(len(rows)), first_rows=rows[:8].tolist(),
                                              out_rows=np.where(np.any(out != r[0], axis=(1, 2)))[0][:8].tolist())), flush=True)
    print(json.dumps(dict(iterations=iters, mismatches=bad)))

if __name__ == "__main__":
    main()
