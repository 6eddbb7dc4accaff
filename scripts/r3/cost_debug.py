import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
import oracle as O
def rel(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
T, dt = 1.0, 0.02; ts = np.linspace(0, T, 6)
for model, kind in (("index", 1), ("chain", 1), ("linear", 2), ("linear", 1), ("chain", 2), ("index", 2)):
    rng = np.random.default_rng(31 + kind)
    if model == "linear":
        n = 24; fun = sa.WideDeviceFunction.dense_linear(f"dbg_lin_{kind}", n); oname, dims = "DENSELIN", (n, 0, 0, 0)
        p = (rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F")
    elif model == "index":
        R, Cc = 12, 9; n = R * Cc; fun = sa.WideDeviceFunction.index_affine(f"dbg_idx_{kind}", R, Cc); oname, dims = "IDXAFF", (R, Cc, 0, 0)
        p = 0.2 * rng.random(2)
    else:
        n, H = 3, 16; fun = sa.WideDeviceFunction.dense_chain(f"dbg_chain_{kind}", (n, H, n)); oname, dims = "MLP1", (n, H, 0, 0)
        p = np.concatenate([rng.standard_normal(H * n) * 0.4, 0.1 * rng.standard_normal(H), rng.standard_normal(n * H) * 0.3, 0.1 * rng.standard_normal(n)])
    u0 = 0.5 * rng.standard_normal((3, n))
    g = sa.FirstStateSquaredPlusFirstParam() if kind == 2 else sa.HalfSquaredSum()
    res = {}
    for gg in (None, g):
        kw = dict(g=gg) if gg is not None else {}
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(0.3), **kw)
        res[gg is not None] = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqShift(0.3), **kw)
        sol.engine.close()
    ref = {}
    for cc in (0, kind):
        pr = O.Problem(oname, alg="INTERPOLATING", t0=0.0, t1=T, save_times=ts, loss="LSQ_SHIFT", loss_shift=0.3, dims=dims, cont_cost=cc, stepper="RK4", dt=dt)
        ref[cc] = pr.adjoint_ensemble(u0, p)
    print(f"{model} kind {kind}: device(no cost) vs oracle(no cost) du0 {rel(res[False][0], ref[0][0]):.1e} dp {rel(res[False][1], ref[0][1]):.1e} | device(cost) vs oracle(cost) du0 {rel(res[True][0], ref[kind][0]):.1e} dp {rel(res[True][1], ref[kind][1]):.1e}"
          f" | device(cost) vs oracle(no cost) du0 {rel(res[True][0], ref[0][0]):.1e} | oracle cost vs no cost du0 {rel(ref[kind][0], ref[0][0]):.1e}")
    d = res[True][0] - ref[kind][0]
    print("   du0 diff rows:", np.abs(d).max(axis=1), " first comps:", d[0][:4])
