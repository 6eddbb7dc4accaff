import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scimlsensitivity_jl_amd as sa
import oracle as O
import bench
N, T, M = 8, 1.0, 11
u0, p = bench.inputs(N)
ts = np.linspace(0, T, M)
import itertools
for tol, (alg, oalg) in itertools.product((1e-8, 1e-11), (("gauss", "GAUSS"), ("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"))):
    ref = O.Problem("LORENZ", alg=oalg, checkpointing=(alg == "backsolve"), stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=tol, reltol=tol, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, np.tile(p, (N, 1)))
    for quad in ("1", "0"):
        os.environ["HIPADJ_QUAD"] = quad
        eng = sa.Engine("lorenz", alg, N, 0.0, T, 0.0, save_times=ts, loss_kind=1, loss_shift=2.0, p_shared=False, stepper=1, abstol=tol, reltol=tol, checkpointing=(alg == "backsolve"))
        eng.forward(u0, np.tile(p, (N, 1)), want_out=False)
        du0, dp = eng.adjoint(None)
        print(alg, "tol", tol, "quad", quad, "du0 vs oracle", np.max(np.abs(du0 - rdu0)) / np.max(np.abs(rdu0)), "dp", np.max(np.abs(dp - rdp)) / np.max(np.abs(rdp)))
        eng.close()
