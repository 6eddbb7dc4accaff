import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scimlsensitivity_jl_amd as sa
import bench
N = int(os.environ.get("DBG_N", "64"))
T = float(os.environ.get("DBG_T", "10.0"))
u0, p = bench.inputs(N)
ts = np.linspace(0, T, int(os.environ.get("DBG_M", "101")))
res = {}
for quad in ("1", "0"):
    os.environ["HIPADJ_QUAD"] = quad
    eng = sa.Engine("lorenz", "gauss", N, 0.0, T, 0.0, save_times=ts, loss_kind=1, loss_shift=2.0, p_shared=False, stepper=1, abstol=1e-8, reltol=1e-8)
    eng.forward(u0, np.tile(p, (N, 1)), want_out=False)
    res[quad] = eng.adjoint(None)
    eng.close()
d = np.abs(res["1"][0] - res["0"][0]).max(axis=1) / np.abs(res["0"][0]).max(axis=1)
print("N", N, "T", T, "M", len(ts), "worst rel du0", d.max(), "bad trajectories", np.nonzero(d > 1e-6)[0][:20], "count", int((d > 1e-6).sum()))
dd = np.abs(res["1"][1] - res["0"][1]).max(axis=1) / np.abs(res["0"][1]).max(axis=1)
print("worst rel dp", dd.max(), "bad", int((dd > 1e-6).sum()))
# cotangent loss instead of the fused lsq loss
rng = np.random.default_rng(0)
delta = rng.standard_normal((N, len(ts), 3))
res = {}
for quad in ("1", "0"):
    os.environ["HIPADJ_QUAD"] = quad
    eng = sa.Engine("lorenz", "gauss", N, 0.0, T, 0.0, save_times=ts, p_shared=False, stepper=1, abstol=1e-8, reltol=1e-8)
    eng.forward(u0, np.tile(p, (N, 1)), want_out=False)
    res[quad] = eng.adjoint(delta)
    eng.close()
d = np.abs(res["1"][0] - res["0"][0]).max(axis=1) / np.abs(res["0"][0]).max(axis=1)
print("cotangent loss: worst rel du0", d.max(), "count", int((d > 1e-6).sum()))
