import sys, numpy as np
sys.path.insert(0, ".")
import scimlsensitivity_jl_amd as sa
which = sys.argv[1]
alg = {"i": sa.InterpolatingAdjoint(), "b": sa.BacksolveAdjoint(), "g": sa.GaussAdjoint(), "bn": sa.BacksolveAdjoint(checkpointing=False)}[which]
N = int(sys.argv[2])
u0 = np.array([1.0, 1.0]) + 0.01 * np.arange(N)[:, None]; p = np.array([1.5, 1.0, 3.0, 1.0])
ts = np.array([0.0, 0.5, 1.0])
from scimlsensitivity_jl_amd.engine import Engine
from scimlsensitivity_jl_amd import _lib
eng = Engine("lv", alg.name, N, 0.0, 1.0, 0.0, save_times=ts, loss_kind=_lib.LOSS_LSQ_SHIFT, loss_shift=2.0, stepper=1,
             abstol=1e-8, reltol=1e-8, checkpointing=getattr(alg, "checkpointing", False))
print("created", flush=True)
out = eng.forward(u0, p)
print("forward ok", out[0].ravel(), flush=True)
du0, dp = eng.adjoint(None)
print("adjoint ok", du0[0], dp, flush=True)
