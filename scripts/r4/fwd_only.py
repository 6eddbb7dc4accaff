#!/usr/bin/env python
"""Forward solves only (bench.py's ensemble), for rocprofv3 --kernel-trace --stats: RK4 (default) or adaptive Tsit5 (FWD_TS5=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scimlsensitivity_jl_amd as sa
import bench
N = int(os.environ.get("FWD_N", "10000"))
u0, p = bench.inputs(N)
ts5 = os.environ.get("FWD_TS5", "0") == "1"
alg = os.environ.get("FWD_ALG", "interpolating")
eng = sa.Engine("lorenz", alg, N, 0.0, bench.T_FINAL, 0.0 if ts5 else bench.DT, save_times=bench.save_times(), loss_kind=1, loss_shift=bench.LOSS_SHIFT, p_shared=True,
                stepper=1 if ts5 else 0, checkpointing=(alg == "backsolve"))
for _ in range(8):
    eng.forward(u0, p, want_out=False)
st = eng.stats()
print("forward_ms_last", st["forward_ms_last"])
if os.environ.get("FWD_REV", "0") == "1":
    for _ in range(4):
        eng.adjoint(None)
    print("adjoint_ms_last", eng.stats()["adjoint_ms_last"])
eng.close()
