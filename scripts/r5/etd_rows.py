"""Timing of the exponential stepper over the documented Brusselator horizon (32 x 32, 7360 steps of dt = 1/640): forward, and the reverse pass of each sensealg.  One JSON line.
HIPADJ_LIBRARY selects a variant build (scripts/r5/ab_variants.sh style)."""
import json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
from test_gpu_parity import bruss_u0
G, dte, Se = 32, 0.0015625, 7360
tsh = 0.5 * np.arange(0, 24); rng = np.random.default_rng(0); rows = []
for alg, N in (("quadrature", 1), ("gauss", 1), ("interpolating", 1), ("interpolating", 64)):
    eng = sa.Engine("bruss", alg, N, 0.0, Se * dte, dte, save_times=tsh, dims=(G, 0, 0, 0), stepper=2)
    u0 = bruss_u0(G, N); p = np.array([3.4, 1.0, 10.0]); d = rng.standard_normal((N, len(tsh), 2 * G * G))
    eng.forward(u0, p, want_out=False); du0a, dpa = eng.adjoint(d); s0 = eng.stats(); du0, dp = eng.adjoint(d); s1 = eng.stats()
    rows.append(dict(alg=alg, N=N, forward_ms=s1["forward_ms_last"], reverse_ms=s1["adjoint_ms_total"] - s0["adjoint_ms_total"], dp=[float(x) for x in np.atleast_2d(dp)[0]], GB=s1["workspace_bytes"] / 1e9))
    eng.close()
print(json.dumps(dict(lib=os.environ.get("HIPADJ_LIBRARY", "shipped"), rows=rows)))
