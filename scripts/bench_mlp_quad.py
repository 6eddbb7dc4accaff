#!/usr/bin/env python
"""QuadratureAdjoint on BASELINE configs[3]'s shape (MLP 2-128-128-2, 4096 columns, 150 steps, 30 loss intervals): device times."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_families import run
import scimlsensitivity_jl_amd as sa
from test_gpu_parity import mlp_params
rng = np.random.default_rng(0)
d, H, B, S, dt = 2, 128, 4096, 150, 0.01
ts = dt * np.arange(5, S + 1, 5)
u0 = rng.standard_normal((1, d * B)); p = mlp_params(d, H); delta = rng.standard_normal((1, len(ts), d * B))
eng = sa.Engine("mlp", "quadrature", 1, 0.0, S * dt, dt, save_times=ts, dims=(d, H, B, 0))
r, du0, dp = run(eng, u0, p, delta, 2)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
eng.close()
