#!/usr/bin/env python
"""dense vs checkpointed Interpolating sweep of the 2-50-2 chain over many seeds, with the pair sum (wg_sum2) and with two single sums in the bodies: how often dp differs in the last bit."""
import os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scimlsensitivity_jl_amd as sa
base = sa.WideDeviceFunction.dense_chain("ck2_pair", (2, 50, 2), input_power=3)
def single(text):
    return re.sub(r"wg_sum2\((\w+), (\w+), (\w+), (\w+)\);", r"\3 = wg_sum(\1); \4 = wg_sum(\2);", text)
alt = sa.WideDeviceFunction("ck2_single", 2, 252, single(base.source["f"]), single(base.source["vjp"]), lds_doubles=1)
n, npar, N, T, dt = 2, 252, 4, 0.6, 0.01
ts = np.array([0.0, 0.1, 0.25, 0.4, 0.6])
for fun in (base, alt):
    bad = []
    for seed in range(20, 60):
        rng = np.random.default_rng(seed)
        u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(-0.4, 0.4, npar); delta = rng.standard_normal((N, len(ts), n))
        res = []
        for ck in (False, True):
            eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, dt, save_times=ts, checkpointing=ck, **(dict(ckpt_stride=7) if ck else {}))
            eng.forward(u0, p); res.append(eng.adjoint(delta)); eng.close()
        if not (np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])):
            bad.append((seed, int(np.count_nonzero(res[0][0] != res[1][0])), np.nonzero(res[0][1] != res[1][1])[0].tolist()))
    print(fun.name, "seeds with a difference:", bad)
