"""PCIe-inclusive rate of the host-pointer C ABI (hipadj_forward / hipadj_adjoint) on BASELINE configs[1]:
u0 goes in (240 KB), du0 (240 KB) and dp come out; the interpolant never leaves the device."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scimlsensitivity_jl_amd as sa
import bench
N = 10000
u0, p = bench.inputs(N)
ts = np.arange(0, 10.0 + 1e-9, 0.1)
sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, 10.0), p), u0), sa.RK4(), dt=0.01, saveat=ts, sensealg=sa.InterpolatingAdjoint(),
               dgdu_discrete=sa.LsqShift(2.0), want_out=False)
eng = sol.engine
for _ in range(5):
    eng.adjoint(None)
t0 = time.perf_counter(); K = 50
for _ in range(K):
    du0, dp = eng.adjoint(None)
adj = (time.perf_counter() - t0) / K
t0 = time.perf_counter()
for _ in range(10):
    eng.forward(u0, p, want_out=False)
fwd = (time.perf_counter() - t0) / 10
print(json.dumps(dict(case="host-pointer API, Lorenz N=1e4 InterpolatingAdjoint RK4", adjoint_host_call_ms=adj * 1e3, forward_host_call_ms=fwd * 1e3,
                      traj_per_s_pcie_inclusive=N / adj)))
