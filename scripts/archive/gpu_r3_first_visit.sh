#!/bin/bash
# First GPU visit of the next round (prepared at the end of round 2, when the GPU budget was spent): re-confirm the final code of round 2 and run the
# two checks that could not be run any more:  (1) traced models with bundle-ready VJP bodies (DeviceFunction.from_callable(..., bundle=True)) against
# the compiled-in Lorenz and the oracle;  (2) the heavy-kernel self-test note for the 8-state dual-number ring (stderr).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r3_first_visit.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r3v1; mkdir -p $OUT; cd $REPO
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/traced_bundle.log
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scimlsensitivity_jl_amd as sa, oracle as O
def rel(a, b): return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
def lorenz(du, u, p, t):
    du[0] = p[0] * (u[1] - u[0]); du[1] = u[0] * (p[1] - u[2]) - u[1]; du[2] = u[0] * u[1] - p[2] * u[2]
rng = np.random.default_rng(0); N, T, dt = 500, 2.0, 0.01
u0 = np.array([1.0, 0, 0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3]); ts = np.arange(0.1, T + 1e-9, 0.1)
delta = rng.standard_normal((N, len(ts), 3))
for alg, oalg in ((sa.InterpolatingAdjoint(), "INTERPOLATING"), (sa.GaussAdjoint(), "GAUSS"), (sa.BacksolveAdjoint(), "BACKSOLVE")):
    ref = O.Problem("LORENZ", alg=oalg, t0=0, t1=T, save_times=ts, loss="COTANGENT", stepper="RK4", dt=dt, checkpointing=(oalg == "BACKSOLVE"))
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    for bundle in (False, True):
        f = sa.DeviceFunction.from_callable(f"lorenz_traced_{int(bundle)}", lorenz, 3, 3, bundle=bundle)
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=alg, time_segments=4)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
        print(oalg, "bundle", bundle, "du0 %.1e dp %.1e" % (rel(du0, rdu0), rel(dp, rdp)), flush=True)
        sol.engine.close()
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "cross_checked" -s 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/selftest.log
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 2>&1 | tail -30 | tee $OUT/pytest_full.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 600 $OUT/bench.json; echo
