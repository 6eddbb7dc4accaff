import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scimlsensitivity_jl_amd as sa
import user_models as UM, oracle as O
T = 2.0; ts = np.arange(0, T + 1e-9, 0.25)
for n in (2, 3, 4, 5):
    m = UM.ring(n)
    f = sa.DeviceFunction(f"ring{n}_dbg", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    u0 = np.array([[0.5, 0.7, 0.9, 0.4, 0.6][:n]]); p = np.array([0.5, 0.8, 1.0, 0.6, 0.7][:n] + [0.9])
    for alg, oalg in ((sa.BacksolveAdjoint(), "BACKSOLVE"), (sa.BacksolveAdjoint(checkpointing=False), "BACKSOLVE"), (sa.InterpolatingAdjoint(), "INTERPOLATING")):
        for tol in (1e-6, 1e-9):
            ck = getattr(alg, "checkpointing", False)
            try:
                sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=alg, dgdu_discrete=sa.LsqShift(0.0), abstol=tol, reltol=tol)
                du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=sa.LsqShift(0.0))
                ref = O.Problem("RING", alg=oalg, t0=0, t1=T, save_times=ts, loss="LSQ_SHIFT", checkpointing=ck, dims=(n, 0, 0, 0), stepper="TSIT5", dt=0.0, abstol=tol, reltol=tol)
                r = ref.adjoint(u0[0], p)
                print(n, alg.name, ck, tol, "rel du0 %.2e dp %.2e" % (np.abs(du0[0] - r[0]).max() / np.abs(r[0]).max(), np.abs(dp - r[1]).max() / np.abs(r[1]).max()), flush=True)
                sol.engine.close()
            except Exception as e:
                print(n, alg.name, ck, tol, "ERR", str(e)[:100], flush=True)
