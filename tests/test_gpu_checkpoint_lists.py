"""General `checkpoints` lists on the device (`-m gpu`; ABI 102): adjoint_sensitivities(...; checkpoints) of the reference
(src/sensitivity_interface.jl:484-486, src/interpolating_adjoint.jl:54-58, src/backsolve_adjoint.jl:523-546) with an arbitrary ascending
list — unequal spacing, not the save times, ends missing — against the oracle at rtol 1e-6."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import RTOL, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("backsolve", "BACKSOLVE")])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("segments", [1, 0])
def test_general_checkpoint_lists_on_device(sa, alg, oalg, stepper, segments):
    if stepper == "tsit5" and segments == 1:
        pytest.skip("time segmentation is a fixed-step feature")
    rng = np.random.default_rng(31)
    N, T, dt = 130, 2.0, 0.01
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
    ts = np.array([0.0, 0.4, 0.9, 1.3, 2.0])
    cks = np.array([0.07, 0.3, 0.45, 1.25, 1.8]) if stepper == "rk4" else np.array([0.0712, 0.3, 0.4567, 1.25, 1.8111])
    delta = rng.standard_normal((N, len(ts), 3))
    senses = {"interpolating": sa.InterpolatingAdjoint(checkpointing=True), "gauss": sa.GaussAdjoint(checkpointing=True), "backsolve": sa.BacksolveAdjoint()}
    tol = 1e-8
    kw = dict(dt=dt, time_segments=segments) if stepper == "rk4" else dict(abstol=tol, reltol=tol)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), sa.RK4() if stepper == "rk4" else sa.Tsit5(), saveat=ts,
                   sensealg=senses[alg], checkpoints=cks, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, sol.alg, t=ts, dgdu_discrete=delta, checkpoints=cks)
    ref = O.Problem("LORENZ", alg=oalg, stepper=stepper.upper(), t0=0, t1=T, dt=dt if stepper == "rk4" else 0.0, abstol=tol, reltol=tol, save_times=ts,
                    loss="COTANGENT", checkpointing=True, checkpoints=cks)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(sol.u, rout) < RTOL and rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    sol.engine.close()


def test_runtime_model_with_a_checkpoint_list(sa):
    """the same through a runtime-compiled right-hand side (hiprtc kernels read the same planner output)"""
    import user_models as U
    rng = np.random.default_rng(32)
    m = U.ROBER
    fun = sa.DeviceFunction("rober_ckl", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    N, T, dt = 70, 1.0, 0.005
    u0 = np.array([1.0, 0.2, 0.1]) * (1 + 0.05 * rng.standard_normal((N, 3))); p = np.array([0.04, 3.0, 1.0])
    ts = np.array([0.25, 0.5, 1.0]); cks = np.array([0.1, 0.15, 0.6])
    delta = rng.standard_normal((N, len(ts), 3))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.BacksolveAdjoint(), checkpoints=cks)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    ref = O.Problem("ROBER", alg="BACKSOLVE", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", checkpointing=True, checkpoints=cks)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    sol.engine.close()
