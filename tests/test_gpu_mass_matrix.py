"""Constant non-singular mass matrices on the device (`-m gpu`): ODEFunction(f; mass_matrix = M) of the reference
(test/Core3/adjoint.jl:1315-1376; src/adjoint_common.jl:110-135, 805-807) for runtime-registered models, against the oracle's
restatement of the reference's formulation (M' lam' = -J' lam with jumps divided by lu(M')) and against the closed form of the
reference's own test problem (tests/golden/mass_matrix.json)."""
import json
import os
import numpy as np
import pytest

import oracle as O
import user_models as UM

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mass_matrix.json")))
_reg = {}


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def fun(sa, key, m, M, auto=False):
    if key not in _reg:
        _reg[key] = sa.DeviceFunction(key, m["n"], m["np"], m["f"], *(() if auto else (m["vjp"], m["vjp_p"])), mass_matrix=M)
    return _reg[key]


def sensealg(sa, alg, **kw):
    return {"interpolating": sa.InterpolatingAdjoint, "backsolve": sa.BacksolveAdjoint, "gauss": sa.GaussAdjoint, "quadrature": sa.QuadratureAdjoint,
            "gausskronrod": sa.GaussKronrodAdjoint}[alg](**kw)


ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE"), ("gausskronrod", "GAUSS_KRONROD")]


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_reference_mass_matrix_problem_closed_form(sa, alg, oalg):
    """test/Core3/adjoint.jl:1322-1376: `res' ≈ ForwardDiff.gradient(G)` for every sensealg (rtol 1e-11 there with 1e-14 solver
    tolerances; RK4 at dt = 1/400 reaches 6e-12 of the closed form, asserted 1e-9), du0 = lam(t0)."""
    f = fun(sa, "affine3_mm", UM.AFFINE3, UM.AFFINE3_MM)
    u0 = np.array([GOLD["u0"]]); p = np.array(GOLD["p"]); ts = np.array(GOLD["ts"])
    kw = dict(abstol=1e-13, reltol=1e-13) if alg == "quadrature" else {}
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 1.0), p), u0), sa.RK4(), dt=0.0025, saveat=ts, sensealg=sensealg(sa, alg, **kw))
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=np.ones((1, len(ts), 3)))
    assert rel(sol.u[0, -1], GOLD["u_end"]) < 1e-9
    assert rel(dp, GOLD["dGdp"]) < 1e-9
    assert rel(du0[0], GOLD["lam0"]) < 1e-8                       # the reference's du0: lam(t0), not M' lam(t0)
    assert rel(np.array(GOLD["M"]).T @ du0[0], GOLD["dGdu0"]) < 1e-8
    sol.engine.close()


def test_reference_mass_matrix_problem_tsit5(sa):
    f = fun(sa, "affine3_mm", UM.AFFINE3, UM.AFFINE3_MM)
    u0 = np.array([GOLD["u0"]]); p = np.array(GOLD["p"]); ts = np.array(GOLD["ts"])
    for alg in ("interpolating", "gauss", "backsolve"):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 1.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sensealg(sa, alg), abstol=1e-12, reltol=1e-12)
        du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), 3)))
        assert rel(dp, GOLD["dGdp"]) < 1e-8 and rel(du0[0], GOLD["lam0"]) < 1e-8
        sol.engine.close()


@pytest.mark.parametrize("auto", [False, True])
@pytest.mark.parametrize("alg,oalg", ALGS)
def test_mass_matrix_ensemble_matches_oracle(sa, alg, oalg, auto):
    """A nonlinear model (Robertson kinetics, mild rates) with a dense well-conditioned M, ensemble with per-trajectory parameters,
    random cotangents: device (nu = M' lam formulation) vs oracle (the reference's lam formulation) to round-off; hand VJPs and
    VJPs by dual numbers (autojacvec = true)."""
    rng = np.random.default_rng(97)
    M = np.eye(3) * 2.0 + 0.4 * rng.standard_normal((3, 3))
    f = fun(sa, "rober_mm" + ("_auto" if auto else ""), UM.ROBER, M, auto=auto)
    N, T, dt = 70, 2.0, 0.01
    u0 = rng.uniform(0.3, 1.0, (N, 3)); pp = rng.uniform(0.4, 1.2, (N, 3))
    ts = np.arange(0, T + 1e-9, 0.25)
    delta = rng.standard_normal((N, len(ts), 3))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), pp[0]), u0, pp), sa.RK4(), dt=dt, saveat=ts, sensealg=sensealg(sa, alg))
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    with O.mass_matrix(M):
        ref = O.Problem("ROBER", alg=oalg, stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", checkpointing=(alg == "backsolve"))
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(sol.u, rout) < 1e-10 and rel(du0, rdu0) < 1e-9 and rel(dp, rdp) < 1e-9
    sol.engine.close()


def test_mass_matrix_checkpointed_and_shared_parameters(sa):
    """InterpolatingAdjoint(checkpointing = true) with checkpoints = sol.t[1:10:end] (test/Core3/adjoint.jl:1362-1369), shared p, LSQ loss, continuous cost."""
    rng = np.random.default_rng(98)
    M = np.array([[1.5, 0.2, 0.0], [0.1, 0.8, -0.3], [0.0, 0.4, 2.0]])
    f = fun(sa, "rober_mm2", UM.ROBER, M)
    N, T, dt = 130, 2.0, 0.01
    u0 = rng.uniform(0.3, 1.0, (N, 3)); p = np.array([0.5, 0.9, 0.7])
    ts = np.arange(0, T + 1e-9, 0.1)
    for cont in (0, 1):
        g = sa.HalfSquaredSum() if cont else None
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(checkpointing=True),
                       checkpoints=np.arange(0, T + 1e-9, 0.1), dgdu_discrete=sa.LsqShift(0.5), g=g)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqShift(0.5), g=g)
        with O.mass_matrix(M):
            ref = O.Problem("ROBER", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=0.5, checkpointing=True,
                            checkpoints=np.arange(0, T + 1e-9, 0.1), cont_cost=cont)
            rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p)
        assert rel(du0, rdu0) < 1e-9 and rel(dp, rdp) < 1e-9
        sol.engine.close()


def test_mass_matrix_can_be_removed_and_identity_is_a_no_op(sa):
    m = UM.ROBER
    plain = sa.DeviceFunction("rober_plain_mm", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    rng = np.random.default_rng(99)
    N, T, dt = 64, 1.0, 0.01
    u0 = rng.uniform(0.3, 1.0, (N, 3)); p = np.array([0.5, 0.9, 0.7]); ts = np.array([0.5, 1.0])

    def run():
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(plain, u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(0.0))
        r = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqShift(0.0)); sol.engine.close(); return r
    a = run()
    plain.set_mass_matrix(np.eye(3)); b = run()
    plain.set_mass_matrix(np.diag([2.0, 2.0, 2.0])); c = run()
    plain.set_mass_matrix(None); d = run()
    assert rel(b[0], a[0]) < 1e-14 and rel(b[1], a[1]) < 1e-14 and rel(d[0], a[0]) == 0 and rel(d[1], a[1]) == 0
    assert rel(c[1], a[1]) > 1e-3                                  # a different system


@pytest.mark.parametrize("auto", [False, True])
@pytest.mark.parametrize("nring", [5, 6])
def test_wide_ring_dense_mass_matrix_gauss_regression(sa, nring, auto):
    """Regression: 5- / 6-state ring (one-column, non-segmented sweeps) behind a dense mass matrix, GaussAdjoint + RK4.  Compiled by the hiprtc a
    torch wheel bundles (ROCm 7.0) this kernel returned parameter gradients wrong from the 6th digit up to non-finite; the library now binds
    the build toolkit's compiler (hipadj_runtime_compiler)."""
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    assert "HIP 7.0" not in _lib.runtime_compiler()
    rng = np.random.default_rng(5)
    m = UM.ring(nring); n, npar = m["n"], m["np"]
    f = sa.DeviceFunction(f"ring{nring}_mmreg{int(auto)}", n, npar, m["f"], *(() if auto else (m["vjp"], m["vjp_p"])))
    for N in (3, 54):
        u0 = rng.uniform(0.3, 1.0, (N, n)); ts = np.array([0.4, 1.1, 2.0]); delta = rng.standard_normal((N, 3, n))
        M = np.eye(n) * 1.5 + 0.3 * rng.standard_normal((n, n))
        f.set_mass_matrix(M)
        pp = rng.uniform(0.4, 1.2, (N, npar))
        for alg, oalg in (("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD"), ("interpolating", "INTERPOLATING")):
            sens = dict(gauss=sa.GaussAdjoint, gausskronrod=sa.GaussKronrodAdjoint, interpolating=sa.InterpolatingAdjoint)[alg]()
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 2.0), pp[0], (nring, 0, 0, 0)), u0, pp), sa.RK4(), dt=0.01, saveat=ts, sensealg=sens)
            du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
            with O.mass_matrix(M):
                ref = O.Problem("RING", alg=oalg, t0=0.0, t1=2.0, save_times=ts, loss="COTANGENT", dims=(nring, 0, 0, 0), stepper="RK4", dt=0.01)
                rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
            assert rel(sol.u, rout) < 1e-12 and rel(du0, rdu0) < 1e-10 and rel(dp, rdp) < 1e-9, (nring, auto, N, alg)
            sol.engine.close()
    f.set_mass_matrix(None)
