"""Rosenbrock23, the stiff stepper of the lane family (HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE; reference use: test/Core2/stiff_adjoints.jl:66-80, 142-157, 191) — CPU side:

  * the oracle's restatement against gradients computed independently of it (tests/golden/stiff_adjoints.json, scipy forward sensitivities; generator committed beside it), at
    the reference's own settings and bars: Lotka-Volterra fit, Rosenbrock23 abstol = reltol = 1e-8, rtol 1e-3 in place (:80) and 1e-4 out of place (:157) — the tighter one here;
  * the same on Robertson kinetics at the classic stiff rates (0.04, 3e7, 1e4) over (0, 100): the problem class the stepper exists for;
  * the device lane bodies (hipadj_adaptive.hpp ros23_integrate + the adjoint's block-triangular W solve), compiled for the host by tests/emu, against the oracle;
  * what the planner refuses for this stepper;
  * the one constant of the restatement that the reference's relations cannot pin (the coefficient of dT in k3): both readings pass the reference's bar, one takes 10 x the steps.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import emu as E
import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ALGS = [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE"), ("gausskronrod", "GAUSS_KRONROD")]
ROS = 3      # HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


def relc(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b) / np.abs(b)))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "stiff_adjoints.json")) as f:
        return json.load(f)


def test_golden_fixture_is_what_its_generator_writes(gold, tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stiff_adjoints", os.path.join(HERE, "golden", "make_stiff_adjoints.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    lv = m.lv()
    assert rel(lv["dp"], gold["lv"]["dp"]) < 1e-9 and rel(lv["target"], gold["lv"]["target"]) < 1e-10
    # closed-form anchor of the Robertson fixture: mass conservation, y1 + y2 + y3 = 1 at both loss times, and d(sum)/dp = 0 => the three state gradients differ only through y3
    u = np.asarray(gold["rober"]["u_at_ts"])
    assert np.max(np.abs(u.sum(axis=1) - 1.0)) < 1e-10
    assert gold["rober"]["spread_between_tolerances"] < 1e-9


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_oracle_lotka_volterra_fit_at_the_reference_bar(gold, alg, oalg):
    """test/Core2/stiff_adjoints.jl:142-157: Rosenbrock23, abstol = reltol = 1e-8, saveat 0:0.5:10, loss = sum(abs2, prediction - target); `fdgrad ≈ rdgrad rtol = 1e-4`."""
    c = gold["lv"]
    pr = O.Problem("LV", alg=oalg, stepper="ROS23", t0=0.0, t1=10.0, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=c["ts"], loss="LSQ_DATA", loss_scale=2.0, quad_abstol=1e-8, quad_reltol=1e-8)
    du0, dp, out = pr.adjoint(c["u0"], c["p"], np.asarray(c["target"]))
    assert rel(dp, c["dp"]) < 1e-4 and rel(du0, c["du0"]) < 1e-4       # measured: 1.2e-5 / 4e-5
    loss = float(((out - np.asarray(c["target"])) ** 2).sum())
    assert abs(loss - c["loss"]) < 1e-4 * c["loss"]


@pytest.mark.parametrize("tol,bar", [(1e-4, 1e-2), (1e-6, 5e-4), (1e-8, 1e-5)])
def test_oracle_lotka_volterra_converges_with_the_tolerance(gold, tol, bar):
    c = gold["lv"]
    pr = O.Problem("LV", alg="INTERPOLATING", stepper="ROS23", t0=0.0, t1=10.0, dt=0.0, abstol=tol, reltol=tol, save_times=c["ts"], loss="LSQ_DATA", loss_scale=2.0)
    du0, dp, _ = pr.adjoint(c["u0"], c["p"], np.asarray(c["target"]))
    assert relc(dp, c["dp"]) < 5 * bar, relc(dp, c["dp"])      # measured 4.8e-3 / 2.7e-4 / 1.3e-5 (componentwise): second order in the step, as the method


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_oracle_robertson_at_the_stiff_rates(gold, alg, oalg):
    """p = (0.04, 3e7, 1e4) (test/Core3/adjoint.jl:1458), G = y3(50) + y3(100) (:1465-1466): the gradient against Radau forward sensitivities, componentwise — dG/dp spans nine
    orders of magnitude.  570 forward and ~5000 reverse steps at 1e-8; an explicit stepper needs ~1e6."""
    c = gold["rober"]
    d = np.zeros((2, 3)); d[:, 2] = 1.0
    pr = O.Problem("ROBER", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-8, reltol=1e-6, save_times=c["ts"], loss="COTANGENT", quad_abstol=1e-12, quad_reltol=1e-6)
    du0, dp, out = pr.adjoint(c["u0"], c["p"], d)
    assert relc(dp, c["dp"]) < 1e-3 and relc(du0, c["du0"]) < 1e-3      # measured 8.5e-5 / 6.9e-5
    assert np.max(np.abs(out - np.asarray(c["u_at_ts"]))) < 1e-5


MODELS = [("lv", "LV", [1.0, 1.0], [1.5, 1.0, 3.0, 1.0]), ("lvt", "LVT", [1.0, 1.0], [1.5, 1.0, 3.0, 1.0]), ("lorenz", "LORENZ", [1.0, 0.0, 0.0], [10.0, 28.0, 8 / 3])]


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("model,omodel,u0c,p", MODELS)
def test_lane_bodies_match_oracle(alg, oalg, model, omodel, u0c, p):
    """The device's stepper and its adjoint W solve (n x n LU of I + d h J', then the parameter rows by substitution) against the oracle's dense (n + np) LU of the same system:
    the same controller, the same accept / reject sequence; `lvt` is the non-autonomous forward problem (dT by differences on the forward pass too).  Loss times off any grid."""
    rng = np.random.default_rng(14)
    N, T = 3, 2.0
    n, npar = len(u0c), len(p)
    u0 = np.asarray(u0c) + 0.05 * rng.standard_normal((N, n))
    pp = np.asarray(p) * (1 + 0.03 * rng.standard_normal((N, npar)))
    ts = np.array([0.0, 0.13, 0.5, 0.77, 1.0, 1.9, 2.0])
    delta = rng.standard_normal((N, len(ts), n))
    cfg = E.make_config(model, alg, N, 0.0, T, 0.0, ts, loss_kind=0, p_shared=False, stepper=ROS, abstol=1e-8, reltol=1e-8, quad_abstol=1e-8, quad_reltol=1e-8, max_steps=100000)
    du0, dp, out = E.forward_adjoint(cfg, n, npar, u0, pp, delta)
    ref = O.Problem(omodel, alg=oalg, stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="COTANGENT", quad_abstol=1e-8, quad_reltol=1e-8)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(out, rout) < 1e-11 and rel(du0, rdu0) < 1e-7 and rel(dp, rdp) < 1e-7      # measured <= 4e-14 / 5e-9 / 8e-10


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lane_bodies_on_robertson_at_the_stiff_rates(gold, alg, oalg):
    """|d h J| ~ 1e2 .. 1e7 in the W solves: the pivoting LU of the lanes doing real work.  Against the oracle AND against the independent gradient."""
    c = gold["rober"]
    d = np.zeros((1, 2, 3)); d[:, :, 2] = 1.0
    cfg = E.make_config("emu_rober", alg, 1, 0.0, 100.0, 0.0, c["ts"], loss_kind=0, stepper=ROS, abstol=1e-8, reltol=1e-6, max_steps=20000, quad_abstol=1e-12, quad_reltol=1e-6)
    du0, dp, out = E.forward_adjoint(cfg, 3, 3, [c["u0"]], c["p"], d)
    pr = O.Problem("ROBER", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-8, reltol=1e-6, save_times=c["ts"], loss="COTANGENT", quad_abstol=1e-12, quad_reltol=1e-6)
    rdu0, rdp, rout = pr.adjoint(c["u0"], c["p"], d[0])
    assert np.max(np.abs(out[0] - rout)) < 1e-13
    assert relc(dp, rdp) < 1e-6 and relc(du0[0], rdu0) < 1e-6
    assert relc(dp, c["dp"]) < 1e-3 and relc(du0[0], c["du0"]) < 1e-3


def test_lane_bodies_least_squares_loss_and_the_reference_fit(gold):
    """The device-resident loss routes with this stepper: the fit of test/Core2/stiff_adjoints.jl as HIPADJ_LOSS_LSQ_DATA (scale 2 = sum(abs2, ...)), and dg = u - 2 (LSQ_SHIFT)."""
    c = gold["lv"]
    tgt = np.asarray(c["target"])[None]
    cfg = E.make_config("lv", "interpolating", 1, 0.0, 10.0, 0.0, c["ts"], loss_kind=2, loss_scale=2.0, stepper=ROS, abstol=1e-8, reltol=1e-8, max_steps=20000)
    du0, dp, out = E.forward_adjoint(cfg, 2, 4, [c["u0"]], c["p"], tgt)
    assert rel(dp, c["dp"]) < 1e-4 and rel(du0[0], c["du0"]) < 1e-4
    for alg, oalg in ALGS:
        cfg = E.make_config("lv", alg, 1, 0.0, 10.0, 0.0, c["ts"], loss_kind=1, loss_shift=2.0, stepper=ROS, abstol=1e-8, reltol=1e-8, max_steps=20000, quad_abstol=1e-8, quad_reltol=1e-8)
        du0, dp, _ = E.forward_adjoint(cfg, 2, 4, [c["u0"]], c["p"])
        pr = O.Problem("LV", alg=oalg, stepper="ROS23", t0=0.0, t1=10.0, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=c["ts"], loss="LSQ_SHIFT", loss_shift=2.0, quad_abstol=1e-8, quad_reltol=1e-8)
        rdu0, rdp, _ = pr.adjoint(c["u0"], c["p"])
        # ~3000 reverse steps with a rejection at most forward knots (the interpolant's kinks): one borderline accept / reject decided differently by an ulp moves the answer
        # by a fraction of the tolerance — measured 1e-7 on InterpolatingAdjoint, 1e-10 on the others
        assert rel(du0[0], rdu0) < 2e-6 and rel(dp, rdp) < 2e-6, alg


def test_step_capacity_overflow_is_an_error_and_what_the_planner_refuses():
    u0 = np.array([[1.0, 1.0]]); p = np.array([1.5, 1.0, 3.0, 1.0])
    cfg = E.make_config("lv", "interpolating", 1, 0.0, 10.0, 0.0, [10.0], loss_kind=1, stepper=ROS, abstol=1e-10, reltol=1e-10, max_steps=50)
    with pytest.raises(RuntimeError, match="rc=-7"):
        E.forward_adjoint(cfg, 2, 4, u0, p)
    # a semi-explicit DAE takes no continuous cost (its loss jump is built for the plain loss routes)
    cfg = E.make_config("emu_roberdae", "interpolating", 1, 0.0, 1.0, 0.0, [1.0], stepper=ROS, cont_cost=1)
    with pytest.raises(RuntimeError, match="rc=-6"):
        E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 0.0]], [0.04, 3e7, 1e4], np.zeros((1, 1, 3)))
    # BacksolveAdjoint on a semi-explicit DAE: refused by both (the reference documents it to fail there, test/Core3/adjoint.jl:1516-1530)
    cfg = E.make_config("emu_roberdae", "backsolve", 1, 0.0, 1.0, 0.0, [1.0], stepper=ROS, checkpointing=True)
    with pytest.raises(RuntimeError, match="rc=-6"):
        E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 0.0]], [0.04, 3e7, 1e4], np.zeros((1, 1, 3)))
    with O.mass_matrix(DAE_M):
        with pytest.raises(RuntimeError):
            O.Problem("ROBERDAE", alg="BACKSOLVE", stepper="ROS23", t0=0, t1=1.0, dt=0.0, save_times=[1.0], loss="COTANGENT", checkpointing=True).adjoint([1.0, 0.0, 0.0], [0.04, 3e7, 1e4], np.zeros((1, 3)))


def test_the_time_derivative_term_of_k3_is_not_pinned_by_the_reference_relation_but_by_the_step_count(gold):
    """ORC_RECALL_ROS_K3_T: with `h dT` instead of `d h dT` the gradient still meets test/Core2/stiff_adjoints.jl's bar — the relation cannot tell the two readings apart — and the
    reverse pass takes >= 8 x the right-hand sides (profiles/r6_rosenbrock23_steps.json: 10 - 270 x).  The restatement and the device hold `d h dT` (Shampine-Reichelt's form, third
    order in h on u' = g(t)); oracle/_ref/README.md lists it among the constants a Julia run would settle."""
    c = gold["lv"]
    L = O.lib()
    counts = {}
    try:
        for name, v in (("d", 0.29289321881345247560), ("one", 1.0)):
            assert L.orc_test_set_recall(10, C.c_double(v)) == 0
            pr = O.Problem("LV", alg="INTERPOLATING", stepper="ROS23", t0=0.0, t1=10.0, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=c["ts"], loss="LSQ_DATA", loss_scale=2.0)
            u0, p, data = O._arr(c["u0"]), O._arr(c["p"]), O._arr(np.asarray(c["target"]))
            du0, dp, out = np.zeros(2), np.zeros(4), np.zeros((len(c["ts"]), 2)); nr = C.c_long()
            assert L.orc_adjoint(C.byref(pr.cfg), O._p(u0), O._p(p), O._p(data), O._p(du0), O._p(dp), O._p(out), C.byref(nr)) == 0
            assert rel(dp, c["dp"]) < 1e-4, name
            counts[name] = nr.value
    finally:
        L.orc_test_set_recall(10, C.c_double(0.29289321881345247560))
    assert counts["one"] > 8 * counts["d"], counts


# ---- constant non-singular mass matrix (the reference tests it WITH a Rosenbrock method: Rodas4, test/Core3/adjoint.jl:1308-1376) ------------------------------------------
@pytest.mark.parametrize("alg,oalg", ALGS)
def test_oracle_mass_matrix_problem_against_its_closed_form(alg, oalg):
    """`foo` with the dense M of test/Core3/adjoint.jl:1315-1321, G = sum of the states at ts: closed form in tests/golden/mass_matrix.json.  W = I - d h M^-1 J here is
    M^-1 (M - d h J): the stages of the mass-matrix form of the method, computed through the M^-1 f rewrite the library uses for every stepper."""
    with open(os.path.join(HERE, "golden", "mass_matrix.json")) as f:
        G = json.load(f)
    M = np.array(G["M"]); ts = np.array(G["ts"])
    with O.mass_matrix(M):
        pr = O.Problem("AFFINE3", alg=oalg, stepper="ROS23", t0=0, t1=1.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, loss="COTANGENT", quad_abstol=1e-9, quad_reltol=1e-9)
        du0, dp, out = pr.adjoint(G["u0"], G["p"], np.ones((len(ts), 3)))
    assert rel(out[-1], G["u_end"]) < 2e-6 and rel(dp, G["dGdp"]) < 2e-6 and rel(du0, G["lam0"]) < 2e-6      # measured 2e-7 .. 5e-7


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lane_bodies_behind_a_mass_matrix(alg, oalg):
    """emu_ring5mm (F = M^-1 f, VJPs through M^-T: the wrapper hipadj_user.hpp generates; non-autonomous flag set, n = 5: the widest unrolled LU the emulator holds) against the
    oracle's lam formulation.  The two formulations weigh the error norm differently (nu = M' lam here), so they agree to the solver tolerance, not to roundoff."""
    rng = np.random.default_rng(5)
    n, npar, N, T = 5, 6, 3, 2.0
    u0 = rng.uniform(0.3, 1.0, (N, n)); ts = np.array([0.4, 1.1, 2.0]); delta = rng.standard_normal((N, 3, n))
    pp = rng.uniform(0.4, 1.2, (N, npar))
    M = np.linalg.inv(E.ring_mm_inverse(n))
    cfg = E.make_config("emu_ring5mm", alg, N, 0.0, T, 0.0, ts, loss_kind=0, p_shared=False, stepper=ROS, abstol=1e-9, reltol=1e-9, quad_abstol=1e-9, quad_reltol=1e-9, max_steps=50000)
    du0, dp, out = E.forward_adjoint(cfg, n, npar, u0, pp, delta)
    with O.mass_matrix(M):
        ref = O.Problem("RING", alg=oalg, t0=0.0, t1=T, save_times=ts, loss="COTANGENT", dims=(n, 0, 0, 0), stepper="ROS23", dt=0.0, abstol=1e-9, reltol=1e-9, quad_abstol=1e-9, quad_reltol=1e-9)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(out, rout) < 1e-10 and rel(du0, rdu0 @ M) < 1e-6 and rel(dp, rdp) < 1e-6      # measured 7e-15 / 2e-8 / 3e-8


# ---- singular mass matrix: the semi-explicit DAE of test/Core3/adjoint.jl:1434-1530 -------------------------------------------------------------------------------------
DAE_M = np.diag([1.0, 1.0, 0.0])


def dae_golden(gold):
    """`rober` with the conservation row as a constraint has the trajectory of the ODE form whenever sum(u0) = 1: the independent Radau gradient of the ODE fixture IS the DAE's
    dG/dp; dG/d(differential u0) follows from it with y3(0) = 1 - y1(0) - y2(0)."""
    c = gold["rober"]
    g = np.asarray(c["du0"])
    return c, np.asarray(c["dp"]), np.array([g[0] - g[2], g[1] - g[2]])


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("u0", [[1.0, 0.0, 0.0], [1.0, 0.0, 1.0]], ids=["consistent", "inconsistent"])
def test_oracle_semi_explicit_dae_against_the_independent_gradient(gold, alg, oalg, u0):
    """ODEFunction(rober, mass_matrix = diag(1, 1, 0)), p = [0.04, 3e7, 1e4], tspan (0, 100), ts = [50, 100], dg = e_3 (test/Core3/adjoint.jl:1434-1466): the reference asks
    every sensealg to agree with ForwardDiff to rtol 1e-5 (:1483, 1493, 1503, 1514).  u0 = [1, 0, 1] is the reference's own, inconsistent, start (:1460): BrownFullBasicInit
    moves y3 to 0.  du0 = lam(t0): its differential entries are dG/d(y1(0), y2(0)) along the constraint."""
    c, gdp, gdu = dae_golden(gold)
    d = np.zeros((2, 3)); d[:, 2] = 1.0
    with O.mass_matrix(DAE_M):
        pr = O.Problem("ROBERDAE", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-10, reltol=1e-8, save_times=c["ts"], loss="COTANGENT", quad_abstol=1e-12, quad_reltol=1e-8)
        du0, dp, out = pr.adjoint(u0, c["p"], d)
    assert relc(dp, gdp) < 1e-5 and relc(du0[:2], gdu) < 2e-4           # measured 1.8e-6 (3.7e-6 Quadrature) / 3e-5 (the small second entry)
    assert np.max(np.abs(out - np.asarray(c["u_at_ts"]))) < 1e-6 and np.max(np.abs(out.sum(axis=1) - 1.0)) < 1e-9      # the constraint holds along the solution


def test_oracle_mass_matrix_forms_that_are_refused():
    L = O.lib()
    for M in (np.array([[1.0, 1.0], [1.0, 1.0]]), np.array([[1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]), np.zeros((2, 2))):     # singular, not [Md 0; 0 0]
        Mc = np.ascontiguousarray(M)
        assert L.orc_set_mass_matrix(M.shape[0], O._p(Mc)) == -2
    L.orc_set_mass_matrix(0, None)
    with O.mass_matrix(DAE_M):      # a DAE needs the implicit stepper
        with pytest.raises(RuntimeError):
            O.Problem("ROBERDAE", alg="INTERPOLATING", stepper="TSIT5", t0=0.0, t1=1.0, dt=0.0, save_times=[1.0], loss="COTANGENT").adjoint([1.0, 0.0, 0.0], [0.04, 3e7, 1e4], np.zeros((1, 3)))


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lane_bodies_semi_explicit_dae(gold, alg, oalg):
    """The DAE lanes (mass-matrix form of the stages, consistent initialisation, the loss jumps of src/adjoint_common.jl:790-813 with one factorisation of the algebraic block
    serving the elimination and the re-initialisation) against the oracle, from the reference's inconsistent start, and against the independent gradient."""
    c, gdp, gdu = dae_golden(gold)
    d = np.zeros((1, 2, 3)); d[:, :, 2] = 1.0
    cfg = E.make_config("emu_roberdae", alg, 1, 0.0, 100.0, 0.0, c["ts"], loss_kind=0, stepper=ROS, abstol=1e-10, reltol=1e-8, max_steps=100000, quad_abstol=1e-12, quad_reltol=1e-8)
    du0, dp, out = E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 1.0]], c["p"], d)
    with O.mass_matrix(DAE_M):
        pr = O.Problem("ROBERDAE", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-10, reltol=1e-8, save_times=c["ts"], loss="COTANGENT", quad_abstol=1e-12, quad_reltol=1e-8)
        rdu0, rdp, rout = pr.adjoint([1.0, 0.0, 1.0], c["p"], d[0])
    assert np.max(np.abs(out[0] - rout)) < 1e-9 and relc(dp, rdp) < 1e-6 and np.max(np.abs(du0[0] - rdu0)) < 1e-8      # measured 5e-11 / 2e-9 / 2e-11
    assert relc(dp, gdp) < 1e-5


def test_lane_bodies_dae_model_needs_the_stiff_stepper():
    for st in (0, 1):
        cfg = E.make_config("emu_roberdae", "interpolating", 1, 0.0, 1.0, 0.01, [1.0], loss_kind=1, stepper=st)
        with pytest.raises(RuntimeError, match="singular"):
            E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 0.0]], [0.04, 3e7, 1e4])


# ---- checkpointing = true: the intervals re-solved with Rosenbrock23 inside the reverse pass (src/interpolating_adjoint.jl:54-109, 207-277) ---------------------------------
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD")])
@pytest.mark.parametrize("model,omodel,u0c,p", MODELS)
def test_lane_bodies_checkpointed(alg, oalg, model, omodel, u0c, p):
    """The forward pass keeps sol(c_j) only; every interval between checkpoints (default: the loss times and the span's ends) is re-solved by the forward stepper — Rosenbrock23 —
    when the reverse solve enters it, from dt = the last step of the interval above.  The re-solved pieces are NOT the forward solution (other step sequences), so the gradient
    moves at the solver tolerance against the dense run and agrees with the oracle's checkpointed run at the level two implementations of one controller do."""
    rng = np.random.default_rng(15)
    N, T = 3, 2.0
    n, npar = len(u0c), len(p)
    u0 = np.asarray(u0c) + 0.05 * rng.standard_normal((N, n))
    pp = np.asarray(p) * (1 + 0.03 * rng.standard_normal((N, npar)))
    ts = np.array([0.0, 0.13, 0.5, 0.77, 1.0, 1.9, 2.0])
    delta = rng.standard_normal((N, len(ts), n))
    cfg = E.make_config(model, alg, N, 0.0, T, 0.0, ts, loss_kind=0, p_shared=False, stepper=ROS, abstol=1e-8, reltol=1e-8, checkpointing=True, max_steps=100000)
    du0, dp, out = E.forward_adjoint(cfg, n, npar, u0, pp, delta)
    ref = O.Problem(omodel, alg=oalg, stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="COTANGENT", checkpointing=True)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(out, rout) < 1e-11 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6
    dense = O.Problem(omodel, alg=oalg, stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="COTANGENT")
    ddu0, ddp, _, _ = dense.adjoint_ensemble(u0, pp, delta)
    assert rel(du0, ddu0) < 1e-4 and rel(dp, ddp) < 1e-4


@pytest.mark.parametrize("ckpts", [None, "every_10"])
def test_semi_explicit_dae_checkpointed(gold, ckpts):
    """test/Core3/adjoint.jl:1505-1514: InterpolatingAdjoint(checkpointing = true), checkpoints = sol.t[1:10:end] on the singular-mass-matrix problem (here: the default list and a
    list of ten points of the span): lanes = oracle, and the gradient still meets the reference's bar against the independent one."""
    c, gdp, gdu = dae_golden(gold)
    d = np.zeros((1, 2, 3)); d[:, :, 2] = 1.0
    ck = None if ckpts is None else np.linspace(0.0, 100.0, 11)
    cfg = E.make_config("emu_roberdae", "interpolating", 1, 0.0, 100.0, 0.0, c["ts"], loss_kind=0, stepper=ROS, abstol=1e-10, reltol=1e-8, max_steps=100000, checkpointing=True, checkpoints=ck)
    du0, dp, out = E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 1.0]], c["p"], d)
    with O.mass_matrix(DAE_M):
        pr = O.Problem("ROBERDAE", alg="INTERPOLATING", stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-10, reltol=1e-8, save_times=c["ts"], loss="COTANGENT", checkpointing=True, checkpoints=ck)
        rdu0, rdp, rout = pr.adjoint([1.0, 0.0, 1.0], c["p"], d[0])
    assert relc(dp, rdp) < 1e-6 and np.max(np.abs(du0[0] - rdu0)) < 1e-7
    assert relc(dp, gdp) < 1e-5


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_dae_with_a_parameter_dependent_constraint(gold, alg, oalg):
    """y1 + y2 + y3 = 1 + 5 (p1 - 0.04): the loss jumps' parameter term f_p' [0; dlam_a] (src/adjoint_common.jl:803; added to dp by src/sensitivity_interface.jl:510-521 and its
    Quadrature / Gauss twins) is 3.0 of dG/dp1 = 12.6 here and identically zero on the reference's own constraint — this is the case that sees it.  Independent gradient: the
    reduced ODE's Radau sensitivities (tests/golden/make_stiff_adjoints.py rober_dae_kappa).  Oracle and lanes, every sensealg (QuadratureAdjoint: the term rides in the
    quadrature's start value, k_quad_sum with add = 1)."""
    c = gold["rober_dae_kappa"]
    d = np.zeros((1, 2, 3)); d[:, :, 2] = 1.0
    with O.mass_matrix(DAE_M):
        pr = O.Problem("ROBERDAE", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-10, reltol=1e-8, save_times=c["ts"], loss="COTANGENT", quad_abstol=1e-12, quad_reltol=1e-8, dims=(5, 0, 0, 0))
        rdu0, rdp, rout = pr.adjoint([1.0, 0.0, 1.0], c["p"], d[0])
    assert relc(rdp, c["dp"]) < 1e-5 and relc(rdu0[:2], c["du0_differential"]) < 2e-4
    assert abs(rdp[0] - gold["rober"]["dp"][0]) > 3.0          # (the term is there: the plain constraint's dG/dp1 is 9.59)
    cfg = E.make_config("emu_roberdae_kappa", alg, 1, 0.0, 100.0, 0.0, c["ts"], loss_kind=0, stepper=ROS, abstol=1e-10, reltol=1e-8, max_steps=100000, quad_abstol=1e-12, quad_reltol=1e-8)
    du0, dp, out = E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 1.0]], c["p"], d)
    bar = 1e-4 if alg == "quadrature" else 1e-6      # quadgk's first panel on a stiff problem moves the answer by 3e-6; the term under test is 24 % of dp[0]
    assert relc(dp, rdp) < bar and np.max(np.abs(du0[0] - rdu0)) < 1e-8 and relc(dp, c["dp"]) < 10 * bar


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD")])
def test_dae_with_a_dense_differential_mass_block(gold, alg, oalg):
    """M = [Md 0; 0 0] with Md = [2 0.3; 0.1 0.5] and the differential rows of `rober` mixed by the same Md: the trajectory and dG/dp of the parameter-dependent case, with
    off-diagonal mass entries in W = M - d h J, in M k, and a real lu(M'[diff, diff]) in the loss jumps; dG/d(differential u0) = Md' lam_d(t0) (du0 is the reference's lam(t0))."""
    c = gold["rober_dae_kappa"]
    Md = np.array([[2.0, 0.3], [0.1, 0.5]]); M = np.zeros((3, 3)); M[:2, :2] = Md
    d = np.zeros((1, 2, 3)); d[:, :, 2] = 1.0
    with O.mass_matrix(M):
        pr = O.Problem("ROBERDAE", alg=oalg, stepper="ROS23", t0=0.0, t1=100.0, dt=0.0, abstol=1e-10, reltol=1e-8, save_times=c["ts"], loss="COTANGENT", dims=(5, 1, 0, 0))
        rdu0, rdp, rout = pr.adjoint([1.0, 0.0, 1.0], c["p"], d[0])
    assert relc(rdp, c["dp"]) < 1e-5 and relc(Md.T @ rdu0[:2], c["du0_differential"]) < 2e-4
    cfg = E.make_config("emu_roberdae_mix", alg, 1, 0.0, 100.0, 0.0, c["ts"], loss_kind=0, stepper=ROS, abstol=1e-10, reltol=1e-8, max_steps=100000)
    du0, dp, out = E.forward_adjoint(cfg, 3, 3, [[1.0, 0.0, 1.0]], c["p"], d)
    assert relc(dp, rdp) < 1e-6 and np.max(np.abs(du0[0] - rdu0)) < 1e-7 and np.max(np.abs(out[0] - rout)) < 1e-9


# ---- BacksolveAdjoint on the stiff stepper (test/Core2/stiff_adjoints.jl:207-222 runs it with every implicit solver on u' = u .* p) ----------------------------------------
@pytest.mark.parametrize("ckpt", [True, False])
@pytest.mark.parametrize("model,omodel,u0c,p", [MODELS[0], MODELS[1], ("lindiag", "LINDIAG", [3.0, 2.0], [0.6, 0.4])])
def test_lane_bodies_backsolve(model, omodel, u0c, p, ckpt):
    """z = [lam; mu; y] is not affine in y; Rosenbrock23 is a W-method and W is formed from the first-derivative blocks only (two n x n factorisations and a substitution per
    step; the reference's W carries the second-derivative blocks too: a deliberate deviation that moves the step sequence, not the order) — lanes against the oracle's same W, and
    against InterpolatingAdjoint at the level BacksolveAdjoint reaches (the reference's own bar for this pair is rtol 1e-2, :221-222)."""
    rng = np.random.default_rng(16)
    N, T = 3, 1.0
    n, npar = len(u0c), len(p)
    u0 = np.asarray(u0c) + 0.05 * rng.standard_normal((N, n))
    pp = np.asarray(p) * (1 + 0.03 * rng.standard_normal((N, npar)))
    ts = np.array([0.0, 0.1, 0.33, 0.5, 0.77, 1.0])
    delta = rng.standard_normal((N, len(ts), n))
    cfg = E.make_config(model, "backsolve", N, 0.0, T, 0.0, ts, loss_kind=0, p_shared=False, stepper=ROS, abstol=1e-8, reltol=1e-8, checkpointing=ckpt, max_steps=100000)
    du0, dp, out = E.forward_adjoint(cfg, n, npar, u0, pp, delta)
    ref = O.Problem(omodel, alg="BACKSOLVE", stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="COTANGENT", checkpointing=ckpt)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(out, rout) < 1e-11 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6
    ia = O.Problem(omodel, alg="INTERPOLATING", stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="COTANGENT")
    idu0, idp, _, _ = ia.adjoint_ensemble(u0, pp, delta)
    assert rel(du0, idu0) < 1e-3 and rel(dp, idp) < 1e-3


# ---- continuous costs on the stiff stepper (accumulate_cost!, src/derivative_wrappers.jl:1411-1442; the costs of test/Core3/adjoint.jl:913-919 and test/Core7/mixed_costs.jl:46-57) ------
@pytest.mark.parametrize("cost", [1, 2])
@pytest.mark.parametrize("alg,oalg,ck", [("interpolating", "INTERPOLATING", False), ("interpolating", "INTERPOLATING", True), ("backsolve", "BACKSOLVE", True), ("gauss", "GAUSS", False),
                                         ("gausskronrod", "GAUSS_KRONROD", True), ("quadrature", "QUADRATURE", False)])
def test_lane_bodies_continuous_costs(alg, oalg, ck, cost):
    """g = (sum u)^2 / 2 and g = u1^2 + p1 (with its dgdp_continuous) added to the discrete loss: the reverse right-hand side becomes affine (W unchanged, the cost's time
    dependence enters through dT), Gauss / GaussKronrod / Quadrature add g_p to their integrands.  Lanes against the oracle on Lotka-Volterra."""
    if alg == "gauss" and cost == 2:
        pytest.skip("GaussAdjoint with dgdp_continuous: the sign convention case of DESIGN 6.5 — covered on Tsit5 with its own test")
    rng = np.random.default_rng(17)
    N, T = 2, 1.5
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0])
    ts = np.array([0.0, 0.4, 1.0, 1.5])
    cfg = E.make_config("lv", alg, N, 0.0, T, 0.0, ts, loss_kind=1, loss_shift=2.0, stepper=ROS, abstol=1e-8, reltol=1e-8, quad_abstol=1e-9, quad_reltol=1e-9, cont_cost=cost, checkpointing=ck, max_steps=100000)
    du0, dp, out = E.forward_adjoint(cfg, 2, 4, u0, p)
    ref = O.Problem("LV", alg=oalg, stepper="ROS23", t0=0, t1=T, dt=0.0, abstol=1e-8, reltol=1e-8, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, quad_abstol=1e-9, quad_reltol=1e-9,
                    cont_cost=cost, checkpointing=ck)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, None)
    assert rel(du0, rdu0) < 2e-6 and rel(dp, rdp) < 2e-6


@pytest.mark.parametrize("alg", ["interpolating", "gauss", "quadrature", "backsolve"])
def test_lane_bodies_initial_dt_hint(alg):
    """`dt` with an adaptive stepper is the first step's length (test/Core2/stiff_adjoints.jl:203 passes dt = 0.01 to every implicit solver), for the forward AND the reverse solve."""
    u0 = np.array([[1.0, 1.0]]); p = np.array([1.5, 1.0, 3.0, 1.0]); ts = [0.0, 0.4, 1.0]
    ck = alg == "backsolve"
    cfg = E.make_config("lv", alg, 1, 0.0, 1.0, 0.05, ts, loss_kind=1, loss_shift=2.0, stepper=ROS, abstol=1e-9, reltol=1e-9, quad_abstol=1e-10, quad_reltol=1e-10, checkpointing=ck, max_steps=100000)
    du0, dp, _ = E.forward_adjoint(cfg, 2, 4, u0, p)
    ref = O.Problem("LV", alg=alg.upper(), stepper="ROS23", t0=0, t1=1.0, dt=0.05, abstol=1e-9, reltol=1e-9, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, quad_abstol=1e-10, quad_reltol=1e-10, checkpointing=ck)
    rdu0, rdp, _ = ref.adjoint(u0[0], p)
    assert rel(du0[0], rdu0) < 1e-10 and rel(dp, rdp) < 1e-10      # measured 6e-14: the same step sequence
    nohint = O.Problem("LV", alg=alg.upper(), stepper="ROS23", t0=0, t1=1.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, quad_abstol=1e-10, quad_reltol=1e-10, checkpointing=ck)
    assert rel(nohint.adjoint(u0[0], p)[1], rdp) > 1e-13          # (the hint is used: another step sequence)
