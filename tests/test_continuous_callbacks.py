"""ContinuousCallback without a GPU: the oracle's restatement (oracle/adjoint_oracle.c section 3b) against CLOSED-FORM gradients, the device lane bodies (compiled for the
host, tests/emu) against the oracle, the registration entry point and the planner's refusals, and the runtime kernels through hiprtc.

Reference: src/callback_tracking.jl:1-223 (forward tracking), :232-479 (reverse callbacks, the implicit correction for the event time), test/Callbacks2/continuous_callbacks.jl
(the bouncing ball: u0 = [5, 0], tspan (0, 2.5), p = [9.8, 0.8], saveat 0.5, abstol = reltol = 1e-12; its bar: rtol 1e-5 against ForwardDiff, :140-145).
The closed forms (tests/golden/make_continuous_callbacks.py) are exact: piecewise ballistic flights, event times as roots of quadratics, differentiated by dual numbers."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

import emu as E
import oracle as O
import user_models as UM

HERE = os.path.dirname(os.path.abspath(__file__))
ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD"), ("quadrature", "QUADRATURE")]
QTOL = dict(quad_abstol=1e-14, quad_reltol=1e-12)
CASES = {"ball": (1, "FALLMASS", "emu_ball"), "ball_long": (1, "FALLMASS", "emu_ball"), "ball_mse": (2, "FALLMASS", None), "relax": (3, "RELAX", "emu_relax"),
         "moving": (4, "FALLMASS", "emu_ball_moving")}
TS5, ROS = 1, 3


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "continuous_callbacks.json")) as f:
        return json.load(f)


def relc(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12 * np.max(np.abs(b)))))


def relmax(du0, dp, g):
    """error of the whole gradient against its largest entry (an entry that is 1e-5 of the others is not asked to 1e-5 of itself from a second-order stepper)"""
    a = np.concatenate([np.ravel(du0), np.ravel(dp)]); b = np.concatenate([np.ravel(g["du0"]), np.ravel(g["dp"])])
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def oracle_run(g, omodel, kind, oalg, stepper="TSIT5", tol=1e-12, mse=False):
    ts = np.asarray(g["ts"])
    pr = O.Problem(omodel, alg=oalg, stepper=stepper, t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=tol, reltol=tol, save_times=ts, event_kind=kind,
                   loss="LSQ_SHIFT" if mse else "COTANGENT", loss_shift=1.0 if mse else 0.0, **QTOL)
    return pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), None if mse else np.ones((len(ts), len(g["u0"]))))


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_against_the_closed_forms(gold, case, alg, oalg):
    """the pin of the restatement: every term of the event jump — a_u' lam, the event-time term with c_u, c_t, a_t, and the parameter terms a_p' lam and kappa c_p — is needed by
    at least one of these numbers (relax: -kappa c_p = 2.7e-4 of a gradient entry asked to 1e-10 by the reference, :342-346)"""
    kind, omodel, _ = CASES[case]; g = gold[case]
    du0, dp, out = oracle_run(g, omodel, kind, oalg, mse=(case == "ball_mse"))
    bar = 2e-9 if case == "relax" else 1e-11
    assert relc(du0, g["du0"]) < bar and relc(dp, g["dp"]) < bar
    if "u_at_ts" in g:
        assert np.max(np.abs(out - np.asarray(g["u_at_ts"]))) < 1e-11


def test_the_reference_records_the_relax_gradient(gold):
    """test/Callbacks2/continuous_callbacks.jl:342 holds the numbers of its own finite-difference run as a comment"""
    assert np.allclose(gold["relax"]["dp"], [0.9999546000702386, 0.00018159971904994378], rtol=1e-9, atol=0)


@pytest.mark.parametrize("case", ["ball", "relax"])
def test_oracle_rosenbrock23(gold, case):
    kind, omodel, _ = CASES[case]; g = gold[case]
    du0, dp, _ = oracle_run(g, omodel, kind, "INTERPOLATING", stepper="ROS23", tol=1e-9)
    assert relmax(du0, dp, g) < 1e-5


def test_oracle_without_crossings_is_the_plain_solve():
    ts = np.array([0.5, 1.0, 2.0]); u0 = np.array([50.0, 3.0]); p = np.array([9.8, 0.8]); d = np.ones((3, 2))
    kw = dict(alg="INTERPOLATING", stepper="TSIT5", t0=0.0, t1=2.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts)
    a = O.Problem("FALLMASS", **kw).adjoint(u0, p, d); b = O.Problem("FALLMASS", event_kind=1, **kw).adjoint(u0, p, d)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_oracle_refusals_and_the_ball_that_comes_to_rest():
    ts = np.array([1.0, 2.0]); u0 = np.array([5.0, 0.0]); p = np.array([9.8, 0.8]); d = np.ones((2, 2))
    kw = dict(t0=0.0, t1=2.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, event_kind=1)
    for bad in (dict(alg="INTERPOLATING", stepper="TSIT5", cont_cost=1), dict(alg="GAUSS", stepper="TSIT5", cont_cost=2)):
        with pytest.raises(RuntimeError, match="rc=-6"):
            O.Problem("FALLMASS", **{**kw, **bad}).adjoint(u0, p, d)
    with pytest.raises(RuntimeError, match="rc=-6"):
        O.Problem("FALLMASS", alg="INTERPOLATING", stepper="RK4", **{**kw, "dt": 0.01}).adjoint(u0, p, d)
    with pytest.raises(RuntimeError, match="rc=-6"):          # event 3 is written for the one-state model
        O.Problem("FALLMASS", alg="INTERPOLATING", stepper="TSIT5", **{**kw, "event_kind": 3}).adjoint(u0, p, d)
    # restitution 0.5 from height 1: the bounces accumulate at t = 1.355 < 4.  The solve must END: with flights below the time resolution the ball passes through the floor (as
    # it does in the reference's solver) or the event bound reports -7; either way not an endless loop
    try:
        O.Problem("FALLMASS", alg="INTERPOLATING", stepper="TSIT5", **{**kw, "t1": 4.0, "save_times": np.array([1.0, 4.0])}).adjoint(np.array([1.0, 0.0]), np.array([9.8, 0.5]), d)
    except RuntimeError as e:
        assert "rc=-7" in str(e)


@pytest.mark.parametrize("case", ["ball", "ball_long", "relax"])
def test_backsolve_with_checkpoints_through_events(gold, case):
    """BacksolveAdjoint(checkpointing = true) — the reference's default and the algorithm its callback tests lean on (test/Callbacks2/continuous_callbacks.jl:46-57): the backsolved
    state is overwritten at the checkpoints AND at the events (with the stored left state); oracle and lane bodies, default checkpoints (the save times) and a list"""
    kind, omodel, emodel = CASES[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    for ck in (None, [0.2, 0.9, 1.1, 2.4]):
        pr = O.Problem(omodel, alg="BACKSOLVE", stepper="TSIT5", t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=kind, checkpointing=True, checkpoints=ck)
        rdu0, rdp, _ = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), np.ones((len(ts), n)))
        assert relmax(rdu0, rdp, g) < 1e-11
        cfg = E.make_config(emodel, "backsolve", 1, g["tspan"][0], g["tspan"][1], 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, checkpointing=True, checkpoints=ck)
        du0, dp, _ = E.forward_adjoint(cfg, n, len(g["p"]), [g["u0"]], g["p"], np.ones((1, len(ts), n)))
        assert relmax(du0[0], dp, g) < 1e-11 and relc(du0[0], rdu0) < 1e-10 and relc(dp, rdp) < 1e-10


@pytest.mark.parametrize("alg,oalg", [ALGS[0], ALGS[2], ALGS[3]])
@pytest.mark.parametrize("case", ["ball", "ball_long", "relax", "moving", "ball_terminate"])
def test_checkpointed_interpolating_and_gauss_through_events(gold, case, alg, oalg):
    """InterpolatingAdjoint(checkpointing = true) — du03 of the reference's callback tests (test/Callbacks2/continuous_callbacks.jl:99-110) — and the checkpointed Gauss sweeps: a
    checkpoint interval is re-solved only as far as the current piece reaches, from the state just after the piece's lower event; default checkpoints (the save times) and a
    list; oracle and lane bodies against the closed forms"""
    kind, omodel, emodel = {**CASES, "ball_terminate": (7, "FALLMASS", "emu_ball_terminate")}[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    for ck in (None, [0.2, 0.9, 1.1, 2.4]):
        pr = O.Problem(omodel, alg=oalg, stepper="TSIT5", t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=kind, checkpointing=True, checkpoints=ck)
        rdu0, rdp, _ = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), np.ones((len(ts), n)))
        assert relmax(rdu0, rdp, g) < (2e-9 if case == "relax" else 1e-11)
        cfg = E.make_config(emodel, alg, 1, g["tspan"][0], g["tspan"][1], 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, checkpointing=True, checkpoints=ck)
        du0, dp, _ = E.forward_adjoint(cfg, n, len(g["p"]), [g["u0"]], g["p"], np.ones((1, len(ts), n)))
        assert relmax(du0[0], dp, g) < (2e-9 if case == "relax" else 1e-11) and relc(du0[0], rdu0) < 1e-9 and relc(dp, rdp) < 1e-9


# ---- the device lane bodies on the host ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", ["ball", "ball_long", "relax", "moving"])
def test_lane_bodies_against_the_oracle_and_the_closed_forms(gold, case, alg, oalg):
    kind, omodel, emodel = CASES[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    cfg = E.make_config(emodel, alg, 1, g["tspan"][0], g["tspan"][1], 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, **QTOL)
    du0, dp, out = E.forward_adjoint(cfg, n, len(g["p"]), [g["u0"]], g["p"], np.ones((1, len(ts), n)))
    rdu0, rdp, rout = oracle_run(g, omodel, kind, oalg)
    # two representations of one dense output (monomial record / stage form) locate the event to ~1 ulp of each other
    eo = 1e-10 if case == "relax" else 1e-11          # (relax: an exponential, solved — and by Backsolve re-solved backward — at 1e-12 by two controllers an ulp apart)
    assert relc(du0[0], rdu0) < eo and relc(dp, rdp) < eo and np.max(np.abs(out[0] - rout)) < 1e-11
    bar = 2e-9 if case == "relax" else 1e-11
    assert relc(du0[0], g["du0"]) < bar and relc(dp, g["dp"]) < bar


@pytest.mark.parametrize("alg,oalg", [ALGS[0], ALGS[1], ALGS[2], ALGS[4]])
@pytest.mark.parametrize("case", ["ball", "relax"])
def test_lane_bodies_rosenbrock23(gold, case, alg, oalg):
    kind, omodel, emodel = CASES[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    cfg = E.make_config(emodel, alg, 1, g["tspan"][0], g["tspan"][1], 0.0, ts, stepper=ROS, abstol=1e-9, reltol=1e-9, max_steps=20000, **QTOL)
    du0, dp, out = E.forward_adjoint(cfg, n, len(g["p"]), [g["u0"]], g["p"], np.ones((1, len(ts), n)))
    rdu0, rdp, rout = oracle_run(g, omodel, kind, oalg, stepper="ROS23", tol=1e-9)
    assert relc(du0[0], rdu0) < 1e-9 and relc(dp, rdp) < 1e-9 and np.max(np.abs(out[0] - rout)) < 1e-9
    assert relmax(du0[0], dp, g) < 1e-5


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lane_bodies_ensemble_with_per_trajectory_events(alg, oalg):
    """every trajectory its own event times and its own number of events (2 .. 7); per-trajectory parameters, random cotangents, loss times off any grid"""
    rng = np.random.default_rng(5)
    N, T = 40, 4.0
    u0 = np.stack([rng.uniform(2.0, 9.0, N), rng.uniform(-1.0, 1.0, N)], axis=1)
    p = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    ts = np.array([0.3, 1.0, 1.7, 2.2, 3.1, 4.0]); d = rng.standard_normal((N, len(ts), 2))
    cfg = E.make_config("emu_ball", alg, N, 0.0, T, 0.0, ts, stepper=TS5, abstol=1e-10, reltol=1e-10, max_steps=4000, p_shared=False, **QTOL)
    du0, dp, out = E.forward_adjoint(cfg, 2, 2, u0, p, d)
    ref = O.Problem("FALLMASS", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, event_kind=1, **QTOL)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, d)
    assert np.max(np.abs(out - rout)) < 1e-9
    sc = np.maximum(np.abs(rdu0), 1e-3 * np.abs(rdu0).max(axis=1, keepdims=True)); scp = np.maximum(np.abs(rdp), 1e-3 * np.abs(rdp).max(axis=1, keepdims=True))
    assert np.max(np.abs(du0 - rdu0) / sc) < 1e-8 and np.max(np.abs(dp - rdp) / scp) < 1e-8


def test_lane_bodies_event_list_overflow_and_refusals():
    ts = np.array([1.0, 4.0]); d = np.ones((1, 2, 2))
    cfg = E.make_config("emu_ball", "interpolating", 1, 0.0, 4.0, 0.0, ts, stepper=TS5, abstol=1e-9, reltol=1e-9, max_steps=4000)
    with pytest.raises(RuntimeError, match="rc=-7"):          # dropped from 0.1 with restitution 0.95: 24 bounces before t = 4, more than the list holds (16 in the emulator)
        E.forward_adjoint(cfg, 2, 2, [[0.1, 0.0]], [9.8, 0.95], d)
    for bad in (dict(alg="interpolating", cont_cost=1), dict(alg="interpolating", stepper=0, dt=0.01)):
        kw = dict(alg="interpolating", stepper=TS5, dt=0.0, checkpointing=False, cont_cost=0); kw.update(bad)
        cfg = E.make_config("emu_ball", kw["alg"], 1, 0.0, 4.0, kw["dt"], ts, stepper=kw["stepper"], abstol=1e-9, reltol=1e-9, checkpointing=kw["checkpointing"], cont_cost=kw["cont_cost"])
        with pytest.raises(RuntimeError, match="rc=-6"):
            E.forward_adjoint(cfg, 2, 2, [[5.0, 0.0]], [9.8, 0.8], d)


# ---- save_positions = (true, true): a loss on the saved event states ---------------------------------------------------------------------------------------------------
SAVED = {"ball_saved": (1, "emu_ball"), "ball_long_saved": (1, "emu_ball"), "ball_mse_saved": (2, None), "moving_saved": (4, "emu_ball_moving")}


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", sorted(SAVED))
def test_loss_on_the_saved_event_states(gold, case, alg, oalg):
    """the constructor's default and the setting of most of the reference's testsets (test/Callbacks2/continuous_callbacks.jl:200-217, 239-250): g also takes the state just before
    and just after every affect.  Event states and gradients against the closed forms: oracle (every sensealg), lane bodies (where the emulator has the model)"""
    kind, emodel = SAVED[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = 2; mse = "mse" in case
    es = np.asarray(g["event_states"]); ne = len(es)
    pr = O.Problem("FALLMASS", alg=oalg, stepper="TSIT5", t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=kind,
                   loss="LSQ_SHIFT" if mse else "COTANGENT", loss_shift=1.0 if mse else 0.0, **QTOL)
    t, ul, ur = pr.event_states(np.asarray(g["u0"]), np.asarray(g["p"]))
    assert len(t) == ne and np.max(np.abs(t - np.asarray(g["event_times"]))) < 1e-12 and np.max(np.abs(ul - es[:, 0])) < 1e-10 and np.max(np.abs(ur - es[:, 1])) < 1e-10
    dl, dr = ((ul - 1.0), (ur - 1.0)) if mse else (np.ones((ne, n)), np.ones((ne, n)))
    pr.set_event_cotangents(dl, dr)
    du0, dp, _ = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), None if mse else np.ones((len(ts), n)))
    assert relmax(du0, dp, g) < 1e-11
    assert relmax(du0, dp, gold[case[:-6]]) > 1e-3          # (the saved states do carry weight: the gradient without them is another one)
    if emodel is None:
        return
    cfg = E.make_config(emodel, alg, 1, g["tspan"][0], g["tspan"][1], 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, **QTOL)
    evo = E.set_event_output(1, n)
    pad = lambda a: np.concatenate([a, np.zeros((E.EMU_MAXEV - ne, n))])[None]
    keep = E.set_event_cotangents(pad(dl), pad(dr))
    try:
        edu0, edp, _ = E.forward_adjoint(cfg, n, 2, [g["u0"]], g["p"], np.ones((1, len(ts), n)))
    finally:
        E.set_event_cotangents(None, None); E.set_event_output(None)
    del keep
    assert np.max(np.abs(evo[0, :ne, 0] - t)) < 1e-12 and np.max(np.abs(evo[0, :ne, 1:1 + n] - ul)) < 1e-10 and np.max(np.abs(evo[0, :ne, 1 + n:] - ur)) < 1e-10 and np.all(evo[0, ne:] == 0.0)
    assert relmax(edu0[0], edp, g) < 1e-11 and relc(edu0[0], du0) < 1e-10 and relc(edp, dp) < 1e-10


# ---- VectorContinuousCallback ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case,kind", [("walls", 5), ("walls_saved", 5), ("clock", 6), ("clock_saved", 6)])
def test_vector_callback_against_the_closed_forms(gold, case, kind, alg, oalg):
    """test/Callbacks2/vector_continuous_callbacks.jl: a vector of conditions, the affect sees which one fired — the ball between two walls (:80-96: components fire in the order
    0, 1, 0) and conditions that depend on time only with an affect whose Jacobian is zero (:100-116: six events, alternating components); MSE loss, with and without the saved
    event states (the testsets use the default save_positions = (true, true)).  Oracle (every sensealg) and — the walls — the lane bodies"""
    g = gold[case]; ts = np.asarray(g["ts"]); n = 4; saved = case.endswith("_saved")
    pr = O.Problem("BALL2D", alg=oalg, stepper="TSIT5", t0=0.0, t1=10.0, dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=kind, loss="LSQ_SHIFT", loss_shift=1.0, **QTOL)
    t, ul, ur = pr.event_states(np.asarray(g["u0"]), np.asarray(g["p"]))
    assert len(t) == len(g["event_times"]) and np.max(np.abs(t - np.asarray(g["event_times"]))) < 1e-11
    if saved:
        es = np.asarray(g["event_states"])
        assert np.max(np.abs(ul - es[:, 0])) < 1e-9 and np.max(np.abs(ur - es[:, 1])) < 1e-9
        pr.set_event_cotangents(ul - 1.0, ur - 1.0)
    du0, dp, out = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), None)
    assert relmax(du0, dp, g) < 1e-11 and np.max(np.abs(out - np.asarray(g["u_at_ts"]))) < 1e-10
    if kind != 5:
        return
    ne = len(t)
    cfg = E.make_config("emu_ball2d", alg, 1, 0.0, 10.0, 0.0, ts, loss_kind=1, loss_shift=1.0, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, **QTOL)
    pad = lambda a: np.concatenate([a, np.zeros((E.EMU_MAXEV - ne, n))])[None]
    keep = E.set_event_cotangents(pad(ul - 1.0), pad(ur - 1.0)) if saved else None
    try:
        edu0, edp, eout = E.forward_adjoint(cfg, n, 2, [g["u0"]], g["p"])
    finally:
        E.set_event_cotangents(None, None)
    del keep
    assert relmax(edu0[0], edp, g) < 1e-11 and relc(edu0[0], du0) < 1e-10 and relc(edp, dp) < 1e-10 and np.max(np.abs(eout[0] - out)) < 1e-10


def test_vector_callback_registration_and_kernels(tmp_path, monkeypatch):
    """hipadj_model_set_vector_continuous_callback: `out[k]` in the condition body, `idx` in the affect body; ncond outside 1 .. 8 refused; the kernels through hiprtc"""
    from scimlsensitivity_jl_amd import _lib
    for kind in (5, 6):
        m, nc, cond, aff = UM.VECTOR_EVENTS[kind]
        mid = _lib.register_model(f"cc_vec_lint_{kind}", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
        with pytest.raises(_lib.HipadjError) as ei:
            _lib.set_model_continuous_callback(mid, cond, aff, 0, 9)
        assert ei.value.status == _lib.ERR_INVALID_ARG
        _lib.set_model_continuous_callback(mid, cond, aff, 8, nc)
        L = _lib.load()
        for alg, stepper in (("interpolating", TS5), ("backsolve", ROS), ("quadrature", TS5)):
            cfg = E.make_config(f"cc_vec_lint_{kind}", alg, 53, 0.0, 10.0, 0.0, [0.5, 5.0, 10.0], loss_kind=1, loss_shift=1.0, stepper=stepper, abstol=1e-8, reltol=1e-8)
            assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)


# ---- terminate! --------------------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", ["ball_terminate", "ball_terminate_saved"])
def test_terminate_against_the_closed_forms(gold, case, alg, oalg):
    """terminate!(integrator) in the affect (test/Callbacks2/continuous_callbacks.jl:226-236, "= callback with terminate"): the trajectory's solve ends at the bounce; the save times
    after it hold the final state and carry no loss; the reverse solve starts at the event with lam+ = 0 (the loss on the final state arrives as the event's right cotangent).
    Oracle and lane bodies, every sensealg, against the closed forms"""
    g = gold[case]; ts = np.asarray(g["ts"]); n = 2; saved = case.endswith("_saved"); es = np.asarray(g["event_states"]); nb = len(g["u_at_ts"])
    pr = O.Problem("FALLMASS", alg=oalg, stepper="TSIT5", t0=0.0, t1=2.5, dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=7, **QTOL)
    t, ul, ur = pr.event_states(np.asarray(g["u0"]), np.asarray(g["p"]))
    assert len(t) == 1 and abs(t[0] - g["event_times"][0]) < 1e-12 and np.max(np.abs(ul - es[:, 0])) < 1e-10 and np.max(np.abs(ur - es[:, 1])) < 1e-10
    if saved:
        pr.set_event_cotangents(np.ones((1, n)), np.ones((1, n)))
    du0, dp, out = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), np.ones((len(ts), n)))
    assert relmax(du0, dp, g) < 1e-12 and np.max(np.abs(out[:nb] - np.asarray(g["u_at_ts"]))) < 1e-11 and np.max(np.abs(out[nb:] - es[0, 1])) < 1e-10
    cfg = E.make_config("emu_ball_terminate", alg, 1, 0.0, 2.5, 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, **QTOL)
    pad = lambda a: np.concatenate([a, np.zeros((E.EMU_MAXEV - 1, n))])[None]
    keep = E.set_event_cotangents(pad(np.ones((1, n))), pad(np.ones((1, n)))) if saved else None
    try:
        edu0, edp, eout = E.forward_adjoint(cfg, n, 2, [g["u0"]], g["p"], np.ones((1, len(ts), n)))
    finally:
        E.set_event_cotangents(None, None)
    del keep
    assert relmax(edu0[0], edp, g) < 1e-12 and np.max(np.abs(eout[0] - out)) < 1e-11


def test_terminate_with_checkpointed_backsolve(gold):
    g = gold["ball_terminate"]; ts = np.asarray(g["ts"]); n = 2
    for ck in (None, [0.3, 0.9, 1.7, 2.2]):
        pr = O.Problem("FALLMASS", alg="BACKSOLVE", stepper="TSIT5", t0=0.0, t1=2.5, dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=7, checkpointing=True, checkpoints=ck)
        du0, dp, _ = pr.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), np.ones((len(ts), n)))
        assert relmax(du0, dp, g) < 1e-12
        cfg = E.make_config("emu_ball_terminate", "backsolve", 1, 0.0, 2.5, 0.0, ts, stepper=TS5, abstol=1e-12, reltol=1e-12, max_steps=4000, checkpointing=True, checkpoints=ck)
        edu0, edp, _ = E.forward_adjoint(cfg, n, 2, [g["u0"]], g["p"], np.ones((1, len(ts), n)))
        assert relmax(edu0[0], edp, g) < 1e-12


# ---- randomized differential test of the lane bodies (the GPU suite runs the same over the whole configuration space on the device) ---------------------------------------
@pytest.mark.parametrize("seed", range(16))
def test_fuzz_lane_bodies_vs_oracle(seed):
    """random problems — event kind, sensealg, stepper, checkpointing, tolerance, states, parameters, loss times, cotangents at the save times and at the saved event states —:
    the lane bodies against the oracle, event for event and trajectory for trajectory.  (This test found the rule "the lowest component wins" losing an earlier crossing of
    another component within the same tenth of a step.)"""
    rng = np.random.default_rng(9100 + seed)
    kind = int(rng.choice([1, 3, 4, 5, 5, 7]))
    emodel, omodel, n = {1: ("emu_ball", "FALLMASS", 2), 3: ("emu_relax", "RELAX", 1), 4: ("emu_ball_moving", "FALLMASS", 2), 5: ("emu_ball2d", "BALL2D", 4), 7: ("emu_ball_terminate", "FALLMASS", 2)}[kind]
    alg, oalg = ALGS[int(rng.integers(len(ALGS)))]
    ros = rng.uniform() < 0.3
    ck = bool(rng.uniform() < 0.35) and alg != "quadrature"
    tol = float(10.0 ** (rng.uniform(-9, -7) if ros else rng.uniform(-11, -8)))
    N = int(rng.integers(1, 7))
    if kind == 5:
        T = float(rng.uniform(6.0, 10.0))
        u0 = np.stack([rng.uniform(20.0, 60.0, N), rng.uniform(-2.0, 2.0, N), rng.uniform(1.0, 9.0, N), rng.uniform(0.5, 2.5, N) * rng.choice([-1.0, 1.0], N)], axis=1)
    elif kind == 3:
        T = float(rng.uniform(2.0, 10.0)); u0 = rng.uniform(0.0, 20.0, (N, 1))
    else:
        T = float(rng.uniform(2.0, 4.0)); u0 = np.stack([rng.uniform(2.0, 9.0, N) + (1.0 if kind == 4 else 0.0), rng.uniform(-1.0, 1.0, N)], axis=1)
    p = np.stack([100.0 * (1 + 0.1 * rng.uniform(-1, 1, N)), 50.0 * (1 + 0.2 * rng.uniform(-1, 1, N))], axis=1) if kind == 3 else np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    ts = np.sort(rng.uniform(0.05 * T, T, int(rng.integers(1, 6)))); d = rng.standard_normal((N, len(ts), n))
    w = rng.standard_normal((N, E.EMU_MAXEV, n)); v = rng.standard_normal((N, E.EMU_MAXEV, n))
    cfg = E.make_config(emodel, alg, N, 0.0, T, 0.0, ts, stepper=ROS if ros else TS5, abstol=tol, reltol=tol, max_steps=20000, p_shared=False, checkpointing=ck, **QTOL)
    evo = E.set_event_output(N, n); keep = E.set_event_cotangents(w, v)
    try:
        du0, dp, out = E.forward_adjoint(cfg, n, 2, u0, p, d)
    finally:
        E.set_event_cotangents(None, None); E.set_event_output(None)
    del keep
    for i in range(N):
        ref = O.Problem(omodel, alg=oalg, stepper="ROS23" if ros else "TSIT5", t0=0.0, t1=T, dt=0.0, abstol=tol, reltol=tol, save_times=ts, event_kind=kind, checkpointing=ck, **QTOL)
        rt, rul, rur = ref.event_states(u0[i], p[i])
        ne = int(np.sum(evo[i, :, 0] != 0.0))
        assert ne == len(rt) and np.max(np.abs(evo[i, :ne, 0] - rt), initial=0.0) < 1e-6, (seed, i, kind)
        ref.set_event_cotangents(w[i, :ne], v[i, :ne])
        rdu0, rdp, rout = ref.adjoint(u0[i], p[i], d[i])
        a = np.concatenate([du0[i], dp[i]]); b = np.concatenate([rdu0, rdp])
        assert np.max(np.abs(a - b)) <= (1e-5 if ros else 1e-6) * np.max(np.abs(b)), (seed, i, kind, alg, ros, ck)
        assert np.max(np.abs(out[i] - rout)) < 1e-5 * max(1.0, np.max(np.abs(rout)))


# ---- ContinuousCallback(condition, affect!, affect_neg!) with one affect `nothing`: only one crossing direction fires --------------------------------------------------
@pytest.mark.parametrize("direction,nev", [(0, 3), (1, 1), (-1, 2)])
def test_oracle_direction_against_finite_differences(direction, nev):
    """the damped pendulum (ORC_MODEL_PENDULUM) with c = angle, affect u2 <- p3 u2: the angle crosses zero downward, upward, downward within (0, 8).  Which crossings fire, and
    the gradient against central differences of the oracle's own forward solve (an independent check of the reverse jump: 1e-8 is the differences' precision)"""
    u0 = np.array([1.2, 0.0]); p = np.array([1.0, -0.1, 0.8]); ts = np.array([2.0, 5.0, 8.0]); d = np.array([[1.0, -0.5], [0.3, 0.7], [-1.0, 0.4]])
    kw = dict(alg="INTERPOLATING", stepper="TSIT5", t0=0.0, t1=8.0, dt=0.0, save_times=ts, event_kind=8, event_dir=direction)
    pr = O.Problem("PENDULUM", abstol=1e-12, reltol=1e-12, **kw)
    t, ul, ur = pr.event_states(u0, p)
    assert len(t) == nev and (direction == 0 or np.all(np.sign(ul[:, 1]) == direction))          # (p1 > 0: the angle rises where u2 > 0)
    du0, dp, _ = pr.adjoint(u0, p, d)
    fine = O.Problem("PENDULUM", abstol=1e-13, reltol=1e-13, **kw)
    G = lambda a, b: float(np.sum(fine.forward(a, b)[0] * d))
    fd = [(G(u0 + e, p) - G(u0 - e, p)) / 2e-6 for e in 1e-6 * np.eye(2)] + [(G(u0, p + e) - G(u0, p - e)) / 2e-6 for e in 1e-6 * np.eye(3)]
    assert np.max(np.abs(np.concatenate([du0, dp]) - fd)) < 2e-7
    for oalg in ("BACKSOLVE", "GAUSS", "QUADRATURE"):
        a = O.Problem("PENDULUM", abstol=1e-12, reltol=1e-12, **{**kw, "alg": oalg}, **QTOL).adjoint(u0, p, d)
        assert relc(a[0], du0) < 1e-7 and relc(a[1], dp) < 1e-7


def test_direction_registration_and_kernels():
    from scimlsensitivity_jl_amd import _lib
    m, cond, aff = UM.EVENTS[8]
    mid = _lib.register_model("cc_dir_host", m["n"], m["np"], m["f"])
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_callback_direction(mid, 1)                       # no callback yet
    assert ei.value.status == _lib.ERR_INVALID_ARG
    _lib.set_model_continuous_callback(mid, cond, aff, 8)
    with pytest.raises(_lib.HipadjError):
        _lib.set_model_callback_direction(mid, 2)
    L = _lib.load()
    for direction in (1, -1, 0):
        _lib.set_model_callback_direction(mid, direction)
        cfg = E.make_config("cc_dir_host", "interpolating", 53, 0.0, 8.0, 0.0, [2.0, 8.0], stepper=TS5, abstol=1e-8, reltol=1e-8)
        assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)


# ---- the C ABI without a device ------------------------------------------------------------------------------------------------------------------------------------------
def test_registration_entry_point_and_its_refusals():
    from scimlsensitivity_jl_amd import _lib
    m = UM.BALL
    mid = _lib.register_model("cc_host_ball", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    _lib.set_model_continuous_callback(mid, "c = u[0];", "un[1] = -p[1] * u[1];", 8)
    _lib.set_model_continuous_callback(mid, "c = u[0];", None)                      # a condition alone: the identity affect (the reference's "callbacks with no effect", :188-193)
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_continuous_callback(mid, None, "un[1] = 0.0;")
    assert ei.value.status == _lib.ERR_INVALID_ARG and "condition" in str(ei.value)
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_continuous_callback(mid, "c = u[0];", "pn[0] = 2.0 * p[0];")
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "parameter-changing" in str(ei.value)
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_continuous_callback(mid, "c = u[0];", None, -1)
    assert ei.value.status == _lib.ERR_INVALID_ARG
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_mass_matrix(mid, 2, [[2.0, 0.0], [0.0, 1.0]])
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "ContinuousCallback" in str(ei.value)
    with pytest.raises(_lib.HipadjError):
        _lib.set_model_continuous_callback(12345678, "c = u[0];", None)
    _lib.set_model_continuous_callback(mid, None, None)                              # removed: the mass matrix is accepted again
    _lib.set_model_mass_matrix(mid, 2, [[2.0, 0.0], [0.0, 1.0]])
    with pytest.raises(_lib.HipadjError) as ei:
        _lib.set_model_continuous_callback(mid, "c = u[0];", None)
    assert ei.value.status == _lib.ERR_UNSUPPORTED and "mass matrix" in str(ei.value)


@pytest.mark.parametrize("kind,auto,alg,stepper", [(1, False, "interpolating", TS5), (2, True, "gauss", TS5), (3, True, "gausskronrod", ROS), (4, False, "interpolating", ROS), (1, True, "backsolve", TS5), (4, False, "backsolve", ROS), (1, False, "quadrature", TS5), (3, True, "quadrature", ROS), (7, True, "interpolating", TS5), (7, False, "quadrature", ROS)])
@pytest.mark.parametrize("ck", [False, True])
def test_runtime_kernels_compile_without_a_device_and_are_clean(tmp_path, monkeypatch, kind, auto, alg, stepper, ck):
    """k_forward_tsit5<U, STEP> with the event search and k_adjoint_tsit5<U, ALG, 0, false, STEP> with the piecewise reverse solve and the jump, condition and affect from text
    (every derivative by dual numbers), through hiprtc; the spill-placement check on what it produced; and the planner's refusals for such a model"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "tools"))
    import isa_lint
    from scimlsensitivity_jl_amd import _lib
    if ck and alg == "quadrature":
        pytest.skip("QuadratureAdjoint has no checkpointing")
    m, cond, aff = UM.EVENTS[kind]
    name = f"cc_lint_{kind}_{int(auto)}_{alg}_{stepper}_{int(ck)}"
    mid = _lib.register_model(name, m["n"], m["np"], m["f"], None if auto else m["vjp"], None if auto else m["vjp_p"])
    _lib.set_model_continuous_callback(mid, cond, aff, 8)
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    L = _lib.load()
    cfg = E.make_config(name, alg, 53, 0.0, 2.5, 0.0, [0.5, 1.0, 2.5], stepper=stepper, abstol=1e-8, reltol=1e-8, checkpointing=ck)
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs
    for o in objs:
        assert isa_lint.lint(o) == []
    for bad, word in (
                      (dict(cont_cost=1), "continuous cost"), (dict(stepper=0, dt=0.01), "adaptive steppers")):
        kw = dict(alg=alg, stepper=stepper, dt=0.0, checkpointing=False, cont_cost=0); kw.update(bad)
        cfg = E.make_config(name, kw["alg"], 53, 0.0, 2.5, kw["dt"], [0.5, 1.0, 2.5], stepper=kw["stepper"], abstol=1e-8, reltol=1e-8, checkpointing=kw["checkpointing"], cont_cost=kw["cont_cost"])
        assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.ERR_UNSUPPORTED and word in L.hipadj_last_error(None).decode()
