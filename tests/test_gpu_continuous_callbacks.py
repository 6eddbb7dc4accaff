"""ContinuousCallback on the device (`-m gpu`): hipadj_model_set_continuous_callback through the host mirror (`DeviceFunction.set_continuous_callback`) — the reference's
test/Callbacks2/continuous_callbacks.jl (the bouncing ball and its relatives; src/callback_tracking.jl:232-479) on the adaptive lane steppers.

  * the reference's own problems as runtime models (condition and affect as text, every derivative of the jump by dual numbers): against the CLOSED-FORM gradients of
    tests/golden/continuous_callbacks.json at the reference's bar (rtol 1e-5, :140-145; the "Re-compile tape" problem 1e-10 at tolerances 1e-14, :342-346 — here 1e-8 at the
    tolerances the device takes, 1e-12) and against the oracle;
  * an ensemble in which every trajectory has its own event times and its own NUMBER of events: all trajectories against the oracle, event counts against the closed form;
  * Rosenbrock23 as the stepper; f-only models (VJPs by dual numbers);
  * the event list overflowing its capacity: reported, not silently truncated;
  * what the library refuses for a model with a ContinuousCallback.
Tolerances: the device and the oracle locate an event on two representations of the same dense output (monomial record / stage form) and restart the reverse controller
at it; they agree to a fraction of the solver tolerance."""
import json
import os

import numpy as np
import pytest

import oracle as O
import user_models as UM
from test_gpu_parity import rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ALGS = [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD"), ("quadrature", "QUADRATURE")]
QTOL = dict(quad_abstol=1e-14, quad_reltol=1e-12)
CASES = {"ball": (1, "FALLMASS"), "ball_long": (1, "FALLMASS"), "ball_mse": (2, "FALLMASS"), "relax": (3, "RELAX"), "moving": (4, "FALLMASS")}
_registered = {}


def relc(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12 * np.max(np.abs(b)))))      # (component-wise; a gradient entry that is exactly zero is held to the vector's scale)


def sens(sa, alg):
    return {"interpolating": sa.InterpolatingAdjoint(), "backsolve": sa.BacksolveAdjoint(checkpointing=False), "gauss": sa.GaussAdjoint(), "gausskronrod": sa.GaussKronrodAdjoint(),
            "quadrature": sa.QuadratureAdjoint(abstol=1e-14, reltol=1e-12)}[alg]


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "continuous_callbacks.json")) as f:
        return json.load(f)


def model(sa, kind, auto=False, max_events=0):
    key = (kind, auto, max_events)
    if key not in _registered:
        m, cond, aff = UM.EVENTS[kind]
        f = sa.DeviceFunction(f"cc_kind{kind}_{int(auto)}_{max_events}", m["n"], m["np"], m["f"], None if auto else m["vjp"], None if auto else m["vjp_p"])
        f.set_continuous_callback(cond, aff, max_events)
        _registered[key] = f
    return _registered[key]


def run(sa, f, g, alg, stepper, tol, mse=False, u0=None, p=None):
    ts = np.asarray(g["ts"]); n = len(g["u0"])
    u0 = np.asarray([g["u0"]]) if u0 is None else u0
    p = np.asarray(g["p"]) if p is None else p
    pr = sa.EnsembleProblem(sa.ODEProblem(f, u0[0], tuple(g["tspan"]), p if p.ndim == 1 else p[0]), u0, None if p.ndim == 1 else p)
    kw = dict(dgdu_discrete=sa.LsqShift(1.0)) if mse else {}
    sol = sa.solve(pr, stepper, saveat=ts, sensealg=sens(sa, alg), abstol=tol, reltol=tol, **kw)
    if mse:
        du0, dp = sa.adjoint_sensitivities(sol, stepper, t=ts)
    else:
        du0, dp = sa.adjoint_sensitivities(sol, stepper, t=ts, dgdu_discrete=np.ones((len(u0), len(ts), n)))
    ne = sol.engine.event_counts()
    out = np.array(sol.u)
    sol.engine.close()
    return du0, dp, out, ne


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", ["ball", "ball_long", "ball_mse", "relax", "moving"])
def test_reference_problems_against_the_closed_forms_and_the_oracle(sa, gold, case, alg, oalg):
    kind, omodel = CASES[case]; g = gold[case]; mse = case == "ball_mse"
    du0, dp, out, ne = run(sa, model(sa, kind), g, alg, sa.Tsit5(), 1e-12, mse=mse)
    assert ne.tolist() == [len(g["event_times"])]
    bar = 1e-8 if case == "relax" else 1e-9          # (relax: an exponential solved at 1e-12; the others are polynomials in t, exact for the stepper)
    assert relc(du0[0], g["du0"]) < bar and relc(dp, g["dp"]) < bar
    ts = np.asarray(g["ts"])
    ref = O.Problem(omodel, alg=oalg, stepper="TSIT5", t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=1e-12, reltol=1e-12, save_times=ts, event_kind=kind,
                    loss="LSQ_SHIFT" if mse else "COTANGENT", loss_shift=1.0 if mse else 0.0, **QTOL)
    rdu0, rdp, rout = ref.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), None if mse else np.ones((len(ts), len(g["u0"]))))[:3]
    assert rel(out[0], rout) < 1e-10 and relc(du0[0], rdu0) < 1e-9 and relc(dp, rdp) < 1e-9
    if "u_at_ts" in g:
        assert np.max(np.abs(out[0] - np.asarray(g["u_at_ts"]))) < 1e-9


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_an_ensemble_where_every_trajectory_has_its_own_events(sa, alg, oalg):
    """300 balls dropped from 2 .. 9 with restitution 0.8 .. 0.9 over (0, 4) — none comes to rest before 4: the bounces of such a ball accumulate at t1 + 2 v1 e / (g (1 - e)) > 5.7 —:
    two to seven bounces each, at times no two trajectories share"""
    rng = np.random.default_rng(5)
    N, T = 300, 4.0
    u0 = np.stack([rng.uniform(2.0, 9.0, N), rng.uniform(-1.0, 1.0, N)], axis=1)
    p = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    ts = np.array([0.3, 1.0, 1.7, 2.2, 3.1, 4.0])
    d = rng.standard_normal((N, len(ts), 2))
    f = model(sa, 1)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, T), p[0]), u0, p), sa.Tsit5(), saveat=ts, sensealg=sens(sa, alg), abstol=1e-10, reltol=1e-10)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=d)
    ne = sol.engine.event_counts(); out = np.array(sol.u)
    sol.engine.close()
    ref = O.Problem("FALLMASS", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, event_kind=1, **QTOL)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, d)
    assert len(set(ne.tolist())) >= 3 and ne.min() >= 1
    assert rel(out, rout) < 1e-8
    sc = np.maximum(np.abs(rdu0), 1e-3 * np.abs(rdu0).max(axis=1, keepdims=True)); scp = np.maximum(np.abs(rdp), 1e-3 * np.abs(rdp).max(axis=1, keepdims=True))
    assert np.max(np.abs(du0 - rdu0) / sc) < 1e-6 and np.max(np.abs(dp - rdp) / scp) < 1e-6
    # the number of bounces in closed form: flight k lasts 2 v_k / g with v_{k+1} = e v_k
    for i in range(0, N, 7):
        x, v, gg, e = u0[i, 0], u0[i, 1], p[i, 0], p[i, 1]
        t = (v + np.sqrt(v * v + 2 * gg * x)) / gg; vk = e * np.sqrt(v * v + 2 * gg * x); k = 0
        while t < T and k < 100:
            k += 1; t += 2 * vk / gg; vk *= e
        assert ne[i] == k


@pytest.mark.parametrize("alg,oalg", [ALGS[0], ALGS[1], ALGS[2], ALGS[4]])
@pytest.mark.parametrize("case", ["ball", "relax"])
def test_rosenbrock23_and_dual_number_vjps(sa, gold, case, alg, oalg):
    kind, omodel = CASES[case]; g = gold[case]
    du0, dp, out, ne = run(sa, model(sa, kind, auto=True), g, alg, sa.Rosenbrock23(), 1e-9)
    assert ne.tolist() == [len(g["event_times"])]
    a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g["du0"], g["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-5            # the reference's bar, :140-145, on the whole gradient (relax: du0 is 5e-5 of dp[0])
    ts = np.asarray(g["ts"])
    ref = O.Problem(omodel, alg=oalg, stepper="ROS23", t0=g["tspan"][0], t1=g["tspan"][1], dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, event_kind=kind, **QTOL)
    rdu0, rdp, rout = ref.adjoint(np.asarray(g["u0"]), np.asarray(g["p"]), np.ones((len(ts), len(g["u0"]))))[:3]
    assert rel(out[0], rout) < 1e-8 and relc(du0[0], rdu0) < 1e-6 and relc(dp, rdp) < 1e-6


@pytest.mark.parametrize("case", ["ball", "ball_long", "relax"])
def test_backsolve_with_checkpoints_through_events(sa, gold, case):
    """BacksolveAdjoint() as the reference's callback tests call it (checkpointing = true, the default; test/Callbacks2/continuous_callbacks.jl:46-57): the backsolved state is
    overwritten at the checkpoints (the save times) and, with the stored left state, at every event"""
    kind, omodel = CASES[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model(sa, kind), u0[0], tuple(g["tspan"]), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.BacksolveAdjoint(), abstol=1e-12, reltol=1e-12)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), n)))
    sol.engine.close()
    a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g["du0"], g["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-9


def test_more_events_than_the_list_holds_is_an_error(sa, gold):
    g = gold["ball_long"]                     # four bounces
    with pytest.raises(Exception) as ei:
        run(sa, model(sa, 1, max_events=2), g, "interpolating", sa.Tsit5(), 1e-10)
    assert "max_events" in str(ei.value)
    du0, dp, out, ne = run(sa, model(sa, 1, max_events=4), g, "interpolating", sa.Tsit5(), 1e-10)
    assert ne.tolist() == [4] and relc(dp, g["dp"]) < 1e-7


def test_a_model_without_crossings_equals_the_model_without_the_callback(sa):
    """the ball thrown upward from 50 never reaches the floor within (0, 2): the callback must change nothing, bit for bit"""
    m = UM.BALL
    ts = np.array([0.5, 1.0, 2.0]); u0 = np.array([[50.0, 3.0]]); p = np.array([9.8, 0.8]); d = np.ones((1, 3, 2))
    res = []
    for cc in (False, True):
        f = sa.DeviceFunction(f"cc_none_{int(cc)}", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
        if cc:
            f.set_continuous_callback("c = u[0];", "un[1] = -p[1] * u[1];")
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 2.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-9, reltol=1e-9)
        res.append(sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=d) + (np.array(sol.u),))
        if cc:
            assert sol.engine.event_counts().tolist() == [0]
        sol.engine.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_refusals(sa, gold):
    from scimlsensitivity_jl_amd import _lib
    g = gold["ball"]; f = model(sa, 1); ts = np.asarray(g["ts"]); u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    pr = sa.EnsembleProblem(sa.ODEProblem(f, u0[0], tuple(g["tspan"]), p), u0)
    for stepper, alg, kw, word in ((sa.RK4(), sa.InterpolatingAdjoint(), dict(dt=0.01), "adaptive steppers"),):
        with pytest.raises(_lib.HipadjError) as ei:
            sa.solve(pr, stepper, saveat=ts, sensealg=alg, abstol=1e-8, reltol=1e-8, **kw)
        assert ei.value.status == _lib.ERR_UNSUPPORTED and word in str(ei.value)
    with pytest.raises(_lib.HipadjError):
        f.set_mass_matrix([[2.0, 0.0], [0.0, 1.0]])
    with pytest.raises(_lib.HipadjError):
        model(sa, 1).set_continuous_callback("c = u[0];", "pn[0] = 2.0 * p[0];")
    f.set_continuous_callback("c = u[0];", "un[1] = -p[1] * u[1];")          # (the failed call left the callback as it was — restore explicitly all the same)


def test_the_solve_keyword_attaches_the_callback(sa, gold):
    """solve(...; callback = ContinuousCallback(condition, affect!)) as the reference writes it (test/Callbacks2/continuous_callbacks.jl:37-45)"""
    g = gold["ball"]; m = UM.BALL; ts = np.asarray(g["ts"]); u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    f = sa.DeviceFunction("cc_keyword_ball", m["n"], m["np"], m["f"])                 # f only: VJPs by dual numbers
    cb = sa.ContinuousCallback("c = u[0];", "un[1] = -p[1] * u[1];")
    for _ in range(2):                                                                # (the second solve finds the callback attached: no new code object)
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], tuple(g["tspan"]), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-12, reltol=1e-12, callback=cb)
        du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), 2)))
        assert sol.engine.event_counts().tolist() == [1]
        sol.engine.close()
        assert relc(du0[0], g["du0"]) < 1e-9 and relc(dp, g["dp"]) < 1e-9
    with pytest.raises(ValueError):
        sa.solve(sa.EnsembleProblem(sa.ODEProblem("fallmass", u0[0], tuple(g["tspan"]), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-8, reltol=1e-8, callback=cb)


@pytest.mark.parametrize("alg", [0, 2, 4])
def test_c_example_of_the_bouncing_ball_matches_the_closed_form(sa, gold, tmp_path, alg):
    """examples/bouncing_ball_demo.c: plain C against include/hipadj.h — hipadj_model_register (f only), hipadj_model_set_continuous_callback, a Tsit5 handle, host-pointer
    forward / event counts / adjoint; trajectory 0 is the reference's problem."""
    import subprocess
    root = os.path.dirname(HERE)
    exe, libdir = str(tmp_path / "bouncing_ball_demo"), os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "bouncing_ball_demo.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir, "-lm"])
    r = subprocess.run([exe, "5", str(alg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    val = {l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in r.stdout.strip().split("\n")}
    g = gold["ball"]
    assert relc(val["du0"], g["du0"]) < 1e-9 and relc(val["dp"], g["dp"]) < 1e-9
    assert np.max(np.abs(val["u_at_2.5"] - np.asarray(g["u_at_ts"])[-1])) < 1e-9
    assert val["events"].tolist() == [1, 1, 1, 1, 1] and val["rk4_on_the_callback"][0] == -6
    assert np.all(np.isfinite(val["dp_last"])) and relc(val["dp_last"], val["dp"]) > 1e-3


def test_python_example_of_the_bouncing_ball(sa, gold):
    import subprocess, sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "bouncing_ball.py"), "256"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    dev = [l for l in r.stdout.splitlines() if l.startswith("device")][0]
    nums = [float(x) for x in dev.replace("[", " ").replace("]", " ").replace(",", " ").split() if x.replace(".", "").replace("-", "").replace("e", "").replace("+", "").isdigit()]
    assert relc(nums[:2], gold["ball"]["du0"]) < 1e-6 and relc(nums[2:4], gold["ball"]["dp"]) < 1e-6


SAVED = {"ball_saved": 1, "ball_long_saved": 1, "ball_mse_saved": 2, "moving_saved": 4}


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", sorted(SAVED))
def test_loss_on_the_saved_event_states(sa, gold, case, alg, oalg):
    """save_positions = (true, true) — the constructor's default, the setting of most of the reference's testsets (test/Callbacks2/continuous_callbacks.jl:200-217, 239-250): the loss
    also takes the state just before and just after every affect.  hipadj_event_states hands them over, hipadj_set_event_cotangents takes the loss's cotangents there; event
    states and gradients against the closed forms, every sensealg"""
    kind = SAVED[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = 2; mse = "mse" in case
    es = np.asarray(g["event_states"]); ne = len(es)
    u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    kw = dict(dgdu_discrete=sa.LsqShift(1.0)) if mse else {}
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model(sa, kind), u0[0], tuple(g["tspan"]), p), u0), sa.Tsit5(), saveat=ts, sensealg=sens(sa, alg), abstol=1e-12, reltol=1e-12, **kw)
    t, ul, ur, cnt = sol.engine.event_states()
    assert cnt.tolist() == [ne] and np.max(np.abs(t[0, :ne] - np.asarray(g["event_times"]))) < 1e-11
    assert np.max(np.abs(ul[0, :ne] - es[:, 0])) < 1e-9 and np.max(np.abs(ur[0, :ne] - es[:, 1])) < 1e-9 and np.all(ul[0, ne:] == 0.0)
    dl = np.zeros_like(ul); dr = np.zeros_like(ur)
    dl[0, :ne] = (ul[0, :ne] - 1.0) if mse else 1.0; dr[0, :ne] = (ur[0, :ne] - 1.0) if mse else 1.0
    sol.engine.set_event_cotangents(dl, dr)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts) if mse else sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), n)))
    a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g["du0"], g["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-9
    sol.engine.set_event_cotangents(None, None)                                   # removed: the gradient of the loss at the save times alone
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts) if mse else sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), n)))
    sol.engine.close()
    g0 = gold[case[:-6]]
    a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g0["du0"], g0["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-9


def test_saved_event_states_of_an_ensemble(sa):
    """per-trajectory event counts: the ragged set of saved states, one row per trajectory, zero beyond its count; u+ = affect(u-), c(u-) = 0; the loss sum over all saved states
    against the oracle for a sample"""
    rng = np.random.default_rng(9)
    N, T = 200, 4.0
    u0 = np.stack([rng.uniform(2.0, 9.0, N), rng.uniform(-1.0, 1.0, N)], axis=1)
    p = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    ts = np.array([0.3, 1.7, 4.0]); d = rng.standard_normal((N, len(ts), 2))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model(sa, 1), u0[0], (0.0, T), p[0]), u0, p), sa.Tsit5(), saveat=ts, sensealg=sa.InterpolatingAdjoint(), abstol=1e-10, reltol=1e-10)
    t, ul, ur, cnt = sol.engine.event_states()
    assert cnt.min() >= 1 and cnt.max() <= 8 and len(set(cnt.tolist())) >= 3
    for i in range(N):
        k = cnt[i]
        assert np.all(np.diff(t[i, :k]) > 0) and np.all(t[i, k:] == 0.0) and np.max(np.abs(ul[i, :k, 0])) < 1e-9          # on the floor, in order
        assert np.max(np.abs(ur[i, :k, 1] + p[i, 1] * ul[i, :k, 1])) < 1e-12 and np.all(ur[i, :k, 0] == ul[i, :k, 0])      # u+ = affect(u-)
    w = rng.standard_normal(ul.shape); v = rng.standard_normal(ur.shape)
    sol.engine.set_event_cotangents(w, v)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=d)
    sol.engine.close()
    for i in range(0, N, 17):
        ref = O.Problem("FALLMASS", alg="INTERPOLATING", stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, event_kind=1)
        ref.set_event_cotangents(w[i, :cnt[i]], v[i, :cnt[i]])
        rdu0, rdp, _ = ref.adjoint(u0[i], p[i], d[i])
        a = np.concatenate([du0[i], dp[i]]); b = np.concatenate([rdu0, rdp])
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-7


def vmodel(sa, kind, auto=False):
    key = ("vec", kind, auto)
    if key not in _registered:
        m, nc, cond, aff = UM.VECTOR_EVENTS[kind]
        f = sa.DeviceFunction(f"cc_vec{kind}_{int(auto)}", m["n"], m["np"], m["f"], None if auto else m["vjp"], None if auto else m["vjp_p"])
        f.set_continuous_callback(cond, aff, 0, ncond=nc)
        _registered[key] = f
    return _registered[key]


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case,kind", [("walls", 5), ("walls_saved", 5), ("clock", 6), ("clock_saved", 6)])
def test_vector_callback_against_the_closed_forms(sa, gold, case, kind, alg, oalg):
    """VectorContinuousCallback (test/Callbacks2/vector_continuous_callbacks.jl:80-116): the ball between two walls and the time-only conditions, condition and affect as text
    (`out[k]`, `idx`), MSE loss with and without the saved event states: event times, which component fired, states and gradients against the closed forms, every sensealg"""
    g = gold[case]; ts = np.asarray(g["ts"]); saved = case.endswith("_saved"); ne = len(g["event_times"])
    u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(vmodel(sa, kind, auto=(alg == "gauss")), u0[0], (0.0, 10.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sens(sa, alg), abstol=1e-12, reltol=1e-12,
                   dgdu_discrete=sa.LsqShift(1.0))
    t, ul, ur, cnt = sol.engine.event_states()
    assert cnt.tolist() == [ne] and np.max(np.abs(t[0, :ne] - np.asarray(g["event_times"]))) < 1e-10
    comp = sol.engine.event_components()
    assert comp[0, :ne].tolist() == g["event_components"] and np.all(comp[0, ne:] == -1)          # which component fired, in order (the reference's event_idx)
    assert np.max(np.abs(np.array(sol.u)[0] - np.asarray(g["u_at_ts"]))) < 1e-9
    if saved:
        es = np.asarray(g["event_states"])
        assert np.max(np.abs(ul[0, :ne] - es[:, 0])) < 1e-9 and np.max(np.abs(ur[0, :ne] - es[:, 1])) < 1e-9
        dl = np.zeros_like(ul); dr = np.zeros_like(ur); dl[0, :ne] = ul[0, :ne] - 1.0; dr[0, :ne] = ur[0, :ne] - 1.0
        sol.engine.set_event_cotangents(dl, dr)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts)
    sol.engine.close()
    a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g["du0"], g["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-9


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("case", ["ball_terminate", "ball_terminate_saved"])
def test_terminate_against_the_closed_forms(sa, gold, case, alg, oalg):
    """terminate!(integrator) (test/Callbacks2/continuous_callbacks.jl:226-236): `terminate = true;` in the affect body ends the trajectory's solve at the event; later save
    times hold the final state and carry no loss.  One terminating and one ordinary trajectory side by side would not share a closed form: an ensemble of two balls dropped
    from 5 and 6, both terminating at their own first bounce"""
    g = gold[case]; ts = np.asarray(g["ts"]); n = 2; saved = case.endswith("_saved"); es = np.asarray(g["event_states"]); nb = len(g["u_at_ts"])
    u0 = np.asarray([g["u0"], [6.0, 0.0]]); p = np.asarray(g["p"])
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model(sa, 7), u0[0], (0.0, 2.5), p), u0, np.tile(p, (2, 1))), sa.Tsit5(), saveat=ts, sensealg=sens(sa, alg), abstol=1e-12, reltol=1e-12)
    t, ul, ur, cnt = sol.engine.event_states()
    assert cnt.tolist() == [1, 1] and abs(t[0, 0] - g["event_times"][0]) < 1e-11 and abs(t[1, 0] - np.sqrt(12.0 / 9.8)) < 1e-11
    assert sol.engine.event_components()[:, 0].tolist() == [256, 256]                               # component 0, terminating
    out = np.array(sol.u)
    assert np.max(np.abs(out[0, :nb] - np.asarray(g["u_at_ts"]))) < 1e-10 and np.max(np.abs(out[0, nb:] - es[0, 1])) < 1e-9 and np.max(np.abs(ur[0, 0] - es[0, 1])) < 1e-9
    if saved:
        dl = np.zeros_like(ul); dr = np.zeros_like(ur); dl[:, 0] = 1.0; dr[:, 0] = 1.0
        sol.engine.set_event_cotangents(dl, dr)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((2, len(ts), n)))
    sol.engine.close()
    a = np.concatenate([du0[0], dp[0]]); b = np.concatenate([g["du0"], g["dp"]])
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-9
    assert np.all(np.isfinite(du0[1])) and np.all(np.isfinite(dp[1]))
    if saved:
        assert abs(dp[1, 1] - dp[0, 1]) > 1e-3          # (the second ball hits the floor faster: another d/d restitution of the saved state after the bounce)


@pytest.mark.parametrize("alg", ["interpolating", "gauss", "gausskronrod"])
@pytest.mark.parametrize("case", ["ball", "ball_long", "relax", "moving", "ball_terminate"])
def test_checkpointed_interpolating_and_gauss_through_events(sa, gold, case, alg):
    """InterpolatingAdjoint(checkpointing = true) — du03 of the reference's callback tests (test/Callbacks2/continuous_callbacks.jl:99-110) — and the checkpointed Gauss sweeps
    through events: default checkpoints (the save times) and a list, against the closed forms"""
    kind = {"ball": 1, "ball_long": 1, "relax": 3, "moving": 4, "ball_terminate": 7}[case]; g = gold[case]; ts = np.asarray(g["ts"]); n = len(g["u0"])
    u0 = np.asarray([g["u0"]]); p = np.asarray(g["p"])
    inner = {"interpolating": sa.InterpolatingAdjoint, "gauss": sa.GaussAdjoint, "gausskronrod": sa.GaussKronrodAdjoint}[alg](checkpointing=True)
    for ck in (None, [0.2, 0.9, 1.1, 2.4]):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model(sa, kind), u0[0], tuple(g["tspan"]), p), u0), sa.Tsit5(), saveat=ts, sensealg=inner, abstol=1e-12, reltol=1e-12, checkpoints=ck)
        du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=np.ones((1, len(ts), n)))
        sol.engine.close()
        a = np.concatenate([du0[0], np.ravel(dp)]); b = np.concatenate([g["du0"], g["dp"]])
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < (1e-8 if case == "relax" else 1e-9)


# ---- randomized differential test: the callback path over its configuration space, device vs oracle ---------------------------------------------------------------------
def _random_event_case(rng):
    kind = int(rng.choice([1, 1, 2, 3, 4, 5, 5, 6, 7]))
    alg = ALGS[int(rng.integers(len(ALGS)))]
    stepper = "ROS23" if rng.uniform() < 0.3 else "TSIT5"
    ck = bool(rng.uniform() < 0.35) and alg[0] != "quadrature"
    tol = float(10.0 ** rng.uniform(-11, -8)) if stepper == "TSIT5" else float(10.0 ** rng.uniform(-9, -7))
    N = int(rng.integers(1, 40))
    if kind in (5, 6):
        T = float(rng.uniform(6.0, 10.0))
        u0 = np.stack([rng.uniform(20.0, 60.0, N), rng.uniform(-2.0, 2.0, N), rng.uniform(1.0, 9.0, N), rng.uniform(0.5, 2.5, N) * rng.choice([-1.0, 1.0], N)], axis=1)
    elif kind == 3:
        T = float(rng.uniform(2.0, 10.0))
        u0 = rng.uniform(0.0, 20.0, (N, 1))
    else:
        T = float(rng.uniform(2.0, 4.0))
        u0 = np.stack([rng.uniform(2.0, 9.0, N), rng.uniform(-1.0, 1.0, N)], axis=1)
        if kind == 4:
            u0[:, 0] += 1.0
    p = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, N)), rng.uniform(0.8, 0.9, N)], axis=1)
    if kind == 3:
        p = np.stack([100.0 * (1 + 0.1 * rng.uniform(-1, 1, N)), 50.0 * (1 + 0.2 * rng.uniform(-1, 1, N))], axis=1)
    M = int(rng.integers(1, 7))
    ts = np.sort(rng.uniform(0.05 * T, T, M)); ts[-1] = T if rng.uniform() < 0.5 else ts[-1]
    saved = bool(rng.uniform() < 0.5)
    return dict(kind=kind, alg=alg, stepper=stepper, ck=ck, tol=tol, N=N, T=T, u0=u0, p=p, ts=ts, saved=saved)


@pytest.mark.parametrize("seed", range(96))
def test_fuzz_events_device_vs_oracle(sa, seed):
    """random problems of the callback path — event kind (scalar, non-linear affect, explicit t, two components, terminating), sensealg, stepper, checkpointing, tolerances, ensemble
    size, per-trajectory states and parameters, loss times anywhere, random cotangents at the save times and (half of the cases) at the saved event states — device against the
    oracle, every trajectory.  A case whose event count differs between the two (an event within rounding of the end of the span) is skipped, not failed"""
    rng = np.random.default_rng(7000 + seed)
    c = _random_event_case(rng)
    kind, (alg, oalg), N, n = c["kind"], c["alg"], c["N"], c["u0"].shape[1]
    f = vmodel(sa, kind) if kind in (5, 6) else model(sa, kind)
    stepper = sa.Rosenbrock23() if c["stepper"] == "ROS23" else sa.Tsit5()
    inner = sens(sa, alg) if not c["ck"] else {"interpolating": sa.InterpolatingAdjoint, "gauss": sa.GaussAdjoint, "gausskronrod": sa.GaussKronrodAdjoint, "backsolve": sa.BacksolveAdjoint}[alg](checkpointing=True)
    d = rng.standard_normal((N, len(c["ts"]), n))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, c["u0"][0], (0.0, c["T"]), c["p"][0]), c["u0"], c["p"]), stepper, saveat=c["ts"], sensealg=inner, abstol=c["tol"], reltol=c["tol"])
    t, ul, ur, cnt = sol.engine.event_states()
    w = rng.standard_normal(ul.shape); v = rng.standard_normal(ur.shape)
    if c["saved"]:
        sol.engine.set_event_cotangents(w, v)
    du0, dp = sa.adjoint_sensitivities(sol, stepper, t=c["ts"], dgdu_discrete=d)
    out = np.array(sol.u)
    sol.engine.close()
    omodel = "BALL2D" if kind in (5, 6) else ("RELAX" if kind == 3 else "FALLMASS")
    for i in range(N):
        ref = O.Problem(omodel, alg=oalg, stepper=c["stepper"], t0=0.0, t1=c["T"], dt=0.0, abstol=c["tol"], reltol=c["tol"], save_times=c["ts"], event_kind=kind, checkpointing=c["ck"], **QTOL)
        rt, rul, rur = ref.event_states(c["u0"][i], c["p"][i])
        if len(rt) != cnt[i]:
            pytest.skip(f"trajectory {i}: {cnt[i]} events on the device, {len(rt)} in the oracle (an event within rounding of a decision)")
        assert np.max(np.abs(t[i, :cnt[i]] - rt), initial=0.0) < 1e-6
        if c["saved"]:
            ref.set_event_cotangents(w[i, :cnt[i]], v[i, :cnt[i]])
        rdu0, rdp, rout = ref.adjoint(c["u0"][i], c["p"][i], d[i])
        a = np.concatenate([du0[i], dp[i]]); b = np.concatenate([rdu0, rdp])
        bar = 1e-5 if c["stepper"] == "ROS23" else 1e-6
        assert np.max(np.abs(a - b)) <= bar * np.max(np.abs(b)), (seed, i, c["kind"], alg, c["stepper"], c["ck"], c["saved"])      # (a terminating event ahead of every loss time: both are exactly zero)
        assert np.max(np.abs(out[i] - rout)) < 1e-5 * max(1.0, np.max(np.abs(rout)))


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("direction,nev", [(0, 3), (1, 1), (-1, 2)])
def test_callback_direction(sa, direction, nev, alg, oalg):
    """ContinuousCallback(condition, affect!, affect_neg!) with one affect `nothing` (hipadj_model_set_callback_direction): the damped pendulum, c = angle, affect u2 <- p3 u2 —
    the angle crosses zero downward, upward, downward within (0, 8); which crossings fire and the gradients, an ensemble of perturbed pendulums against the oracle (whose
    gradient for this problem is checked against finite differences in the CPU suite)"""
    rng = np.random.default_rng(3)
    N = 24
    u0 = np.array([1.2, 0.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.0, -0.1, 0.8]) * (1 + 0.02 * rng.standard_normal((N, 3)))
    u0[0] = [1.2, 0.0]; p[0] = [1.0, -0.1, 0.8]
    ts = np.array([2.0, 5.0, 8.0]); d = rng.standard_normal((N, 3, 2))
    m, cond, aff = UM.EVENTS[8]
    key = ("dir", direction)
    if key not in _registered:
        f = sa.DeviceFunction(f"cc_pendulum_dir{direction + 1}", m["n"], m["np"], m["f"])
        f.set_continuous_callback(cond, aff, direction=direction)
        _registered[key] = f
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(_registered[key], u0[0], (0.0, 8.0), p[0]), u0, p), sa.Tsit5(), saveat=ts, sensealg=sens(sa, alg), abstol=1e-11, reltol=1e-11)
    t, ul, ur, cnt = sol.engine.event_states()
    assert cnt[0] == nev and (direction == 0 or np.all(np.sign(ul[0, :nev, 1]) == direction))
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=d)
    sol.engine.close()
    ref = O.Problem("PENDULUM", alg=oalg, stepper="TSIT5", t0=0.0, t1=8.0, dt=0.0, abstol=1e-11, reltol=1e-11, save_times=ts, event_kind=8, event_dir=direction, **QTOL)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, d)
    a = np.concatenate([du0, dp], axis=1); b = np.concatenate([rdu0, rdp], axis=1)
    assert np.max(np.abs(a - b) / np.max(np.abs(b), axis=1, keepdims=True)) < 1e-6
