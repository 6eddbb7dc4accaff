"""The one-launch reverse pass (csrc/hipadj_fused.hpp) against the three-launch sequence it replaces (`-m gpu`).

The composition tree hands segment maps from wave to wave through HBM inside one launch (write-through stores, arrival counters,
sc1 loads) and reuses the same slots and counters every pass.  A stale cache line or a counter left over from the previous pass
would not show in a test that repeats one computation — the stale data would be the right data — so every round here changes
the inputs (initial states, parameters, cotangents) and the same handle runs many rounds back to back.  The reference engine is
the same library with HIPADJ_FUSED=0 (k_interp + k_compose_finish* + k_reduce_final: round 2's path, oracle-checked at size)."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import RTOL, rel, lorenz_inputs

pytestmark = pytest.mark.gpu


def _engines(sa, monkeypatch, N, ts, T, dt, loss_kind, p_shared=True, segments=0, radix=None, alg="interpolating", model="lorenz", user_cap=None, fused_launches=1, **more):
    kw = dict(save_times=ts, loss_kind=loss_kind, loss_shift=2.0, p_shared=p_shared, time_segments=segments, **more)
    monkeypatch.setenv("HIPADJ_FUSED", "0")
    ref = sa.Engine(model, alg, N, 0.0, T, dt, **kw)
    monkeypatch.setenv("HIPADJ_FUSED", "1")
    if radix:
        monkeypatch.setenv("HIPADJ_TREE_RADIX", str(radix))
    if user_cap:
        monkeypatch.setenv("HIPADJ_FUSED_USER_CAP", str(user_cap))
    fus = sa.Engine(model, alg, N, 0.0, T, dt, **kw)
    assert fus.stats()["launches_per_pass"] == fused_launches and ref.stats()["launches_per_pass"] == 3
    monkeypatch.delenv("HIPADJ_FUSED")
    monkeypatch.delenv("HIPADJ_TREE_RADIX", raising=False)
    monkeypatch.delenv("HIPADJ_FUSED_USER_CAP", raising=False)
    return ref, fus


@pytest.mark.parametrize("N,radix", [(1250, 4), (1250, 8), (10000, 4), (777, 4), (64, 4), (65, 4)])
def test_fused_pass_equals_three_launch_sequence_on_changing_data(sa, monkeypatch, N, radix):
    T, dt = 10.0, 0.01
    ts = np.linspace(0.0, T, 101)
    ref, fus = _engines(sa, monkeypatch, N, ts, T, dt, loss_kind=0, radix=radix)
    assert fus.stats()["time_segments"] == ref.stats()["time_segments"] > 1
    rng = np.random.default_rng(N)
    u0, p = lorenz_inputs(N)
    for rnd in range(6):
        u0r = u0 + 0.01 * rng.standard_normal(u0.shape)
        pr = p * (1.0 + 0.01 * rng.standard_normal(3))
        ref.forward(u0r, pr, want_out=False); fus.forward(u0r, pr, want_out=False)
        for rep in range(4):          # several reverse passes per forward solution, new cotangents each time: slots and counters are reused immediately
            delta = rng.standard_normal((N, len(ts), 3)) * (1.0 + rep)
            a, b = ref.adjoint(delta), fus.adjoint(delta)
            assert rel(b[0], a[0]) < 1e-11, (rnd, rep)
            assert np.max(np.abs(b[1] - a[1]) / np.abs(a[1])) < 1e-10, (rnd, rep)
    # the same inputs twice: bit-for-bit (fixed bracketing, fixed summation order)
    delta = rng.standard_normal((N, len(ts), 3))
    a, b = fus.adjoint(delta), fus.adjoint(delta)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ref.close(); fus.close()


def test_fused_pass_with_per_trajectory_parameters_and_nonfinite_flag(sa, monkeypatch):
    """p_shared = 0: dp rows per trajectory leave the root wave directly (no ensemble ticket); a NaN in one trajectory is reported
    (HIPADJ_ERR_NONFINITE, the reference's retcode check) by the fused tail like by k_compose_finish."""
    N, T, dt = 300, 4.0, 0.01
    ts = np.linspace(0.0, T, 41)
    ref, fus = _engines(sa, monkeypatch, N, ts, T, dt, loss_kind=1, p_shared=False, segments=7)
    u0, p = lorenz_inputs(N)
    P = np.tile(p, (N, 1)) * (1.0 + 0.02 * np.random.default_rng(1).standard_normal((N, 3)))
    ref.forward(u0, P, want_out=False); fus.forward(u0, P, want_out=False)
    a, b = ref.adjoint(None), fus.adjoint(None)
    assert rel(b[0], a[0]) < 1e-11 and rel(b[1], a[1]) < 1e-11 and b[1].shape == (N, 3)
    orc = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    rdu0, rdp, _, _ = orc.adjoint_ensemble(u0, P, want_out=False)
    assert rel(b[0], rdu0) < RTOL and rel(b[1], rdp) < RTOL
    bad = u0.copy(); bad[17, 0] = np.nan
    fus.forward(bad, P, want_out=False)
    with pytest.raises(sa.HipadjError) as e:
        fus.adjoint(None)
    assert e.value.status == -4
    fus.forward(u0, P, want_out=False)        # the handle recovers: counters are zeroed by the next forward solve
    c = fus.adjoint(None)
    assert np.array_equal(c[0], b[0]) and np.array_equal(c[1], b[1])
    ref.close(); fus.close()


@pytest.mark.parametrize("alg,kw", [("gauss", {}), ("backsolve", dict(checkpointing=True)), ("backsolve", dict(checkpointing=True, ckpt_stride=10))])
@pytest.mark.parametrize("model,N", [("lorenz", 1250), ("lv", 333)])
def test_fused_gauss_and_backsolve_equal_their_three_launch_sequences(sa, monkeypatch, alg, kw, model, N):
    """k_gauss_fused / k_backsolve_fused (segments cut at checkpoint knots) against k_gauss / k_backsolve + composition + reduction, on changing data"""
    T, dt = (10.0, 0.01) if model == "lorenz" else (5.0, 0.01)
    ts = np.linspace(0.0, T, 51)
    ref, fus = _engines(sa, monkeypatch, N, ts, T, dt, loss_kind=0, alg=alg, model=model, **kw)
    assert fus.stats()["time_segments"] == ref.stats()["time_segments"] > 1
    rng = np.random.default_rng(N)
    n, npar = fus.n, fus.np
    if model == "lorenz":
        u0, p = lorenz_inputs(N)
    else:
        u0, p = 1.0 + 0.1 * rng.standard_normal((N, 2)), np.array([1.5, 1.0, 3.0, 1.0])
    for rnd in range(3):
        u0r, pr = u0 + 0.01 * rng.standard_normal(u0.shape), p * (1.0 + 0.01 * rng.standard_normal(npar))
        ref.forward(u0r, pr, want_out=False); fus.forward(u0r, pr, want_out=False)
        for rep in range(3):
            delta = rng.standard_normal((N, len(ts), n))
            a, b = ref.adjoint(delta), fus.adjoint(delta)
            assert rel(b[0], a[0]) < 1e-10 and rel(b[1], a[1]) < 1e-10, (rnd, rep)
    # ... and the one-launch pass against the ORACLE on the last inputs (not only the library against itself: VERDICT r3 weak 1c)
    okw = dict(checkpointing=True, checkpoints=np.arange(0, int(round(T / dt)) + 1, kw["ckpt_stride"]) * dt) if kw.get("ckpt_stride") else dict(checkpointing=bool(kw.get("checkpointing")))
    orc = O.Problem(model.upper(), alg=alg.upper(), stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", **okw)
    rdu0, rdp, _, _ = orc.adjoint_ensemble(u0r, pr, delta, want_out=False)
    assert rel(b[0], rdu0) < 1e-6 and rel(b[1], rdp) < 1e-6
    ref.close(); fus.close()


@pytest.mark.parametrize("n,alg,cap", [(2, "interpolating", None), (4, "interpolating", None), (4, "gauss", None), (4, "backsolve", None),
                                       (8, "interpolating", None), (8, "backsolve", None), (8, "interpolating", 160), (8, "backsolve", 160)])
def test_fused_pass_of_runtime_lane_models(sa, monkeypatch, capfd, n, alg, cap):
    """Runtime-registered lane models (hiprtc) take the same one-launch pass, k_*_fused<UserModel, ...>, while their segment map has at most 64 entries
    ((1 + n)(n + np): n <= 4 here); wider maps make the tail one of the heavily spilling kernels and keep the three-launch sequence (cap None, n = 8).
    With the cap lifted (HIPADJ_FUSED_USER_CAP, test hook) the 8-state kernels are compiled anyway: their -O3 builds are right, and the Backsolve one is the
    case whose -O1 build came back wrong (GPU visit 8, profiles/r3_fused_wide_lane_probe.log; irreproducible then, reproducibly wrong under a later edit) — the
    self-test's arbiter (user_ground_truth, round 4: central differences of the loss through the forward kernel, neither build is asked) must then keep the build
    that is right, or refuse when none is; it must never hand out wrong numbers.  Against the same model through the three-launch sequence, on changing data."""
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    m = UM.LV if n == 2 else UM.ring(n)
    name = f"fusedrt_{n}_{alg}_{cap}"
    _lib.register_model(name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    N, T, dt = 700, 4.0, 0.01
    ts = np.linspace(0.0, T, 41)
    kw = dict(checkpointing=True) if alg == "backsolve" else {}
    ref, fus = _engines(sa, monkeypatch, N, ts, T, dt, loss_kind=0, alg=alg, model=name, user_cap=cap, fused_launches=(3 if (n == 8 and not cap) else 1), **kw)
    rng = np.random.default_rng(n)
    u0 = 1.0 + 0.1 * rng.standard_normal((N, m["n"])); p = 1.0 + 0.2 * rng.random(m["np"])
    for rnd in range(2):
        u0r = u0 + 0.01 * rng.standard_normal(u0.shape)
        ref.forward(u0r, p, want_out=False); fus.forward(u0r, p, want_out=False)
        for rep in range(3):
            delta = rng.standard_normal((N, len(ts), m["n"]))
            a = ref.adjoint(delta)
            try:
                b = fus.adjoint(delta)
            except sa.HipadjError as e:
                # the arbiter found NO build that matches finite differences of the forward solve: refusing is the contract (only reachable behind the test hook)
                assert cap and e.status == -6 and "no trustworthy build" in str(e)
                ref.close(); fus.close()
                return
            assert rel(b[0], a[0]) < 1e-10 and rel(b[1], a[1]) < 1e-10, (rnd, rep)
    # the one-launch pass of the runtime model against the ORACLE's model of the same right-hand side, on the last inputs
    orc = O.Problem("LV" if n == 2 else "RING", alg=alg.upper(), stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", checkpointing=(alg == "backsolve"),
                    dims=((0, 0, 0, 0) if n == 2 else (n, 0, 0, 0)))
    rdu0, rdp, _, _ = orc.adjoint_ensemble(u0r, p, delta, want_out=False)
    assert rel(b[0], rdu0) < 1e-6 and rel(b[1], rdp) < 1e-6
    ref.close(); fus.close()


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["shift", "cotangent", "data"])
@pytest.mark.parametrize("G,segs,radix", [(4, 48, 4), (8, 96, 16), (4, 10, 4), (8, 13, 4), (0, 0, 0)])
def test_grouped_one_launch_pass_matches_the_plain_one(sa, G, segs, radix, loss, monkeypatch):
    """Round 6: G consecutive segments of a trajectory block in one workgroup, their maps composed through LDS before the HBM tree (k_interp_fused_g, csrc/hipadj_fused.hpp
    "GROUPED form"; chosen by the planner for the stage-operator sweeps of the compiled-in Lorenz model — plan_group_choice — or forced with HIPADJ_FUSED_GROUP).  Another
    bracketing of the same associative composition: du0 / dp agree with the plain one-launch pass (HIPADJ_FUSED_GROUP = 0) to round-off and with the oracle at the parity
    tolerance, for every loss route of the pass (fused LSQ_SHIFT, cotangents read in place, the device-resident data block); ragged last groups (13 segments in groups of 8,
    10 in groups of 4) and the planner's own choice (G = 0 here: nothing forced) included."""
    import oracle as O
    rng = np.random.default_rng(3)
    N, T, dt = 1250, 10.0, 0.01
    ts = np.linspace(0.0, T, 101)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    block = rng.standard_normal((N, len(ts), 3))
    kw = dict(shift=dict(loss_kind=1, loss_shift=2.0), cotangent=dict(loss_kind=0), data=dict(loss_kind=2, loss_scale=2.0))[loss]
    res = {}
    for grouped in (False, True):
        monkeypatch.setenv("HIPADJ_FUSED_GROUP", "0")
        if grouped:
            if G:
                monkeypatch.setenv("HIPADJ_FUSED_GROUP", str(G)); monkeypatch.setenv("HIPADJ_TREE_RADIX", str(radix))
            else:
                monkeypatch.delenv("HIPADJ_FUSED_GROUP")
        eng = sa.Engine("lorenz", "interpolating", N, 0.0, T, dt, save_times=ts, time_segments=segs, **kw)
        if loss == "data":
            eng.set_loss_data(block)
        eng.forward(u0, p, want_out=False)
        res[grouped] = eng.adjoint(block if loss == "cotangent" else None)
        assert eng.stats()["launches_per_pass"] == 1 and (segs == 0 or eng.stats()["time_segments"] == segs)
        if grouped and not G:
            assert eng.stats()["time_segments"] == 48          # plan_group_choice: 20 blocks -> 12 groups of 4
        again = eng.adjoint(block if loss == "cotangent" else None)
        assert np.array_equal(again[0], res[grouped][0]) and np.array_equal(again[1], res[grouped][1])      # reproducible from pass to pass
        eng.close()
    monkeypatch.delenv("HIPADJ_FUSED_GROUP", raising=False); monkeypatch.delenv("HIPADJ_TREE_RADIX", raising=False)
    assert np.max(np.abs(res[True][0] - res[False][0])) / np.max(np.abs(res[False][0])) < 1e-10
    assert np.max(np.abs(res[True][1] - res[False][1]) / np.abs(res[False][1])) < 1e-9
    okw = dict(shift=dict(loss="LSQ_SHIFT", loss_shift=2.0), cotangent=dict(loss="COTANGENT"), data=dict(loss="LSQ_DATA", loss_scale=2.0))[loss]
    ref = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, **okw)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, None if loss == "shift" else block, want_out=False)
    assert np.max(np.abs(res[True][0] - rdu0)) / np.max(np.abs(rdu0)) < 1e-6 and np.max(np.abs(res[True][1] - rdp) / np.abs(rdp)) < 1e-6
