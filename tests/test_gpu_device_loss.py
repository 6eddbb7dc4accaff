"""Discrete losses that stay ON THE DEVICE and ONE handle over several devices (`-m gpu`; ABI 108).

  * HIPADJ_LOSS_LSQ_DATA — dgdu_discrete = scale (u - data_i) with the data block resident in the handle: `sum(abs2, sol .- data)` (docs/src/Benchmark.md:80) without a
    cotangent block crossing the host link — on every kernel family, both steppers, all sensealgs, against the oracle and against the cotangent path;
  * HIPADJ_LOSS_MODEL — dgdu_discrete AND dgdp_discrete bodies attached to a runtime model (src/adjoint_common.jl:771-779; test/Core7/mixed_costs.jl:199-390) on the lane
    and the wide family, against the oracle's test losses and the scipy forward-sensitivity gradients of tests/golden/discrete_losses.json;
  * hipadj_loss_value, hipadj_adjoint_dev_soa;
  * hipadj_config.device_ids: virtual shards on device 0 against the single-device handle.
Tolerance: rtol 1e-6 against the oracle (Float64), tighter where two device paths do the same arithmetic."""
import json
import os

import numpy as np
import pytest

import oracle as O
import user_models as UM
from test_gpu_parity import RTOL, rel, ALGS, lorenz_inputs, sensealg_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_registered = {}

# l_i = (i + 1) p1 u1 u_n + sin(t_i) u1 + p2^2 d_i1 u_n: every argument of the reference's callback in use (oracle: dloss_id 4)
FULL_DGDU = "for (int j = 0; j < N; ++j) out[j] = 0.0; out[0] += (i + 1) * p[0] * u[N - 1] + sin(t); out[N - 1] += (i + 1) * p[0] * u[0] + p[1] * p[1] * d[0];"
FULL_DGDP = "for (int j = 0; j < NP; ++j) out[j] = 0.0; out[0] = (i + 1) * u[0] * u[N - 1]; out[1] = 2.0 * p[1] * d[0] * u[N - 1];"
FULL_L = "l = (i + 1) * p[0] * u[0] * u[N - 1] + sin(t) * u[0] + p[1] * p[1] * d[0] * u[N - 1];"


def _lv_with_loss(sa, how):
    name = f"lv_dloss_{how}"
    if name not in _registered:
        f = sa.DeviceFunction(name, 2, 4, UM.LV["f"], UM.LV["vjp"], UM.LV["vjp_p"])
        if how == "bodies":
            f.set_discrete_loss(dgdu=FULL_DGDU, dgdp=FULL_DGDP)
        elif how == "function":
            f.set_discrete_loss(l=FULL_L)
        elif how in ("traced", "traced_value"):      # the same loss as a host-language callable, traced once (trace.discrete_loss_bodies)
            from scimlsensitivity_jl_amd import trace
            f.set_discrete_loss(l=lambda u, p, t, i, d: (i + 1.0) * p[0] * u[0] * u[1] + trace.sin(t) * u[0] + p[1] ** 2 * d[0] * u[1], value=(how == "traced_value"))
        elif how == "u1sq_p1":      # test/Core7/mixed_costs.jl:199-227
            f.set_discrete_loss(dgdu="out[0] = 2.0 * u[0]; out[1] = 0.0;", dgdp="out[0] = 1.0; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0;")
        elif how == "lsq":          # sum(abs2, sol .- data) as a model body: must equal the built-in HIPADJ_LOSS_LSQ_DATA
            f.set_discrete_loss(dgdu="for (int j = 0; j < N; ++j) out[j] = 2.0 * (u[j] - d[j]);")
        _registered[name] = f
    return _registered[name]


@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lsq_data_loss_on_the_lane_family(sa, alg, oalg, stepper):
    """Lorenz ensemble, loss = sum(abs2, sol .- data): device-resident data block vs the oracle, vs the cotangent path with Delta = 2 (out - data), and the loss value."""
    N, T, dt = 130, 1.5, 0.01
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0, T, 16)
    rng = np.random.default_rng(5)
    data = rng.standard_normal((N, len(ts), 3))
    rk = stepper == "rk4"
    alg_obj = sa.RK4() if rk else sa.Tsit5()
    kw = dict(dt=dt) if rk else dict(abstol=1e-10, reltol=1e-10)
    sens = sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11) if alg == "quadrature" else sensealg_of(sa, alg)
    loss = sa.LsqData(data, 2.0)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), alg_obj, saveat=ts, sensealg=sens, dgdu_discrete=loss, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, alg_obj, t=ts, dgdu_discrete=loss)
    lv = sol.loss_value()
    out = sol.u.copy()
    sol.engine.close()
    assert abs(lv - np.sum((out - data) ** 2)) <= 1e-12 * abs(lv)
    ref = O.Problem("LORENZ", alg=oalg, stepper="RK4" if rk else "TSIT5", t0=0, t1=T, dt=dt if rk else 0.0, abstol=1e-10, reltol=1e-10, save_times=ts, loss="LSQ_DATA", loss_scale=2.0,
                    checkpointing=(alg == "backsolve"), quad_abstol=1e-11, quad_reltol=1e-11)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, data)
    tol = RTOL if rk else 1e-5      # adaptive: two implementations of one controller agree at the solver's tolerance, not at roundoff
    assert rel(out, rout) < tol and rel(du0, rdu0) < tol and rel(dp, rdp) < tol
    # the same gradient through the cotangent path (the AD route: Delta computed by the caller from `out`)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), alg_obj, saveat=ts, sensealg=sens, **kw)
    cdu0, cdp = sa.adjoint_sensitivities(sol, alg_obj, t=ts, dgdu_discrete=2.0 * (sol.u - data))
    sol.engine.close()
    assert rel(du0, cdu0) < 1e-11 and rel(dp, cdp) < 1e-11


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_lsq_data_loss_on_the_workgroup_families(sa, alg, oalg):
    """The wide family (2-50-2 neural ODE of docs/src/Benchmark.md:62-96, whose loss IS sum(abs2, ode_data .- pred)), the PDE family and the FP64-MFMA family: the data block in the
    cotangents' place against the cotangent path."""
    rng = np.random.default_rng(11)
    # wide: the published neural ODE
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30); dt = T / (29 * 4)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((4, d))
    data = rng.standard_normal((4, len(ts), d))
    name = f"node_ldata_{alg}"
    if name not in _registered:
        _registered[name] = sa.WideDeviceFunction.dense_chain(name, (d, H, d), input_power=3)
    fun = _registered[name]
    sens = sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11) if alg == "quadrature" else sensealg_of(sa, alg)
    res = []
    for loss in (sa.LsqData(data, 2.0), None):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=loss)
        res.append(sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss if loss is not None else 2.0 * (sol.u - data)))
        if loss is not None:
            assert abs(sol.loss_value() - np.sum((sol.u - data) ** 2)) <= 1e-12 * np.sum((sol.u - data) ** 2)
        sol.engine.close()
    assert rel(res[0][0], res[1][0]) < 1e-11 and rel(res[0][1], res[1][1]) < 1e-11
    ref = O.Problem("MLP1", alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, loss="LSQ_DATA", loss_scale=2.0, dims=(d, H, 0, 0), checkpointing=(alg == "backsolve"),
                    quad_abstol=1e-11, quad_reltol=1e-11)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, data)
    assert rel(res[0][0], rdu0) < RTOL and rel(res[0][1], rdp) < RTOL
    if alg == "backsolve":
        return          # the two bespoke families offer Interpolating / Gauss / Quadrature
    # PDE family (Brusselator 8 x 8) and MFMA family (MLP 2 -> 32 -> 32 -> 2, 16 columns)
    for model, dims, n, npar, T2, dt2 in (("bruss", (8, 0, 0, 0), 128, 3, 0.02, 2e-4), ("mlp", (2, 32, 16, 0), 32, None, 0.4, 0.01)):
        nn, npp = sa.model_sizes(model, dims)
        ts2 = np.linspace(0.0, T2, 5)
        u02 = rng.uniform(0.5, 1.5, (2, nn)); p2 = np.array([3.4, 1.0, 10.0]) if model == "bruss" else 0.3 * rng.standard_normal(npp)
        data2 = rng.standard_normal((2, len(ts2), nn))
        res = []
        for loss in (sa.LsqData(data2, 2.0), None):
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model, u02[0], (0.0, T2), p2, dims), u02), sa.RK4(), dt=dt2, saveat=ts2, sensealg=sens, dgdu_discrete=loss)
            res.append(sa.adjoint_sensitivities(sol, sa.RK4(), t=ts2, dgdu_discrete=loss if loss is not None else 2.0 * (sol.u - data2)))
            sol.engine.close()
        assert rel(res[0][0], res[1][0]) < 1e-11 and rel(res[0][1], res[1][1]) < 1e-11, model


@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("alg,oalg", ALGS + [("gausskronrod", "GAUSS_KRONROD")])
@pytest.mark.parametrize("how", ["bodies", "function", "traced", "traced_value"])
def test_model_discrete_loss_bodies_on_the_lane_family(sa, alg, oalg, stepper, how):
    """dgdu_discrete + dgdp_discrete as device bodies (or the loss itself, differentiated by dual numbers) inside the sweeps — every sensealg, both steppers — against the
    oracle's test loss 4 and, on the reference's own problem (test/Core7/mixed_costs.jl:10-16), against scipy forward sensitivities."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "discrete_losses.json")))
    ts = np.array(G["ts"]); p = np.array(G["p"])
    rng = np.random.default_rng(3)
    N = 70
    u0 = np.array(G["u0"]) + 0.05 * rng.standard_normal((N, 2)); u0[0] = G["u0"]
    data = rng.uniform(0.5, 2.0, (N, len(ts), 2)); data[0] = np.array(G["data"])
    rk = stepper == "rk4"
    alg_obj = sa.RK4() if rk else sa.Tsit5()
    kw = dict(dt=0.005) if rk else dict(abstol=1e-11, reltol=1e-11)
    sens = {"quadrature": sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12), "gausskronrod": sa.GaussKronrodAdjoint()}.get(alg) or sensealg_of(sa, alg)
    f = _lv_with_loss(sa, how)
    loss = sa.ModelLoss(data)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 10.0), p), u0), alg_obj, saveat=ts, sensealg=sens, dgdu_discrete=loss, save_start=False, save_end=False, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, alg_obj, t=ts, dgdu_discrete=loss)
    if how in ("function", "traced_value"):
        k = np.arange(1, len(ts) + 1)[None, :]
        want = np.sum(k * p[0] * sol.u[:, :, 0] * sol.u[:, :, 1] + np.sin(ts)[None, :] * sol.u[:, :, 0] + p[1] ** 2 * data[:, :, 0] * sol.u[:, :, 1])
        assert abs(sol.loss_value() - want) <= 1e-12 * abs(want)
    sol.engine.close()
    ref = O.Problem("LV", alg=oalg, stepper="RK4" if rk else "TSIT5", t0=0.0, t1=10.0, dt=0.005 if rk else 0.0, abstol=1e-11, reltol=1e-11, save_times=ts, loss="TEST", dloss_id=4,
                    checkpointing=(alg == "backsolve"), quad_abstol=1e-12, quad_reltol=1e-12)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, data)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    if not rk:      # trajectory 0 is the golden problem: the device against scipy directly, not only through the oracle
        g = G["losses"]["full"]
        assert rel(du0[0], g["du0"]) < 1e-7


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_reference_discrete_cost_with_dgdp(sa, alg, oalg):
    """test/Core7/mixed_costs.jl:199-390: cost = sum over the saving times of u1^2 + p1, dgdu = [2 u1, 0], dgdp = [1, 0, 0, 0], Tsit5 at abstol = reltol = 1e-12, against
    ForwardDiff there — against the scipy gradient here (the reference's `@test du0 ≈ ...` is rtol sqrt(eps))."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "discrete_losses.json")))
    ts = np.array(G["ts"]); p = np.array(G["p"]); u0 = np.array(G["u0"])[None]
    sens = sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12) if alg == "quadrature" else sensealg_of(sa, alg)
    f = _lv_with_loss(sa, "u1sq_p1")
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 10.0), p), u0), sa.Tsit5(), saveat=ts, sensealg=sens, dgdu_discrete=sa.ModelLoss(), abstol=1e-12, reltol=1e-12,
                   save_start=False, save_end=False)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=sa.ModelLoss())
    sol.engine.close()
    g = G["losses"]["u1sq_p1"]
    assert rel(du0[0], g["du0"]) < 1e-8 and rel(dp, g["dp"]) < 1e-8


def test_model_body_for_lsq_equals_the_builtin_kind(sa):
    """sum(abs2, sol .- data) written as a model body gives what HIPADJ_LOSS_LSQ_DATA gives (same sweep, two routes to the loss gradient), incl. checkpointing and off-grid times."""
    rng = np.random.default_rng(9)
    N, T, dt = 66, 2.0, 0.01
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0])
    for ts, sens in ((np.linspace(0, T, 11), sa.InterpolatingAdjoint(checkpointing=True)), (np.array([0.503, 1.0, 1.777, 2.0]), sa.InterpolatingAdjoint()),
                     (np.array([0.503, 1.0, 1.777, 2.0]), sa.GaussAdjoint()), (np.array([0.503, 1.0, 1.777, 2.0]), sa.BacksolveAdjoint())):
        data = rng.standard_normal((N, len(ts), 2))
        res = []
        for f, loss in ((_lv_with_loss(sa, "lsq"), sa.ModelLoss(data)), ("lv", sa.LsqData(data, 2.0))):
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=loss)
            res.append(sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss))
            sol.engine.close()
        assert rel(res[0][0], res[1][0]) < 1e-11 and rel(res[0][1], res[1][1]) < 1e-11


RING_DLOSS = ("if (tid == 0) { const double d0 = d ? d[0] : 0.0; dlam[0] += (i + 1) * p[0] * u[N - 1] + sin(t); dlam[N - 1] += (i + 1) * p[0] * u[0] + p[1] * p[1] * d0;"
              " if (WP) { gp[0] += (i + 1) * u[0] * u[N - 1]; gp[1] += 2.0 * p[1] * d0 * u[N - 1]; } }")


@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("alg,oalg", ALGS)
def test_model_discrete_loss_body_on_the_wide_family(sa, alg, oalg, stepper):
    """The same loss as one SPMD body of a wide model (a traced 40-state ring): dgdu into the vjp tile, dgdp into the gradient row — all four sensealgs, both steppers."""
    def ring(u, p, t, ops):
        n = u.length
        return p[0:n] * (ops.roll(u, -1) - u) + p[n] * ops.sin(ops.roll(u, 1))
    nr = 40
    name = "ring40_dloss"
    if name not in _registered:
        f = sa.WideDeviceFunction.from_callable(name, ring, nr, nr + 1)
        f.set_discrete_loss(body=RING_DLOSS)
        _registered[name] = f
    fun = _registered[name]
    rng = np.random.default_rng(21)
    N, T = 3, 0.6
    u0 = rng.uniform(0.3, 1.0, (N, nr)); p = rng.uniform(0.2, 0.6, nr + 1)
    ts = np.linspace(0.0, T, 7)
    data = rng.uniform(0.5, 2.0, (N, len(ts), nr))
    rk = stepper == "rk4"
    alg_obj = sa.RK4() if rk else sa.Tsit5()
    kw = dict(dt=0.01) if rk else dict(abstol=1e-10, reltol=1e-10)
    sens = sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11) if alg == "quadrature" else sensealg_of(sa, alg)
    loss = sa.ModelLoss(data)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), alg_obj, saveat=ts, sensealg=sens, dgdu_discrete=loss, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, alg_obj, t=ts, dgdu_discrete=loss)
    sol.engine.close()
    ref = O.Problem("RING", alg=oalg, stepper="RK4" if rk else "TSIT5", t0=0.0, t1=T, dt=0.01 if rk else 0.0, abstol=1e-10, reltol=1e-10, save_times=ts, loss="TEST", dloss_id=4,
                    dims=(nr, 0, 0, 0), checkpointing=(alg == "backsolve"), quad_abstol=1e-11, quad_reltol=1e-11)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, data)
    tol = RTOL if rk else 1e-5
    assert rel(du0, rdu0) < tol and rel(dp, rdp) < tol


def test_reference_literal_gauss_drops_dgdp_discrete(sa):
    """hipadj_config.reference_literal: the reference's GaussAdjoint never adds dgdp_discrete (src/adjoint_common.jl:776 `!isq`) — the switch reproduces that, device and oracle alike;
    the default keeps Gauss == Interpolating."""
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "discrete_losses.json")))
    ts = np.array(G["ts"]); p = np.array(G["p"]); u0 = np.array(G["u0"])[None]
    f = _lv_with_loss(sa, "u1sq_p1")
    out = {}
    for lit in (False, True):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 10.0), p), u0), sa.RK4(), dt=0.005, saveat=ts, sensealg=sa.GaussAdjoint(), dgdu_discrete=sa.ModelLoss(),
                       save_start=False, save_end=False, reference_literal=lit)
        out[lit] = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.ModelLoss())
        sol.engine.close()
        ref = O.Problem("LV", alg="GAUSS", stepper="RK4", t0=0.0, t1=10.0, dt=0.005, save_times=ts, loss="TEST", dloss_id=2, reference_literal=lit)
        rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, None)
        assert rel(out[lit][0], rdu0) < RTOL and rel(out[lit][1], rdp) < RTOL
    assert np.array_equal(out[True][0], out[False][0])
    assert abs((out[False][1] - out[True][1])[0] - len(ts)) < 1e-9 and np.max(np.abs((out[False][1] - out[True][1])[1:])) < 1e-12     # sum_i dgdp = M e_1


def test_cotangents_in_the_streaming_layout(sa):
    """hipadj_adjoint_dev_soa: Delta handed over as [M][n][ld] (ld = hipadj_soa_stride) gives the bits of hipadj_adjoint_dev with [N][M][n] — without the transposition launch."""
    import torch
    N, T, dt = 300, 1.0, 0.01
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0, T, 11)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint())
    eng = sol.engine
    dev = torch.device("cuda:0")
    delta = torch.randn((N, len(ts), 3), dtype=torch.float64, device=dev)
    du0a, dpa = (torch.empty((N, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev))
    du0b, dpb = torch.empty_like(du0a), torch.empty_like(dpa)
    eng.use_torch_stream()
    eng.adjoint_dev(delta, du0a, dpa)
    ld = eng.soa_stride()
    assert ld % 64 == 0 and ld >= N
    soa = torch.zeros((len(ts), 3, ld), dtype=torch.float64, device=dev)
    soa[:, :, :N] = delta.permute(1, 2, 0)
    eng.adjoint_dev_soa(soa, du0b, dpb)
    eng.synchronize()
    assert torch.equal(du0a, du0b) and torch.equal(dpa, dpb)
    eng.close()


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("G", [2, 8])
def test_one_handle_over_virtual_shards(sa, alg, oalg, G):
    """hipadj_config.device_ids with the same ordinal repeated: G shards on device 0.  With the time segmentation pinned, every trajectory's du0 and out are the
    bits of the single handle; dp is the shards' sum (1e-12).  Host-pointer and device-pointer entry points; shared and per-trajectory parameters; the device-resident loss."""
    import torch
    N, T, dt = 333, 1.0, 0.01
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0, T, 11)
    rng = np.random.default_rng(2)
    data = rng.standard_normal((N, len(ts), 3))
    sens = sensealg_of(sa, alg)
    res = []
    for devs in (None, [0] * G):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=sa.LsqData(data, 2.0), devices=devs,
                       time_segments=4)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqData(data, 2.0))
        res.append((sol.u.copy(), du0, dp, sol.loss_value()))
        if devs is not None:      # the device-pointer entry points of the multi handle (buffers of the primary device)
            dev = torch.device("cuda:0")
            eng = sol.engine
            tu0, tp = torch.as_tensor(u0, device=dev), torch.as_tensor(p, device=dev)
            tout = torch.empty((N, len(ts), 3), dtype=torch.float64, device=dev)
            tdu0, tdp, tl = torch.empty((N, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev), torch.empty(1, dtype=torch.float64, device=dev)
            eng.use_torch_stream()
            eng.set_loss_data_dev(torch.as_tensor(data, device=dev))
            eng.forward_dev(tu0, tp, tout)
            eng.adjoint_dev(None, tdu0, tdp)
            eng.loss_value_dev(tout, tl)
            eng.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(tout.cpu().numpy(), sol.u) and np.array_equal(tdu0.cpu().numpy(), du0) and np.array_equal(tdp.cpu().numpy(), dp)
            assert abs(tl.item() - res[-1][3]) <= 1e-13 * abs(tl.item())
            st = eng.stats()
            assert st["ntraj"] == N
        sol.engine.close()
    (o1, a1, b1, l1), (o2, a2, b2, l2) = res
    assert np.array_equal(o1, o2) and np.array_equal(a1, a2)
    assert rel(b2, b1) < 1e-12 and abs(l1 - l2) <= 1e-12 * abs(l1)
    # per-trajectory parameters: nothing is summed, everything is sliced
    pN = np.tile(p, (N, 1)) * (1.0 + 0.01 * rng.standard_normal((N, 1)))
    res = []
    for devs in (None, [0] * G):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0, pN), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=sa.LsqShift(2.0), devices=devs, time_segments=4)
        res.append(sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqShift(2.0)))
        sol.engine.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_multi_handle_refusals_and_wide_models(sa):
    """What a handle over several devices does not offer (RCCL communicators, the streaming cotangent layout) is refused by name; a wide runtime model shards like any other."""
    N, T, dt = 64, 1.0, 0.01
    u0, p = lorenz_inputs(N)
    ts = np.linspace(0, T, 6)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), devices=[0, 0])
    with pytest.raises(sa.HipadjError):
        sol.engine.comm_init_rank(b"\0" * 128, 1, 0)
    with pytest.raises(sa.HipadjError):
        sol.engine.soa_stride()
    sol.engine.close()
    with pytest.raises(sa.HipadjError):      # more shards than trajectories
        sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0[:2]), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), devices=[0, 0, 0])
    rng = np.random.default_rng(8)
    d, H = 2, 50
    pn = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    u0n = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((5, d))
    tsn = np.linspace(0.0, 1.5, 30); dtn = 1.5 / (29 * 4)
    delta = rng.standard_normal((5, len(tsn), d))
    if "node_multi" not in _registered:
        _registered["node_multi"] = sa.WideDeviceFunction.dense_chain("node_multi", (d, H, d), input_power=3)
    res = []
    for devs in (None, [0, 0, 0]):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(_registered["node_multi"], u0n[0], (0.0, 1.5), pn), u0n), sa.RK4(), dt=dtn, saveat=tsn, sensealg=sa.InterpolatingAdjoint(), devices=devs)
        res.append(sa.adjoint_sensitivities(sol, sa.RK4(), t=tsn, dgdu_discrete=delta))
        sol.engine.close()
    assert np.array_equal(res[0][0], res[1][0]) and rel(res[1][1], res[0][1]) < 1e-12


@pytest.mark.gpu
def test_host_pointer_calls_staged_and_direct_agree(sa):
    """Round 6: every transfer of the host-pointer calls goes through the handle's own registered staging block (an anonymous mapping with guard pages;
    csrc/hipadj_api.hip "host-pointer transfers") — HIPADJ_HOST_DIRECT=1 restores round 5's direct pageable copies.  Both modes, in fresh processes (the switch is read once):
    forward / adjoint / set_loss_data / loss_value on a plain handle, a handle over three virtual shards and a routed dense chain return the same bits; ragged sizes (a block of
    odd length, N not a multiple of anything) included."""
    import json, os, subprocess, sys
    code = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
import scimlsensitivity_jl_amd as sa
rng = np.random.default_rng(4)
out = {}
ts = np.linspace(0.0, 2.0, 21); p = np.array([10.0, 28.0, 8.0 / 3.0])
for tag, N, kw in (("plain", 777, {}), ("multi", 1001, dict(devices=[0, 0, 0]))):
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); delta = rng.standard_normal((N, len(ts), 3)); data = rng.standard_normal((N, len(ts), 3))
    eng = sa.Engine("lorenz", "interpolating", N, 0.0, 2.0, 0.01, save_times=ts, loss_kind=0, **kw)
    o = eng.forward(u0, p); du0, dp = eng.adjoint(delta); eng.close()
    eng = sa.Engine("lorenz", "gauss", N, 0.0, 2.0, 0.01, save_times=ts, loss_kind=2, loss_scale=2.0, **kw)
    eng.set_loss_data(data); o2 = eng.forward(u0, p); du1, dp1 = eng.adjoint(None); lv = eng.loss_value(o2); eng.close()
    out[tag] = [float(np.sum(o * np.arange(o.size).reshape(o.shape) %% 7)), float(du0.sum()), [float(x) for x in dp], float(du1.sum()), [float(x) for x in dp1], lv]
fun = sa.WideDeviceFunction.dense_chain("stage_chain", (2, 32, 32, 2))
N = 48; u0 = rng.standard_normal((N, 2)); pc = 0.3 * rng.standard_normal(fun.np); tsc = np.array([0.1, 0.2, 0.3]); delta = rng.standard_normal((N, 3, 2))
eng = sa.Engine(fun.name, "interpolating", N, 0.0, 0.3, 0.05, save_times=tsc)
assert eng.stats()["routed_family"] == 3
o = eng.forward(u0, pc); du0, dp = eng.adjoint(delta); eng.close()
out["routed"] = [float(o.sum()), float(du0.sum()), float(dp.sum())]
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("staged", "direct"):
        env = dict(os.environ)
        env.pop("HIPADJ_HOST_DIRECT", None)
        if mode == "direct":
            env["HIPADJ_HOST_DIRECT"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["staged"] == res["direct"]


def test_pipelined_host_transfers_from_concurrent_host_threads(sa):
    """Blocks of 4 MB and more cross the link in chunks with the CPU copies on a process-wide copier pool (hipadj_api.hip "Pipelined staging"): four host threads, each with its
    own handle and its own 12 MB cotangent / output blocks, drive hipadj_forward / hipadj_adjoint at once — results bit-identical to the serial run, and to HIPADJ_HOST_PIPELINE=0
    semantics (one block) by construction of the same kernels."""
    import threading
    N, T, dt = 5000, 10.0, 0.01
    ts = np.arange(0, T + 1e-9, 0.1)
    rng = np.random.default_rng(8)
    jobs = []
    for k in range(4):
        u0, p = lorenz_inputs(N, seed=20 + k)
        jobs.append((u0, p, rng.standard_normal((N, len(ts), 3))))

    def work(k, dst):
        u0, p, delta = jobs[k]
        eng = sa.Engine("lorenz", "interpolating", N, 0.0, T, dt, save_times=ts, loss_kind=0)
        for _ in range(3):
            out = eng.forward(u0, p, want_out=True)
            du0, dp = eng.adjoint(delta)
        dst[k] = (out.copy(), du0.copy(), dp.copy())
        eng.close()
    serial, threaded = {}, {}
    for k in range(4):
        work(k, serial)
    th = [threading.Thread(target=work, args=(k, threaded)) for k in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for k in range(4):
        for a, b in zip(serial[k], threaded[k]):
            assert np.array_equal(a, b)


def test_handles_created_on_a_loaded_device_see_their_data_block(sa):
    """hipadj_create's zero fills run on the null stream; before round 6's fix the fill of d_cotT could land AFTER hipadj_set_loss_data had transposed the data block into
    it on the handle's (non-blocking) stream — on a loaded device one handle in five computed the gradient of an all-zero data block (scripts/r6/create_race_inprocess.py with
    HIPADJ_CREATE_NO_DRAIN=1: 300 of 1500).  Three host threads keep the device busy while 500 small LSQ_DATA handles are created, used and closed: every gradient bit-identical."""
    import threading
    stop = threading.Event()

    def load(seed):
        u0, p = lorenz_inputs(10000, seed=seed)
        eng = sa.Engine("lorenz", "interpolating", 10000, 0.0, 10.0, 0.01, save_times=np.arange(0, 10.0 + 1e-9, 0.1), loss_kind=1, loss_shift=2.0)
        eng.forward(u0, p, want_out=False)
        while not stop.is_set():
            eng.adjoint(None)
        eng.close()
    th = [threading.Thread(target=load, args=(40 + k,)) for k in range(3)]
    [t.start() for t in th]
    rng = np.random.default_rng(9)
    N, T, dt = 66, 2.0, 0.01
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0]); ts = np.array([0.503, 1.0, 1.777, 2.0])
    data = rng.standard_normal((N, len(ts), 2))
    ref, wrong = None, 0
    try:
        for _ in range(500):
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lv", u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.GaussAdjoint(), dgdu_discrete=sa.LsqData(data, 2.0))
            du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqData(data, 2.0))
            sol.engine.close()
            if ref is None:
                ref = (du0.copy(), dp.copy())
            else:
                wrong += not (np.array_equal(du0, ref[0]) and np.array_equal(dp, ref[1]))
    finally:
        stop.set(); [t.join() for t in th]
    assert wrong == 0
    rdu0, rdp, _, _ = O.Problem("LV", alg="GAUSS", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_DATA", loss_scale=2.0).adjoint_ensemble(u0, p, data)
    assert rel(ref[0], rdu0) < RTOL and rel(ref[1], rdp) < RTOL
