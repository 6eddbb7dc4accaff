"""The workgroup-per-trajectory family of runtime models (csrc/hipadj_wide.hpp; `-m gpu`): parity against the CPU oracle on the same
seeded inputs, rtol 1e-6 (Float64), for all four sensealgs.

  (i)   the reference's matrix-state problem, a 30 x 50 state with df[i, j] = p1 i + p2 j (test/Core5/size_handling_adjoint.jl:37-70; the
        reference asserts that every VJP backend gives the same dp) — also against the closed form: u(t) = u0 + t (p1 i + p2 j)
  (ii)  the 2 -> 50 -> 2 neural ODE of the reference's published benchmark (docs/src/Benchmark.md:62: Chain(x -> x.^3, Dense(2, 50, tanh), Dense(50, 2)),
        u0 = [2, 0], tspan = (0, 1.5), 30 loss times, loss = sum(abs2, data - pred)) as a runtime model (252 parameters)
  (iii) dense linear maps u' = A u with every entry of A a parameter: np = n^2 = 576 (gradient accumulator in LDS) and 10 000 (in HBM)
"""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import RTOL, rel, ALGS

pytestmark = pytest.mark.gpu


def _alg(sa, name):
    return dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(), gauss=sa.GaussAdjoint(), quadrature=sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10))[name]


def _run(sa, fun, oname, dims, u0, p, T, dt, ts, alg, oalg, delta_of_out, p_shared=True):
    """device gradient and oracle gradient for the loss whose cotangents are delta_of_out(out)"""
    N = len(u0)
    ens = sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p if p_shared else p[0]), u0, None if p_shared else p)
    sol = sa.solve(ens, sa.RK4(), dt=dt, saveat=ts, sensealg=_alg(sa, alg))
    ref = O.Problem(oname, alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", dims=dims, checkpointing=(oalg == "BACKSOLVE"),
                    quad_abstol=1e-10, quad_reltol=1e-10)
    out_ref = np.stack([ref.forward(u0[i], p if p_shared else p[i])[0] for i in range(N)])
    assert rel(sol.u, out_ref) < 1e-12
    delta = delta_of_out(sol.u)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    sol.engine.close()
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    return du0, dp, rdu0, rdp, sol.u


@pytest.mark.parametrize("alg,oalg", ALGS)
def test_reference_matrix_state_30x50(sa, alg, oalg):
    R, Cc, T, dt = 30, 50, 1.0, 0.01
    ts = np.linspace(0.0, T, 11)
    rng = np.random.default_rng(7)
    N = 3
    u0 = rng.standard_normal((N, R * Cc)); p = rng.random(2)
    fun = sa.WideDeviceFunction.index_affine(f"idxaff_{alg}", R, Cc)
    du0, dp, rdu0, rdp, out = _run(sa, fun, "IDXAFF", (R, Cc, 0, 0), u0, p, T, dt, ts, alg, oalg, lambda o: 2.0 * o)     # l = sum(abs2, sol)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    # closed form: u_c(t) = u0_c + t (p1 i + p2 j)  =>  dl/dp1 = sum_s sum_c 2 u_c(t_s) t_s i_c, dl/du0 = sum_s 2 u(t_s)
    ii = np.tile(np.arange(1, R + 1), Cc).astype(float); jj = np.repeat(np.arange(1, Cc + 1), R).astype(float)
    ex = np.zeros(2); exu = np.zeros_like(u0)
    for t in ts:
        u = u0 + t * (p[0] * ii + p[1] * jj)
        ex += [np.sum(2 * u * t * ii), np.sum(2 * u * t * jj)]; exu += 2 * u
    assert rel(dp, ex) < 1e-9 and rel(du0, exu) < 1e-9


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("N", [1, 5])
def test_benchmark_neural_ode_2_50_2(sa, alg, oalg, N):
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30)                       # range(tspan..., length = 30)
    dt = T / (29 * 8)                                  # every loss time on the step grid
    rng = np.random.default_rng(100)
    p = np.concatenate([rng.standard_normal(H * d) * np.sqrt(1.0 / d), np.zeros(H), rng.standard_normal(d * H) * np.sqrt(1.0 / H), np.zeros(d)]) * 0.5
    u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d)); u0[0] = [2.0, 0.0]
    data = rng.standard_normal((N, len(ts), d))
    fun = sa.WideDeviceFunction.dense_chain(f"node_{alg}_{N}", (d, H, d), input_power=3)
    du0, dp, rdu0, rdp, _ = _run(sa, fun, "MLP1", (d, H, 0, 0), u0, p, T, dt, ts, alg, oalg, lambda o: 2.0 * (o - data))
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL and dp.shape == (252,)


@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("n", [24, 100])
def test_dense_linear_map_every_entry_a_parameter(sa, alg, oalg, n):
    T, dt = 1.0, 0.02
    ts = np.linspace(0.0, T, 6)
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)
    p = A.flatten(order="F")
    N = 2
    u0 = rng.standard_normal((N, n))
    w = rng.standard_normal((N, len(ts), n))
    fun = sa.WideDeviceFunction.dense_linear(f"lin{n}_{alg}", n)
    du0, dp, rdu0, rdp, _ = _run(sa, fun, "DENSELIN", (n, 0, 0, 0), u0, p, T, dt, ts, alg, oalg, lambda o: w)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL and dp.shape == (n * n,)


def test_per_trajectory_parameters_and_lsq_loss(sa):
    """p_shared = 0 (dp rows per trajectory) and the in-kernel loss dgdu = u - shift on a wide model"""
    n, T, dt = 24, 1.0, 0.02
    ts = np.linspace(0.0, T, 6)
    rng = np.random.default_rng(3)
    N = 4
    P = np.stack([(rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F") for _ in range(N)])
    u0 = rng.standard_normal((N, n))
    fun = sa.WideDeviceFunction.dense_linear("lin24_rows", n)
    ens = sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), P[0]), u0, P)
    for alg, oalg in ALGS:
        sol = sa.solve(ens, sa.RK4(), dt=dt, saveat=ts, sensealg=_alg(sa, alg), dgdu_discrete=sa.LsqShift(0.3), want_out=False)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts)
        sol.engine.close()
        ref = O.Problem("DENSELIN", alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=0.3, dims=(n, 0, 0, 0),
                        checkpointing=(oalg == "BACKSOLVE"), quad_abstol=1e-10, quad_reltol=1e-10)
        rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, P)
        assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL and dp.shape == (N, n * n), alg


# ---- adaptive Tsit5 in the workgroup family (GaussAdjoint, InterpolatingAdjoint) ----------------------------------------------------------------
def _run_ts5(sa, fun, oname, dims, u0, p, T, ts, tol, delta_of_out, p_shared=True, loss=None, max_steps=0, alg="gauss"):
    N = len(u0)
    ens = sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p if p_shared else p[0]), u0, None if p_shared else p)
    kw = dict(dgdu_discrete=loss) if loss is not None else {}
    base = alg.split("_")[0]
    sens = dict(gauss=sa.GaussAdjoint(), interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(), backsolve_nockpt=sa.BacksolveAdjoint(checkpointing=False),
                quadrature=sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10), gauss_ckpt=sa.GaussAdjoint(checkpointing=True), interpolating_ckpt=sa.InterpolatingAdjoint(checkpointing=True),
                gausskronrod_ckpt=sa.GaussKronrodAdjoint(checkpointing=True))[alg]
    sol = sa.solve(ens, sa.Tsit5(), saveat=ts, sensealg=sens, abstol=tol[0], reltol=tol[1], max_steps=max_steps, **kw)
    ref = O.Problem(oname, alg={"gausskronrod": "GAUSS_KRONROD"}.get(base, base.upper()), stepper="TSIT5", checkpointing=(alg == "backsolve" or alg.endswith("_ckpt")), quad_abstol=1e-10, quad_reltol=1e-10, t0=0.0, t1=T, dt=0.0, abstol=tol[0], reltol=tol[1], save_times=ts, dims=dims,
                    **(dict(loss="COTANGENT") if loss is None else dict(loss="LSQ_SHIFT", loss_shift=loss.shift)))
    if loss is None:
        delta = delta_of_out(sol.u)
        du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=delta)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    else:
        du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p)
    out = sol.u
    sol.engine.close()
    return du0, dp, rdu0, rdp, out, rout


# the controller's norms are summed over the workgroup (per-thread partials, then a butterfly) instead of component by component: the error estimate differs
# from the oracle's in the last bits, the accepted step sizes with it — far below the tolerance asserted here
TS5_RTOL = 1e-8


@pytest.mark.parametrize("alg", ["gauss", "interpolating", "backsolve", "backsolve_nockpt", "quadrature"])
@pytest.mark.parametrize("tol", [(1e-6, 1e-3), (1e-9, 1e-9)])
@pytest.mark.parametrize("N", [1, 5])
def test_adaptive_tsit5_benchmark_neural_ode(sa, tol, N, alg):
    """docs/src/Benchmark.md:62 as published: Tsit5 with tolerances, 30 loss times — Gauss- and InterpolatingAdjoint on the adaptive solution of a wide runtime model"""
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30)
    rng = np.random.default_rng(100)
    p = np.concatenate([rng.standard_normal(H * d) * np.sqrt(1.0 / d), np.zeros(H), rng.standard_normal(d * H) * np.sqrt(1.0 / H), np.zeros(d)]) * 0.5
    u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d)); u0[0] = [2.0, 0.0]
    data = rng.standard_normal((N, len(ts), d))
    fun = sa.WideDeviceFunction.dense_chain(f"node_ts5_{N}_{tol[1]:.0e}_{alg}", (d, H, d), input_power=3)
    du0, dp, rdu0, rdp, out, rout = _run_ts5(sa, fun, "MLP1", (d, H, 0, 0), u0, p, T, ts, tol, lambda o: 2.0 * (o - data), alg=alg)
    assert rel(out, rout) < TS5_RTOL and rel(du0, rdu0) < TS5_RTOL and rel(dp, rdp) < TS5_RTOL and dp.shape == (252,)


@pytest.mark.parametrize("alg", ["gauss", "interpolating", "backsolve", "quadrature"])
def test_adaptive_tsit5_matrix_state_30x50(sa, alg):
    """(the two parameters of this model are the reduced kind: every component feeds them — Interpolating sums their stage values over the workgroup per stage)"""
    R, Cc, T = 30, 50, 1.0
    ts = np.linspace(0.0, T, 11)
    rng = np.random.default_rng(7)
    N = 3
    u0 = rng.standard_normal((N, R * Cc)); p = rng.random(2)
    fun = sa.WideDeviceFunction.index_affine(f"idxaff_ts5_{alg}", R, Cc)
    du0, dp, rdu0, rdp, out, rout = _run_ts5(sa, fun, "IDXAFF", (R, Cc, 0, 0), u0, p, T, ts, (1e-6, 1e-3), lambda o: 2.0 * o, alg=alg)
    assert rel(out, rout) < TS5_RTOL and rel(du0, rdu0) < TS5_RTOL and rel(dp, rdp) < TS5_RTOL


@pytest.mark.parametrize("n,alg", [(24, "gauss"), (100, "gauss"), (24, "interpolating"), (24, "backsolve"), (24, "quadrature"), (100, "quadrature")])
def test_adaptive_tsit5_dense_linear_rows_and_lsq(sa, n, alg):
    """per-trajectory parameters (np = n^2: 576 in LDS, 10 000 in HBM), the in-kernel loss dgdu = u - shift, loss times that are not step boundaries"""
    T = 1.0
    ts = np.array([0.0, 0.13, 0.37, 0.5, 0.81, 1.0])
    rng = np.random.default_rng(n)
    N = 3
    P = np.stack([(rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F") for _ in range(N)])
    u0 = rng.standard_normal((N, n))
    fun = sa.WideDeviceFunction.dense_linear(f"lin{n}_ts5_{alg}", n)
    du0, dp, rdu0, rdp, out, rout = _run_ts5(sa, fun, "DENSELIN", (n, 0, 0, 0), u0, P, T, ts, (1e-8, 1e-6), None, p_shared=False, loss=sa.LsqShift(0.3), alg=alg)
    assert rel(out, rout) < TS5_RTOL and rel(du0, rdu0) < TS5_RTOL and rel(dp, rdp) < TS5_RTOL and dp.shape == (N, n * n)


@pytest.mark.parametrize("alg", ["gauss_ckpt", "interpolating_ckpt", "gausskronrod_ckpt"])
@pytest.mark.parametrize("model", ["chain", "idxaff", "linear"])
def test_adaptive_tsit5_checkpointing_on_wide_models(sa, model, alg):
    """checkpointing = true for Interpolating / Gauss / GaussKronrod on the ADAPTIVE solution of a wide model (VERDICT r4 missing 5, coverage row a5; src/interpolating_adjoint.jl:54-109,
    207-277): the forward solve keeps the states at the checkpoint times only (t0, the loss times, T), the sweep re-solves one interval at a time into a per-trajectory record —
    with the forward tolerances and the last step of the interval above as the first guess — exactly the oracle's (and the lane family's) scheme; the workspace shrinks with it."""
    rng = np.random.default_rng(53)
    N, T = 3, 1.2
    ts = np.array([0.0, 0.2, 0.45, 0.7, 0.95, 1.2])
    if model == "chain":
        if alg == "gausskronrod_ckpt":
            pytest.skip("the oracle's GK restatement holds its np-vectors on the stack (ORC_MAXNP_COST = 64); the 2-parameter and the 36-parameter model cover the sensealg")
        fun, omodel, dims, n = sa.WideDeviceFunction.dense_chain("ckts5_chain_" + alg, (2, 50, 2), input_power=3), "MLP1", (2, 50, 0, 0), 2
        p = rng.standard_normal(252) * 0.3; u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, 2))
    elif model == "idxaff":
        fun, omodel, dims, n = sa.WideDeviceFunction.index_affine("ckts5_idx_" + alg, 30, 50), "IDXAFF", (30, 50, 0, 0), 1500
        p = rng.random(2); u0 = rng.standard_normal((N, n))
    else:
        n = 6
        fun, omodel, dims = sa.WideDeviceFunction.dense_linear("ckts5_lin_" + alg, n), "DENSELIN", (n, 0, 0, 0)
        p = (rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F"); u0 = rng.standard_normal((N, n))
    data = rng.standard_normal((N, len(ts), n))
    du0, dp, rdu0, rdp, out, rout = _run_ts5(sa, fun, omodel, dims, u0, p, T, ts, (1e-8, 1e-6), lambda o: 2.0 * (o - data), alg=alg)
    assert rel(out, rout) < TS5_RTOL and rel(du0, rdu0) < TS5_RTOL and rel(dp, rdp) < TS5_RTOL
    # against the dense sweep of the same handle family: the same gradient up to the solver tolerance, from a smaller workspace
    ens = sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0, None)
    wsb, grads = [], []
    for ck in (False, True):
        sens = {"gauss": sa.GaussAdjoint, "interpolating": sa.InterpolatingAdjoint, "gausskronrod": sa.GaussKronrodAdjoint}[alg.split("_")[0]](checkpointing=ck)
        sol = sa.solve(ens, sa.Tsit5(), saveat=ts, sensealg=sens, abstol=1e-8, reltol=1e-6)
        grads.append(sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=2.0 * (sol.u - data)))
        wsb.append(sol.engine.stats()["workspace_bytes"]); sol.engine.close()
    assert wsb[1] < wsb[0] and rel(grads[1][0], grads[0][0]) < 1e-4 and rel(grads[1][1], grads[0][1]) < 1e-4


def test_adaptive_tsit5_wide_reports_a_record_that_is_too_small(sa):
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30)
    rng = np.random.default_rng(1)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    u0 = np.array([[2.0, 0.0]])
    fun = sa.WideDeviceFunction.dense_chain("node_ts5_small", (d, H, d), input_power=3)
    with pytest.raises(Exception, match="max_steps"):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=sa.GaussAdjoint(), abstol=1e-10, reltol=1e-10, max_steps=3)
        sol.engine.synchronize()


@pytest.mark.parametrize("n,where", [(100, "five parameter-sized rows"), (60, "KB of LDS")])
def test_adaptive_interpolating_wide_refuses_what_does_not_fit_the_lds(sa, n, where):
    """np = n^2 = 10 000: refused by the planner; 3 600: by the LDS budget of the handle (5 np + the parameter copy > 160 KB) — both name GaussAdjoint"""
    rng = np.random.default_rng(n)
    u0 = rng.standard_normal((2, n)); p = (rng.standard_normal((n, n)) / np.sqrt(n)).flatten()
    fun = sa.WideDeviceFunction.dense_linear(f"lin{n}_ts5_refused", n)
    with pytest.raises(Exception, match=where) as ei:
        sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, 1.0), p), u0), sa.Tsit5(), saveat=np.linspace(0, 1, 5), sensealg=sa.InterpolatingAdjoint(), abstol=1e-6, reltol=1e-3)
    assert "GaussAdjoint" in str(ei.value)


# ---- continuous costs g(u, p, t) on wide models (the built-in kinds), both steppers, all four sensealgs -----------------------------------------
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("alg,oalg", ALGS)
@pytest.mark.parametrize("model,kind", [("linear", 2), ("index", 1), ("chain", 1), ("chain", 2)])
def test_continuous_costs_on_wide_models(sa, stepper, alg, oalg, model, kind):
    """g = u_1^2 + p_1 (test/Core7/mixed_costs.jl:46-57) and g = (sum u)^2 / 2 (test/Core3/adjoint.jl:913-919) accumulated inside the wide reverse kernels
    (WideWithCost), mixed with a discrete loss; dense linear map (owned parameters), 12 x 9 matrix state (reduced parameters), a 3-8-3 dense chain (59 parameters)."""
    rng = np.random.default_rng(31 + kind)
    T = 1.0
    if model == "linear":
        n = 8; fun = sa.WideDeviceFunction.dense_linear(f"cost_lin_{alg}_{stepper}", n); oname, dims = "DENSELIN", (n, 0, 0, 0)     # np = 64: the oracle's cost buffers hold 64 parameters
        p = (rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F")
    elif model == "index":
        R, Cc = 12, 9; n = R * Cc; fun = sa.WideDeviceFunction.index_affine(f"cost_idx_{alg}_{stepper}", R, Cc); oname, dims = "IDXAFF", (R, Cc, 0, 0)
        p = 0.2 * rng.random(2)
    else:
        n, H = 3, 8; fun = sa.WideDeviceFunction.dense_chain(f"cost_chain_{kind}_{alg}_{stepper}", (n, H, n), input_power=3); oname, dims = "MLP1", (n, H, 0, 0)   # the oracle's MLP1 cubes its input; np = 59
        p = np.concatenate([rng.standard_normal(H * n) * 0.4, 0.1 * rng.standard_normal(H), rng.standard_normal(n * H) * 0.3, 0.1 * rng.standard_normal(n)])
    N = 3
    u0 = 0.5 * rng.standard_normal((N, n))
    g = sa.FirstStateSquaredPlusFirstParam() if kind == 2 else sa.HalfSquaredSum()
    if stepper == "rk4":
        dt = 0.02; ts = np.linspace(0.0, T, 6); salg = sa.RK4(); kw = dict(dt=dt); okw = dict(stepper="RK4", dt=dt)
    else:
        ts = np.array([0.0, 0.21, 0.5, 0.77, 1.0]); salg = sa.Tsit5(); kw = dict(abstol=1e-9, reltol=1e-9); okw = dict(stepper="TSIT5", dt=0.0, abstol=1e-9, reltol=1e-9)
    sens = sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11) if alg == "quadrature" else _alg(sa, alg)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, dgdu_discrete=sa.LsqShift(0.3), g=g, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=sa.LsqShift(0.3), g=g)
    sol.engine.close()
    ref = O.Problem(oname, alg=oalg, t0=0.0, t1=T, save_times=ts, loss="LSQ_SHIFT", loss_shift=0.3, dims=dims, checkpointing=(oalg == "BACKSOLVE"),
                    quad_abstol=1e-11, quad_reltol=1e-11, cont_cost=kind, **okw)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p)
    tol = RTOL if stepper == "rk4" else 1e-7
    assert rel(du0, rdu0) < tol and rel(dp, rdp) < tol


def test_model_cost_selected_without_a_cost_is_refused(sa):
    fun = sa.WideDeviceFunction.dense_linear("cost_text_missing", 12)
    u0 = np.ones((2, 12)); p = np.zeros(144)
    with pytest.raises(Exception, match="has no cost"):
        sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, 1.0), p), u0), sa.RK4(), dt=0.1, saveat=np.linspace(0, 1, 3), sensealg=sa.InterpolatingAdjoint(), g=sa.ModelCost())


COST_BODIES = {   # the two costs the reference's tests use, as SPMD bodies of hipadj_wmodel_set_cost
    1: "double part = 0.0; HIPADJ_W_FOR(i, N) part += u[i]; const double s = wg_sum(part); HIPADJ_W_FOR(i, N) dlam[i] += s;",          # g = (sum u)^2 / 2
    2: "if (tid == 0) { dlam[0] += 2.0 * u[0]; if (WP) gp[0] += w; }",                                                                   # g = u_1^2 + p_1
}
COST_TRACED = {1: lambda u, p, t, ops: 0.5 * ops.sum(u) * ops.sum(u), 2: lambda u, p, t, ops: ops.sum(ops.gather(u, [0]) * ops.gather(u, [0])) + p[0]}


@pytest.mark.parametrize("how", ["text", "traced"])
@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_cost_attached_to_a_wide_model(sa, stepper, alg, oalg, kind, how):
    """dgdu_continuous / dgdp_continuous of a wide model (src/derivative_wrappers.jl:1411-1442; VERDICT r3 missing 5): the cost as one SPMD body (hipadj_wmodel_set_cost,
    cont_cost = HIPADJ_CCOST_MODEL), hand-written or traced by wtrace, against the oracle's restatement of the reference's two test costs (cont_cost 1 / 2) — and, on the
    fixed step, equal to the library's built-in cost kernels at 1e-12."""
    from test_wtrace import ring
    n, npar = 12, 13
    name = f"wcost_{how}_{kind}"
    if name not in _COSTFUN:
        if how == "text":
            _COSTFUN[name] = sa.WideDeviceFunction.from_callable(name, ring, n, npar).set_cost(body=COST_BODIES[kind])
        else:
            _COSTFUN[name] = sa.WideDeviceFunction.from_callable(name, ring, n, npar, cost=COST_TRACED[kind])
    fun = _COSTFUN[name]
    rng = np.random.default_rng(31)
    N, T, dt = 5, 0.8, 0.01
    ts = np.linspace(0.0, T, 5)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.2, 0.9, npar)
    delta = rng.standard_normal((N, len(ts), n))
    sens = dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(checkpointing=True), gauss=sa.GaussAdjoint(),
                quadrature=sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12))[alg]
    salg, kw, okw = (sa.RK4(), dict(dt=dt), dict(stepper="RK4", dt=dt)) if stepper == "rk4" else (sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, g=sa.ModelCost(), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta, g=sa.ModelCost())
    sol.engine.close()
    ref = O.Problem("RING", alg=oalg, t0=0.0, t1=T, save_times=ts, checkpointing=(alg == "backsolve"), dims=(n, 0, 0, 0), cont_cost=kind, quad_abstol=1e-12, quad_reltol=1e-12, **okw)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6
    if stepper == "rk4":
        g = [None, sa.HalfSquaredSum(), sa.FirstStateSquaredPlusFirstParam()][kind]
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, g=g, **kw)
        b0, b1 = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta, g=g)
        sol.engine.close()
        assert rel(du0, b0) < 1e-12 and rel(dp, b1) < 1e-12


_COSTFUN = {}


@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_device_gradients_against_independent_scipy_sensitivities(sa, stepper):
    """Not through the oracle: the HIP path vs tests/golden/wide_models.json (DOP853 forward sensitivities of numpy restatements written from the reference's
    definitions, make_wide_models.py) — the published neural ODE with Lux's parameter order and the matrix-state problem; rtol 1e-6 (north_star)."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide_models.json")) as f:
        G = json.load(f)
    g = G["node"]; d, H = g["dims"]; ts = np.asarray(g["ts"]); u0 = np.asarray(g["u0"])[None, :]; p = np.asarray(g["p"])
    fun = sa.WideDeviceFunction.dense_chain(f"golden_node_{stepper}", (d, H, d), input_power=3)
    salg, kw = (sa.RK4(), dict(dt=g["T"] / (29 * 32))) if stepper == "rk4" else (sa.Tsit5(), dict(abstol=1e-11, reltol=1e-11))
    for sens in (sa.InterpolatingAdjoint(), sa.GaussAdjoint(), sa.BacksolveAdjoint(), sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11)):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, g["T"]), p), u0), salg, saveat=ts, sensealg=sens, **kw)
        assert rel(sol.u[0], np.asarray(g["out"])) < 1e-8
        du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=2.0 * (sol.u - np.asarray(g["data"])[None]))
        sol.engine.close()
        assert rel(du0[0], g["du0"]) < 1e-6 and rel(dp, g["dp"]) < 1e-6, type(sens).__name__
    g = G["matrix"]; R, Cc = g["dims"]; ts = np.asarray(g["ts"]); u0 = np.asarray(g["u0"])[None, :]; p = np.asarray(g["p"])
    fun = sa.WideDeviceFunction.index_affine(f"golden_matrix_{stepper}", R, Cc)
    salg, kw = (sa.RK4(), dict(dt=0.01)) if stepper == "rk4" else (sa.Tsit5(), dict(abstol=1e-11, reltol=1e-11))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, g["T"]), p), u0), salg, saveat=ts, sensealg=sa.InterpolatingAdjoint(), **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=2.0 * sol.u)
    sol.engine.close()
    assert rel(du0[0], g["du0"]) < 1e-6 and rel(dp, g["dp"]) < 1e-6


# ---- GaussKronrodAdjoint on wide models: the step's quadrature is an adaptive (7,15) rule (wide_gk_panels), both steppers ----------------------------
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
@pytest.mark.parametrize("model,cost", [("linear", 0), ("index", 0), ("chain", 0), ("chain", 1), ("linear", 2)])
def test_gauss_kronrod_on_wide_models(sa, stepper, model, cost):
    """vs the oracle's GAUSS_KRONROD (its buffers hold np <= 64: an 8-state dense linear map, the 12 x 9 matrix state, a 3-8-3 chain); also close to GaussAdjoint,
    from which it differs by the quadrature rule's error only."""
    rng = np.random.default_rng(77 + cost)
    T = 1.0
    if model == "linear":
        n = 8; fun = sa.WideDeviceFunction.dense_linear(f"gk_lin_{stepper}_{cost}", n); oname, dims = "DENSELIN", (n, 0, 0, 0)
        p = (rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n)).flatten(order="F")
    elif model == "index":
        R, Cc = 12, 9; n = R * Cc; fun = sa.WideDeviceFunction.index_affine(f"gk_idx_{stepper}_{cost}", R, Cc); oname, dims = "IDXAFF", (R, Cc, 0, 0)
        p = 0.2 * rng.random(2)
    else:
        n, H = 3, 8; fun = sa.WideDeviceFunction.dense_chain(f"gk_chain_{stepper}_{cost}", (n, H, n), input_power=3); oname, dims = "MLP1", (n, H, 0, 0)
        p = np.concatenate([rng.standard_normal(H * n) * 0.4, 0.1 * rng.standard_normal(H), rng.standard_normal(n * H) * 0.3, 0.1 * rng.standard_normal(n)])
    N = 3
    u0 = 0.5 * rng.standard_normal((N, n))
    g = {0: None, 1: sa.HalfSquaredSum(), 2: sa.FirstStateSquaredPlusFirstParam()}[cost]
    skw = dict(g=g) if g is not None else {}
    if stepper == "rk4":
        dt = 0.02; ts = np.linspace(0.0, T, 6); salg = sa.RK4(); kw = dict(dt=dt); okw = dict(stepper="RK4", dt=dt)
    else:
        ts = np.array([0.0, 0.21, 0.5, 0.77, 1.0]); salg = sa.Tsit5(); kw = dict(abstol=1e-9, reltol=1e-9); okw = dict(stepper="TSIT5", dt=0.0, abstol=1e-9, reltol=1e-9)
    res = {}
    for name, sens in (("gk", sa.GaussKronrodAdjoint()), ("gauss", sa.GaussAdjoint())):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, dgdu_discrete=sa.LsqShift(0.3), **skw, **kw)
        res[name] = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=sa.LsqShift(0.3), **skw)
        sol.engine.close()
    ref = O.Problem(oname, alg="GAUSS_KRONROD", t0=0.0, t1=T, save_times=ts, loss="LSQ_SHIFT", loss_shift=0.3, dims=dims, cont_cost=cost, **okw)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p)
    tol = RTOL if stepper == "rk4" else 1e-7
    assert rel(res["gk"][0], rdu0) < tol and rel(res["gk"][1], rdp) < tol
    assert rel(res["gk"][1], res["gauss"][1]) < 1e-5


def test_checkpointed_sweep_equals_the_dense_one_bit_for_bit_over_many_inputs(sa):
    """The bit-for-bit statement of the test below is about TWO kernels (k_wide_adjoint, k_wide_adjoint_ck) that inline the same step code: it holds only while nothing in that
    code can be contracted in more than one way.  Round 4 found the Hermite midpoint 0.5 (u_lo + u_hi) + (dt / 8)(f_lo - f_hi) — a sum of two products — fused one way in one
    kernel and the other way in the other: 1-3 of 252 dp entries of the 2-50-2 chain differed in the last bit on ~15 % of random inputs (du0 never), and the single seed of the
    test below happened to be clean.  The midpoint is an explicit fma now; 24 inputs here."""
    fun = sa.WideDeviceFunction.dense_chain("ck_chain_seeds", (2, 50, 2), input_power=3)
    n, npar, N, T, dt = 2, 252, 4, 0.6, 0.01
    ts = np.array([0.0, 0.1, 0.25, 0.4, 0.6])
    for seed in range(20, 44):
        rng = np.random.default_rng(seed)
        u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(-0.4, 0.4, npar); delta = rng.standard_normal((N, len(ts), n))
        res = []
        for ck in (False, True):
            eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, dt, save_times=ts, checkpointing=ck, **(dict(ckpt_stride=7) if ck else {}))
            eng.forward(u0, p); res.append(eng.adjoint(delta)); eng.close()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), seed


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD")])
@pytest.mark.parametrize("model", ["idxaff", "chain", "linear"])
@pytest.mark.parametrize("ckpts", ["default", "stride7", "list"])
def test_checkpointing_on_wide_models_fixed_step(sa, alg, oalg, model, ckpts):
    """checkpointing = true for Interpolating / Gauss / GaussKronrod on a wide model (src/interpolating_adjoint.jl:54-109, 207-277): the forward solve keeps
    the checkpoint states only, the sweep re-solves interval by interval (k_wide_adjoint_ck).  On the fixed step the re-solved knots are the forward solve's own, so
    du0 / dp equal the dense sweep's BIT FOR BIT, and the oracle's checkpointed run at rtol 1e-6; the workspace shrinks."""
    rng = np.random.default_rng(29)
    if model == "idxaff":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.index_affine("ck_idx", 30, 50), "IDXAFF", (30, 50, 0, 0), 1500, 2
    elif model == "chain":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.dense_chain("ck_chain", (2, 50, 2), input_power=3), "MLP1", (2, 50, 0, 0), 2, 252
    else:
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.dense_linear("ck_lin", 12), "DENSELIN", (12, 0, 0, 0), 12, 144
    N, T, dt = 4, 0.6, 0.01
    ts = np.array([0.0, 0.1, 0.25, 0.4, 0.6]) if ckpts != "default" else np.linspace(0.0, T, 7)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(-0.4, 0.4, npar)
    delta = rng.standard_normal((N, len(ts), n))
    kw = dict(default={}, stride7=dict(ckpt_stride=7), list=dict(checkpoints=[0.07, 0.3, 0.31, 0.55]))[ckpts]
    res, wsb = [], []
    for ck in (False, True):
        eng = sa.Engine(fun.name, alg, N, 0.0, T, dt, save_times=ts, checkpointing=ck, **(kw if ck else {}))
        eng.forward(u0, p)
        res.append(eng.adjoint(delta))
        wsb.append(eng.stats()["workspace_bytes"])
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert wsb[1] < wsb[0]
    if alg == "gausskronrod" and npar > 64:
        return                      # the oracle's GK restatement holds its np-vectors on the stack (ORC_MAXNP_COST = 64); bit-equality with the dense sweep is the statement here
    okw = dict(checkpoints=kw["checkpoints"]) if ckpts == "list" else {}
    if ckpts == "stride7":
        okw = dict(checkpoints=[k * 7 * dt for k in range(0, 9)] + [T])
    ref = O.Problem(omodel, alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, checkpointing=True, dims=dims, **okw)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(res[1][0], rdu0) < 1e-6 and rel(res[1][1], rdp) < 1e-6


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("gausskronrod", "GAUSS_KRONROD")])
@pytest.mark.parametrize("model", ["idxaff", "chain", "ring"])
@pytest.mark.parametrize("no_start", [False, True])
def test_loss_times_off_the_step_grid_on_wide_models(sa, alg, oalg, model, no_start):
    """Loss times that are not multiples of dt on the fixed step (src/adjoint_common.jl:848-855: PresetTimeCallback makes the reverse solve stop on them):
    the sweep runs the planner's reverse step list with y(t) from the forward Hermite interpolant (k_wide_adjoint_og), out = sol(ts) by interpolation.  Against the oracle,
    whose generic integrator takes the same clipped steps."""
    from test_wtrace import ring
    rng = np.random.default_rng(37)
    if model == "idxaff":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.index_affine("og_idx", 30, 50), "IDXAFF", (30, 50, 0, 0), 1500, 2
    elif model == "chain":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.dense_chain("og_chain", (2, 50, 2), input_power=3), "MLP1", (2, 50, 0, 0), 2, 252
    else:
        if "og_ring" not in _COSTFUN:
            _COSTFUN["og_ring"] = sa.WideDeviceFunction.from_callable("og_ring", ring, 40, 41)
        fun, omodel, dims, n, npar = _COSTFUN["og_ring"], "RING", (40, 0, 0, 0), 40, 41
    if alg == "gausskronrod" and npar > 64:
        pytest.skip("the oracle's GK restatement holds its np-vectors on the stack (ORC_MAXNP_COST = 64); the 2- and the 41-parameter model cover the sensealg (round 5: k_wide_adjoint_og<., 4>)")
    N, T, dt = 4, 0.5, 0.01
    ts = np.array([0.0, 0.0333, 0.1, 0.2171, 0.455, 0.5]) if not no_start else np.array([0.0, 0.123, 0.3707])
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.2, 0.6, npar)
    delta = rng.standard_normal((N, len(ts), n))
    eng = sa.Engine(fun.name, alg, N, 0.0, T, dt, save_times=ts, no_start=no_start)
    out = eng.forward(u0, p)
    du0, dp = eng.adjoint(delta)
    eng.close()
    ref = O.Problem(omodel, alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, dims=dims, no_start=no_start)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(out, rout) < 1e-10 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6


@pytest.mark.parametrize("alg,oalg,ck", [("backsolve", "BACKSOLVE", True), ("backsolve", "BACKSOLVE", False), ("quadrature", "QUADRATURE", False)])
@pytest.mark.parametrize("model", ["idxaff", "chain", "ring"])
@pytest.mark.parametrize("no_start", [False, True])
def test_loss_times_off_the_step_grid_backsolve_and_quadrature_on_wide_models(sa, alg, oalg, ck, model, no_start):
    """The two sensealgs the off-grid sweeps of the wide family lacked (VERDICT r4 missing 5; src/adjoint_common.jl:848-855): BacksolveAdjoint over the reverse step list with the
    default checkpoints = t0, the save times, T interpolated from the forward knots (or none), and QuadratureAdjoint — the dense adjoint record on the non-uniform reverse grid and
    quadgk per loss interval over it.  Against the oracle's generic integrator on the same clipped steps."""
    from test_wtrace import ring
    rng = np.random.default_rng(41)
    if model == "idxaff":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.index_affine("ogq_idx", 30, 50), "IDXAFF", (30, 50, 0, 0), 1500, 2
    elif model == "chain":
        fun, omodel, dims, n, npar = sa.WideDeviceFunction.dense_chain("ogq_chain", (2, 50, 2), input_power=3), "MLP1", (2, 50, 0, 0), 2, 252
    else:
        if "og_ring" not in _COSTFUN:
            _COSTFUN["og_ring"] = sa.WideDeviceFunction.from_callable("og_ring", ring, 40, 41)
        fun, omodel, dims, n, npar = _COSTFUN["og_ring"], "RING", (40, 0, 0, 0), 40, 41
    N, T, dt = 4, 0.5, 0.01
    ts = np.array([0.0, 0.0333, 0.1, 0.2171, 0.455, 0.5]) if not no_start else np.array([0.0, 0.123, 0.3707])
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = rng.uniform(0.2, 0.6, npar)
    delta = rng.standard_normal((N, len(ts), n))
    eng = sa.Engine(fun.name, alg, N, 0.0, T, dt, save_times=ts, no_start=no_start, checkpointing=ck)
    out = eng.forward(u0, p)
    du0, dp = eng.adjoint(delta)
    du0b, dpb = eng.adjoint(delta)                                     # a second pass over the same forward solution: the records and cursors are re-entrant
    eng.close()
    assert np.array_equal(du0, du0b) and np.array_equal(dp, dpb)
    ref = O.Problem(omodel, alg=oalg, stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, dims=dims, no_start=no_start, checkpointing=ck)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(out, rout) < 1e-10 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
def test_adaptive_record_regrows_when_the_budget_cut_it_short(sa, monkeypatch, alg, oalg):
    """max_steps = 0 on a wide model: the dense record's capacity comes from a memory budget; when that leaves fewer steps than a trajectory takes the forward solve stores
    the true step counts, the host regrows the record (and Quadrature's adjoint record) and repeats the solve (wide_autosize) — HIPADJ_WIDE_REC_BUDGET makes the budget tiny here."""
    monkeypatch.setenv("HIPADJ_WIDE_REC_BUDGET", "4096")
    n = 12
    fun = sa.WideDeviceFunction.dense_linear("regrow_lin", n) if "regrow" not in _COSTFUN else _COSTFUN["regrow"]
    _COSTFUN["regrow"] = fun
    rng = np.random.default_rng(41)
    N, T = 5, 2.0
    ts = np.linspace(0.0, T, 7)
    A = np.zeros((n, n))
    for k in range(n // 2):                                  # six rotations, 5 (k + 1) rad per unit time: hundreds of accepted steps at 1e-10
        A[2 * k, 2 * k + 1], A[2 * k + 1, 2 * k] = 5.0 * (k + 1), -5.0 * (k + 1)
    p = A.flatten(order="F")
    u0 = rng.standard_normal((N, n))
    delta = rng.standard_normal((N, len(ts), n))
    eng = sa.Engine(fun.name, alg, N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=1e-10, reltol=1e-10, quad_abstol=1e-12, quad_reltol=1e-12)
    ws0 = eng.stats()["workspace_bytes"]
    eng.forward(u0, p)
    assert eng.stats()["workspace_bytes"] > ws0            # 64 steps did not hold the solution at 1e-10
    du0, dp = eng.adjoint(delta)
    eng.close()
    ref = O.Problem("DENSELIN", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, dims=(n, 0, 0, 0), quad_abstol=1e-12, quad_reltol=1e-12)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6


@pytest.mark.parametrize("alg", ["gauss", "interpolating", "quadrature"])
@pytest.mark.parametrize("H", [32, 64])
def test_dense_chain_2_H_H_2_is_routed_to_the_mfma_family(sa, alg, H):
    """Round 5: a weight-shared tanh chain 2 -> H -> H -> 2 over an ensemble is the FP64-MFMA family's model with the trajectories as batch columns;
    hipadj_create routes it there (csrc/hipadj_route.hpp, ABI 109: the chain is declared with hipadj_wmodel_declare_dense_chain) unless hipadj_config.family says "as registered" (mfma=False).  Both routes — the workgroup-per-trajectory kernels of the runtime model and the MFMA kernels of
    csrc/hipadj_mlp*.hpp — must give the same out, du0 and dp, for cotangents and for the device-resident data loss."""
    from test_gpu_parity import mlp_params
    rng = np.random.default_rng(5)
    N, d, T, dt = 32, 2, 0.3, 0.05
    fun = sa.WideDeviceFunction.dense_chain(f"route_chain_{H}", (d, H, H, d))
    u0 = rng.standard_normal((N, d)); p = mlp_params(d, H)
    ts = np.array([0.0, 0.1, 0.2, 0.3])
    delta = rng.standard_normal((N, len(ts), d))
    sens = {"gauss": sa.GaussAdjoint(), "interpolating": sa.InterpolatingAdjoint(), "quadrature": sa.QuadratureAdjoint(abstol=1e-11, reltol=1e-11)}[alg]
    res = {}
    for route in (False, None):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, mfma=route)
        assert sol.extra["mfma_routed"] == (route is None) and (sol.engine.stats()["routed_family"] == 3) == (route is None)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
        res[route] = (sol.u.copy(), du0, dp)
        sol.engine.close()
    for a, b in zip(res[False], res[None]):
        assert rel(b, a) < 1e-9
    data = 0.5 * rng.standard_normal((N, len(ts), d))
    res = {}
    for route in (False, None):
        rng0 = np.random.default_rng(23)
        loss = sa.LsqData(data, 2.0)
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=loss, mfma=route)
        dgdp = rng0.standard_normal((N, len(ts), len(p)))      # a loss with a direct parameter term: added to dp once (the routed handle has N = 1)
        du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss, dgdp_discrete=dgdp)
        res[route] = (du0, dp, np.array([sol.loss_value()]), np.array([np.sum((sol.u - data) ** 2)]))      # scale 2: the loss is (scale / 2) sum(abs2, sol .- data)
        with pytest.raises(ValueError):
            sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss, checkpoints=[0.0, 0.15, 0.3])
        sol.engine.close()
    assert rel(res[None][0], res[False][0]) < 1e-9 and rel(res[None][1], res[False][1]) < 1e-9
    assert rel(res[None][2], res[False][2]) < 1e-12 and rel(res[None][2], res[None][3]) < 1e-12


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
@pytest.mark.parametrize("stepper", ["rk4", "tsit5"])
def test_hand_written_wide_bodies_with_a_mass_matrix(sa, alg, oalg, stepper):
    """Round 5 (VERDICT r4 missing 5): ODEFunction(f; mass_matrix = M) for a wide model whose bodies are hand-written text (test/Core3/adjoint.jl:1315-1376 restated on a
    12-state dense linear map, np = 144): WideDeviceFunction.with_mass_matrix wraps the bodies — M^{-1} on the way out of f, M^{-T} lam on the way into the joint VJP — the
    device integrates nu = M' lam, the host maps du0; against the oracle, which carries M the reference's way."""
    n = 12
    rng = np.random.default_rng(17)
    M = np.eye(n) * 2.0 + 0.3 * rng.standard_normal((n, n))
    key = "hw_lin_mm"
    if key not in _MM:
        _MM[key] = sa.WideDeviceFunction.dense_linear("hw_lin12", n).with_mass_matrix(M)
    fun = _MM[key]
    N, T, dt = 4, 0.6, 0.01
    ts = np.linspace(0.0, T, 5)
    u0 = rng.uniform(0.3, 1.0, (N, n)); p = (-0.5 * np.eye(n) + 0.2 * rng.standard_normal((n, n))).ravel(order="F")
    delta = rng.standard_normal((N, len(ts), n))
    sens = dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(checkpointing=True), gauss=sa.GaussAdjoint(), quadrature=sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12))[alg]
    salg, kw, okw = (sa.RK4(), dict(dt=dt), dict(stepper="RK4", dt=dt)) if stepper == "rk4" else (sa.Tsit5(), dict(abstol=1e-10, reltol=1e-10), dict(stepper="TSIT5", dt=0.0, abstol=1e-10, reltol=1e-10))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), salg, saveat=ts, sensealg=sens, **kw)
    du0, dp = sa.adjoint_sensitivities(sol, salg, t=ts, dgdu_discrete=delta)
    sol.engine.close()
    with O.mass_matrix(M):
        ref = O.Problem("DENSELIN", alg=oalg, t0=0.0, t1=T, save_times=ts, checkpointing=(alg == "backsolve"), dims=(n, 0, 0, 0), quad_abstol=1e-12, quad_reltol=1e-12, **okw)
        rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(sol.u, rout) < 1e-6 and rel(du0, rdu0) < 1e-6 and rel(dp, rdp) < 1e-6


@pytest.mark.gpu
def test_routed_dense_chain_through_the_device_pointer_calls(sa):
    """The C ABI's device-pointer entry points on a handle the library routed to the FP64-MFMA family (hipadj_route.hpp): forward_dev / adjoint_dev / set_loss_data_dev /
    loss_value_dev take and return the ensemble's shapes ([N][M][d] blocks) — against the same model on the family it was registered for; the streaming-layout call is refused;
    a configuration the MFMA family does not take (loss times off the step grid, adaptive Tsit5, N not a multiple of 16) silently stays on the registered family."""
    import torch
    from test_gpu_parity import mlp_params
    rng = np.random.default_rng(9)
    N, d, H, T, dt = 48, 2, 32, 0.3, 0.05
    fun = sa.WideDeviceFunction.dense_chain("route_dev_chain", (d, H, H, d))
    u0 = rng.standard_normal((N, d)); p = mlp_params(d, H)
    ts = np.array([0.1, 0.2, 0.3]); delta = rng.standard_normal((N, len(ts), d)); data = rng.standard_normal((N, len(ts), d))
    dev = torch.device("cuda:0")
    res = {}
    for fam in (1, 0):
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, T, dt, save_times=ts, family=fam)
        assert (eng.stats()["routed_family"] == 3) == (fam == 0) and eng.n == d and eng.N == N
        tu0, tp, td = torch.tensor(u0, device=dev), torch.tensor(p, device=dev), torch.tensor(delta, device=dev)
        out = torch.empty((N, len(ts), d), dtype=torch.float64, device=dev); du0 = torch.empty((N, d), dtype=torch.float64, device=dev); dp = torch.empty(len(p), dtype=torch.float64, device=dev)
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, out); eng.adjoint_dev(td, du0, dp); eng.synchronize()
        res[fam] = [out.cpu().numpy(), du0.cpu().numpy(), dp.cpu().numpy()]
        if fam == 0:
            with pytest.raises(sa.HipadjError):
                eng.soa_stride()
        eng.close()
        eng = sa.Engine(fun.name, "gauss", N, 0.0, T, dt, save_times=ts, loss_kind=2, loss_scale=2.0, family=fam)
        eng.use_torch_stream()
        eng.set_loss_data_dev(torch.tensor(data, device=dev)); eng.forward_dev(tu0, tp, out); eng.adjoint_dev(None, du0, dp)
        lv = torch.empty(1, dtype=torch.float64, device=dev); eng.loss_value_dev(out, lv); eng.synchronize()
        res[fam] += [du0.cpu().numpy(), dp.cpu().numpy(), lv.cpu().numpy(), np.array([eng.loss_value(out.cpu().numpy())])]
        eng.close()
    for a, b in zip(res[1], res[0]):
        assert rel(b, a) < 1e-9
    assert rel(res[0][5], np.array([np.sum((res[0][0] - data) ** 2)])) < 1e-12
    for kw in (dict(save_times=np.array([0.07, 0.3])), dict(save_times=ts, stepper=1), dict(save_times=ts, ntraj=N - 8)):
        n_ = kw.pop("ntraj", N)
        eng = sa.Engine(fun.name, "interpolating", n_, 0.0, T, 0.0 if kw.get("stepper") else dt, **kw)
        assert eng.stats()["routed_family"] == 0
        eng.close()


_MM = {}
