"""Parity at the sizes BASELINE.json's configs state (`-m gpu`): the HIP path through the C ABI vs the CPU oracle on the
same seeded inputs, including the reduced dL/dp — the one quantity that crosses GPUs.  Tolerance rtol = 1e-6 (Float64).

  C2  Lorenz-63, 10^4 trajectories x 1000 RK4 steps, InterpolatingAdjoint              du0 and dp over ALL trajectories
  C3  the same ensemble, BacksolveAdjoint(checkpointing=true), checkpoints every 10 steps  (test/Core3/adjoint.jl:1201-1241:
      Backsolve with checkpoints ~ Interpolating on Lorenz)                              du0 and dp over ALL trajectories
  C4  MLP 2 -> 128 -> 128 -> 2, 4096 columns, 150 RK4 steps, GaussAdjoint                du0 on a column sample, dp on a column chunk
  C5  Brusselator 32 x 32 (n = 2048), 400 RK4 steps, QuadratureAdjoint                  du0, dp, out

The oracle runs multi-threaded over trajectories (tests/oracle.py); the whole file takes well under a minute of host time."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import RTOL, rel, lorenz_inputs, mlp_params, bruss_u0

pytestmark = pytest.mark.gpu


def _c2_setup():
    N, T, dt = 10000, 10.0, 0.01
    u0, p = lorenz_inputs(N)            # seed 20240601: bench.py's ensemble
    return N, T, dt, u0, p, np.linspace(0.0, T, 101)


def test_config2_full_ensemble_du0_and_dp_vs_oracle(sa):
    """BASELINE configs[1] at size: every trajectory's du0 and the REDUCED dp against the oracle run on all 10^4 trajectories,
    for the fused LSQ loss (bench.py's workload) and for random cotangents (the AD path)."""
    N, T, dt, u0, p, ts = _c2_setup()
    prob = sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0)
    sol = sa.solve(prob, sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(2.0), want_out=False)
    assert sol.engine.stats()["time_segments"] > 1          # the time-segmented kernel is the one under test
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts)
    sol.engine.close()
    ref = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, want_out=False)
    assert rel(du0, rdu0) < RTOL
    assert np.max(np.abs(dp - rdp) / np.abs(rdp)) < RTOL     # component-wise: no component hides behind a larger one
    # cotangent path
    delta = np.random.default_rng(3).standard_normal((N, len(ts), 3))
    sol = sa.solve(prob, sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), want_out=False)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    sol.engine.close()
    ref = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT")
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta, want_out=False)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS")])
def test_config2_ensemble_on_adaptive_tsit5_at_size(sa, alg, oalg):
    """The C2 ensemble on the stepper of the reference's own tests (test/Core3/adjoint.jl:1157-1241 solves Lorenz with Tsit5): adaptive Tsit5 at 1e-9 / 1e-9, per-trajectory
    step control, the four-lanes-per-trajectory kernels (k_forward_tsit5_quad, k_adjoint_tsit5_quad: inlined one-log-one-exp controller, look-ahead cursor) — every
    trajectory's du0 and the reduced dp against the oracle's Tsit5 run on all 10^4 trajectories.  Device and oracle take the same step sequences up to the rounding of the
    step-size factor, so they agree far below their common discretisation error; Lorenz over T = 10 amplifies that rounding to a few 1e-9."""
    N, T, dt, u0, p, ts = _c2_setup()
    ck = alg == "backsolve"
    eng = sa.Engine("lorenz", alg, N, 0.0, T, 0.0, save_times=ts, loss_kind=1, loss_shift=2.0, p_shared=True, stepper=1, abstol=1e-9, reltol=1e-9, checkpointing=ck)
    eng.forward(u0, p, want_out=False)
    du0, dp = eng.adjoint(None)
    eng.close()
    ref = O.Problem("LORENZ", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, checkpointing=ck)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, want_out=False)
    assert rel(du0, rdu0) < RTOL
    assert np.max(np.abs(dp - rdp) / np.abs(rdp)) < RTOL


@pytest.mark.parametrize("shards", [1, 8])
def test_config3_backsolve_checkpointed_at_size(sa, shards):
    """BASELINE configs[2]: 10^4 x 1000 steps, BacksolveAdjoint(checkpointing=true), a checkpoint every 10 steps (= every loss time).
    shards = 8 runs the eight 1250-trajectory shards of the 8-GPU layout one after the other on this GPU and sums their dp on the
    host (the all-reduce): per-trajectory du0 and the reduced dp against the oracle over the whole ensemble; and, as the reference
    asserts on Lorenz (test/Core3/adjoint.jl:1201-1203, 1236-1241), Backsolve with checkpoints ~ Interpolating at rtol 1e-5... here
    on the reduced dp at 1e-4 (observed 6e-6; the backsolved states between checkpoints differ by the RK4 truncation error of a 10-step interval)."""
    N, T, dt, u0, p, ts = _c2_setup()
    du0 = np.empty((N, 3)); dp = np.zeros(3)
    for r in range(shards):
        lo, hi = sa.shard_range(N, r, shards)
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0[lo:hi]), sa.RK4(), dt=dt, saveat=ts,
                       sensealg=sa.BacksolveAdjoint(checkpointing=True), dgdu_discrete=sa.LsqShift(2.0), want_out=False)
        a, b = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts)
        du0[lo:hi] = a; dp += b
        sol.engine.close()
    ref = O.Problem("LORENZ", alg="BACKSOLVE", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, checkpointing=True)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, want_out=False)
    assert rel(du0, rdu0) < RTOL
    assert np.max(np.abs(dp - rdp) / np.abs(rdp)) < RTOL
    refi = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    _, idp, _, _ = refi.adjoint_ensemble(u0, p, want_out=False)
    assert rel(dp, idp) < 1e-4


def test_config4_mlp_gauss_150_steps(sa):
    """BASELINE configs[3] at its benchmarked length (150 RK4 steps, 30 loss times, 4096 columns): du0 of EVERY column and the full-width dp against the
    oracle.  The batch columns are independent 2-state ODEs that share the weights, so the oracle runs the 4096 columns as 256 "trajectories" of 16 columns
    with shared parameters, OpenMP over them (its dp is then the sum over all columns): about half a minute on the 16 cores of the GPU box
    (until round 3 the full-width dp was only compared with the sum of sixteen device runs)."""
    d, H, B, T, dt = 2, 128, 4096, 1.5, 0.01
    rng = np.random.default_rng(8)
    u0 = rng.standard_normal((1, d * B)); p = mlp_params(d, H)
    ts = np.linspace(0.05, T, 30)
    delta = rng.standard_normal((1, len(ts), d * B))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("mlp", u0[0], (0, T), p, (d, H, B, 0)), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.GaussAdjoint(), want_out=False)
    a_full, b_full = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=delta)
    sol.engine.close()
    C = 16                                                       # columns per oracle trajectory
    ref = O.Problem("MLP", alg="GAUSS", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="COTANGENT", dims=(d, H, C, 0))
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0[0].reshape(B // C, C * d), p,
                                           np.ascontiguousarray(delta[0].reshape(len(ts), B // C, C * d).transpose(1, 0, 2)), want_out=False)
    assert rel(a_full[0], rdu0.ravel()) < RTOL
    assert rel(b_full, rdp) < RTOL                               # all 33 410 parameters, all 4096 columns
    # the same gradient assembled from sixteen 256-column device runs (the layout a sharded batch would use)
    b_sum = np.zeros_like(b_full)
    for k in range(B // 256):
        sel = np.arange(256 * k * d, 256 * (k + 1) * d)
        s2 = sa.solve(sa.EnsembleProblem(sa.ODEProblem("mlp", u0[0, sel], (0, T), p, (d, H, 256, 0)), u0[:, sel]), sa.RK4(), dt=dt, saveat=ts,
                      sensealg=sa.GaussAdjoint(), want_out=False)
        b_sum += sa.adjoint_sensitivities(s2, sa.RK4(), t=ts, dgdu_discrete=delta[:, :, sel])[1]
        s2.engine.close()
    assert rel(b_full, b_sum) < 1e-9


def test_config5_brusselator_quadrature_400_steps(sa):
    """BASELINE configs[4]: 32 x 32 grid (n = 2048), QuadratureAdjoint, 400 explicit RK4 steps at the stability limit dt = 2.5e-5."""
    G, dt, t0, t1 = 32, 2.5e-5, 0.0, 0.01
    u0 = bruss_u0(G, 1); p = np.array([3.4, 1.0, 10.0])
    ts = np.array([0.0, 0.0025, 0.005, 0.0075, 0.01])
    dims = (G, 0, 0, 0)
    d1 = np.random.default_rng(1).standard_normal((1, len(ts), 2048))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (t0, t1), p, dims), u0), sa.RK4(), dt=dt, saveat=ts,
                   sensealg=sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10))
    a1, b1 = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=d1)
    ref = O.Problem("BRUSS", alg="QUADRATURE", stepper="RK4", t0=t0, t1=t1, dt=dt, save_times=ts, loss="COTANGENT", dims=dims,
                    quad_abstol=1e-10, quad_reltol=1e-10)
    rdu0, rdp, rout = ref.adjoint(u0[0], p, d1[0])
    assert rel(sol.u[0], rout) < RTOL and rel(a1[0], rdu0) < RTOL and rel(b1, rdp) < RTOL
    sol.engine.close()


def test_config5_documented_horizon(sa):
    """BASELINE configs[4] over the horizon the reference documents, tspan = (0, 11.5) with loss times 0:0.5:11.5
    (docs/src/examples/pde/brusselator.md:73-115; the docs solve it with the implicit FBDF because the system is stiff).  On an MI355X the
    explicit path reaches it as it stands: 460 000 RK4 steps of dt = 2.5e-5 (the diffusion stability limit), 14.7 GB of interpolant knots in
    the 288 GB of HBM, about a second per pass.  Checked: (1) the forward solution against an independent IMPLICIT solve (scipy BDF with the
    stencil sparsity, tests/golden/make_bruss_horizon.py); (2) dL/dp of InterpolatingAdjoint against central finite differences of the loss
    through the device forward solve; (3) dL/du0 against a finite difference along a random direction."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bruss_horizon.json")))
    G, dt, T = 32, 2.5e-5, 11.5
    ts = np.asarray(gold["ts"])
    xs = np.linspace(0.0, 1.0, G)
    U = 22.0 * (xs[None, :] * (1 - xs[None, :])) ** 1.5 * np.ones((G, 1)); V = 27.0 * (xs[:, None] * (1 - xs[:, None])) ** 1.5 * np.ones((1, G))
    u0 = np.concatenate([U.ravel(order="F"), V.ravel(order="F")])[None, :]
    p = np.asarray(gold["p"])
    dims = (G, 0, 0, 0)

    def forward(uu, pp, sens=None):
        return sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", uu[0], (0.0, T), pp, dims), uu), sa.RK4(), dt=dt, saveat=ts,
                        sensealg=sens or sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(2.0))

    loss = lambda sol: 0.5 * float(np.sum((sol.u - 2.0) ** 2))
    sol = forward(u0, p)
    assert sol.engine.stats()["nsteps"] == 460000
    got = sol.u[0][:, gold["sample_indices"]].T                       # [sample][time]
    assert rel(got, np.asarray(gold["u"])) < 1e-6                     # explicit RK4 at dt = 2.5e-5 vs implicit BDF at 1e-10
    assert rel(np.linalg.norm(sol.u[0], axis=1), np.asarray(gold["norm_per_time"])) < 1e-6      # observed 2e-7: the accuracy of the BDF solve
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts)
    sol.engine.close()
    fd = np.zeros(3)
    for j in range(3):
        h = 1e-6 * abs(p[j])
        e = np.zeros(3); e[j] = h
        sp = forward(u0, p + e); lp = loss(sp); sp.engine.close()
        sm = forward(u0, p - e); lm = loss(sm); sm.engine.close()
        fd[j] = (lp - lm) / (2 * h)
    assert np.max(np.abs(dp - fd) / np.abs(fd)) < 1e-5, (dp, fd)
    d = np.random.default_rng(4).standard_normal(u0.shape); d /= np.linalg.norm(d)
    h = 1e-4      # the loss is ~7e4 and a sum of 49 152 terms: its last digits (1e-11) bound the step from below
    sp = forward(u0 + h * d, p); lp = loss(sp); sp.engine.close()
    sm = forward(u0 - h * d, p); lm = loss(sm); sm.engine.close()
    assert abs(float(np.sum(du0 * d)) - (lp - lm) / (2 * h)) < 2e-5 * abs((lp - lm) / (2 * h)), (float(np.sum(du0 * d)), (lp - lm) / (2 * h))


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("backsolve", "BACKSOLVE"), ("gauss", "GAUSS"), ("quadrature", "QUADRATURE")])
def test_published_neural_ode_benchmark_4096_trajectories_adaptive(sa, alg, oalg):
    """bench.py's "AS PUBLISHED" rows at size: the 2-50-2 neural ODE of docs/src/Benchmark.md:62 as a wide runtime model, adaptive Tsit5 at the default
    tolerances, 4096 trajectories with their own step sequences — every du0 and the reduced dp against the oracle's Tsit5 on all of them."""
    d, H, T, N = 2, 50, 1.5, 4096
    ts = np.linspace(0.0, T, 30)
    rng = np.random.default_rng(11)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d))
    data = rng.standard_normal((N, len(ts), d))
    fun = sa.WideDeviceFunction.dense_chain(f"node_at_size_{alg}", (d, H, d), input_power=3)
    sens = dict(interpolating=sa.InterpolatingAdjoint(), backsolve=sa.BacksolveAdjoint(), gauss=sa.GaussAdjoint(), quadrature=sa.QuadratureAdjoint())[alg]
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(fun, u0[0], (0.0, T), p), u0), sa.Tsit5(), saveat=ts, sensealg=sens, abstol=1e-6, reltol=1e-3)
    delta = 2.0 * (sol.u - data)
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=ts, dgdu_discrete=delta)
    out = sol.u
    sol.engine.close()
    ref = O.Problem("MLP1", alg=oalg, stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-6, reltol=1e-3, save_times=ts, loss="COTANGENT", dims=(d, H, 0, 0),
                    checkpointing=(oalg == "BACKSOLVE"))
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(out, rout) < 1e-8 and rel(du0, rdu0) < 1e-7 and rel(dp, rdp) < 1e-7


def test_config2_device_resident_data_loss_and_eight_shards_in_one_handle_at_size(sa):
    """Round 5 at BASELINE configs[1]'s size: (i) the loss sum(abs2, sol .- data) evaluated inside the sweep (HIPADJ_LOSS_LSQ_DATA, the data block resident in the handle, no
    cotangents) — every trajectory's du0 and the reduced dp against the oracle on all 10^4 trajectories, and the device-side loss value; (ii) the same ensemble through ONE
    handle over eight (virtual) shards (hipadj_config.device_ids): du0 and dp equal to the single-device handle's at round-off (the shards choose their own time segmentation; the dp partials are summed in shard order)."""
    N, T, dt, u0, p, ts = _c2_setup()
    data = 1.0 + 0.5 * np.random.default_rng(11).standard_normal((N, len(ts), 3))
    loss = sa.LsqData(data, 2.0)
    prob = sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0, T), p), u0)
    sol = sa.solve(prob, sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=loss)
    du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=loss)
    lv = sol.loss_value()
    assert sol.engine.stats()["launches_per_pass"] == 1
    sol.engine.close()
    ref = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=T, dt=dt, save_times=ts, loss="LSQ_DATA", loss_scale=2.0)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, data)
    assert rel(du0, rdu0) < RTOL
    assert np.max(np.abs(dp - rdp) / np.abs(rdp)) < RTOL
    want = float(np.sum((rout - data) ** 2))
    assert abs(lv - want) <= 1e-9 * abs(want)
    sol8 = sa.solve(prob, sa.RK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=loss, devices=[0] * 8)
    du8, dp8 = sa.adjoint_sensitivities(sol8, sa.RK4(), t=ts, dgdu_discrete=loss)
    sol8.engine.close()
    assert rel(du8, du0) < 1e-11          # not bit-identical at this size: a 1250-trajectory shard runs 51 time segments, the 10^4 ensemble 13 (another association of the same maps)
    assert np.max(np.abs(dp8 - dp) / np.abs(dp)) < 1e-11


def test_config5_exponential_stepper_at_size_across_the_forcing_switch(sa):
    """configs[4]'s grid (32 x 32, n = 2048) on the stiff stepper of round 5 (HIPADJ_STEPPER_ETDRK4_FIXED) over (0, 2.2) — 1408 steps of dt = 1/640 across the switch of
    the forcing at t = 1.1 — QuadratureAdjoint (the config's sensealg) and InterpolatingAdjoint against the oracle's ETDRK4: out, du0, dp."""
    G, dt, S = 32, 0.0015625, 1408
    u0 = bruss_u0(G, 1); p = np.array([3.4, 1.0, 10.0])
    ts = np.array([0.55, 1.1, 1.65, 2.2])
    delta = np.random.default_rng(1).standard_normal((1, len(ts), 2 * G * G))
    for sens, oalg in ((sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10), "QUADRATURE"), (sa.InterpolatingAdjoint(), "INTERPOLATING")):
        sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (0.0, S * dt), p, (G, 0, 0, 0)), u0), sa.ETDRK4(), dt=dt, saveat=ts, sensealg=sens, save_start=False)
        du0, dp = sa.adjoint_sensitivities(sol, sa.ETDRK4(), t=ts, dgdu_discrete=delta)
        ref = O.Problem("BRUSS", alg=oalg, stepper="ETDRK4", t0=0.0, t1=S * dt, dt=dt, save_times=ts, loss="COTANGENT", dims=(G, 0, 0, 0), quad_abstol=1e-10, quad_reltol=1e-10)
        rdu0, rdp, rout = ref.adjoint(u0[0], p, delta[0])
        assert rel(sol.u[0], rout) < RTOL and rel(du0[0], rdu0) < RTOL and rel(dp, rdp) < RTOL, oalg
        sol.engine.close()
