"""The exponential stepper of the PDE family on the device (HIPADJ_STEPPER_ETDRK4_FIXED, csrc/hipadj_field_etd.hpp: ETDRK4 with the diffusion term exact in the DFT
basis, in-workgroup FFTs) against the oracle's restatement (ORC_STEPPER_ETDRK4) — forward solution, Interpolating-, Gauss- and QuadratureAdjoint, all grids, the loss kinds of
the family, spans across the forcing switch — and against scipy's Radau directly (tests/golden/bruss_etd.json)."""
import json
import os

import numpy as np
import pytest

import oracle as O
from test_gpu_parity import bruss_u0, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-9


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("quadrature", "QUADRATURE"), ("gauss", "GAUSS")])
@pytest.mark.parametrize("G,dt,t0,t1,N", [(8, 0.0125, 0.9, 1.4, 3), (16, 0.00625, 1.0, 1.2, 2), (32, 0.003125, 1.05, 1.15, 1)])
def test_etdrk4_lsq_matches_oracle_across_the_forcing_switch(sa, alg, oalg, G, dt, t0, t1, N):
    u0 = bruss_u0(G, N); p = np.array([3.4, 1.0, 10.0])
    S = int(round((t1 - t0) / dt))
    ts = t0 + dt * np.arange(0, S + 1, S // 4)
    dims = (G, 0, 0, 0)
    sens = sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10) if alg == "quadrature" else (sa.GaussAdjoint() if alg == "gauss" else sa.InterpolatingAdjoint())
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (t0, t1), p, dims), u0), sa.ETDRK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=sa.LsqShift(2.0))
    du0, dp = sa.adjoint_sensitivities(sol, sa.ETDRK4(), t=ts)
    ref = O.Problem("BRUSS", alg=oalg, stepper="ETDRK4", t0=t0, t1=t1, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0, dims=dims, quad_abstol=1e-10, quad_reltol=1e-10)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p)
    assert rel(sol.u, rout) < RTOL
    assert rel(du0, rdu0) < RTOL
    assert rel(dp, rdp) < RTOL
    sol.engine.close()


@pytest.mark.parametrize("alg,oalg", [("interpolating", "INTERPOLATING"), ("quadrature", "QUADRATURE"), ("gauss", "GAUSS")])
def test_etdrk4_cotangents_per_trajectory_parameters_and_data_loss(sa, alg, oalg):
    G, dt, t0, t1, N = 8, 0.0125, 0.0, 0.5, 4
    rng = np.random.default_rng(2)
    u0 = bruss_u0(G, N, seed=3); p = np.array([3.4, 1.0, 10.0]) * (1 + 0.02 * rng.standard_normal((N, 3)))
    ts = np.array([0.0, 0.25, 0.5])
    delta = rng.standard_normal((N, len(ts), 2 * G * G))
    dims = (G, 0, 0, 0)
    sens = sa.QuadratureAdjoint(abstol=1e-10, reltol=1e-10) if alg == "quadrature" else (sa.GaussAdjoint() if alg == "gauss" else sa.InterpolatingAdjoint())
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (t0, t1), p[0], dims), u0, p), sa.ETDRK4(), dt=dt, saveat=ts, sensealg=sens)
    du0, dp = sa.adjoint_sensitivities(sol, sa.ETDRK4(), t=ts, dgdu_discrete=delta)
    ref = O.Problem("BRUSS", alg=oalg, stepper="ETDRK4", t0=t0, t1=t1, dt=dt, save_times=ts, loss="COTANGENT", dims=dims, quad_abstol=1e-10, quad_reltol=1e-10)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(sol.u, rout) < RTOL and rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    sol.engine.close()
    # the device-resident data loss through the same sweeps
    data = 1.0 + 0.1 * rng.standard_normal((N, len(ts), 2 * G * G))
    loss = sa.LsqData(data, 2.0)
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (t0, t1), p[0], dims), u0, p), sa.ETDRK4(), dt=dt, saveat=ts, sensealg=sens, dgdu_discrete=loss)
    du0, dp = sa.adjoint_sensitivities(sol, sa.ETDRK4(), t=ts, dgdu_discrete=loss)
    ref = O.Problem("BRUSS", alg=oalg, stepper="ETDRK4", t0=t0, t1=t1, dt=dt, save_times=ts, loss="LSQ_DATA", loss_scale=2.0, dims=dims, quad_abstol=1e-10, quad_reltol=1e-10)
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, data)
    assert rel(du0, rdu0) < RTOL and rel(dp, rdp) < RTOL
    sol.engine.close()


def test_etdrk4_gradient_against_radau_directly(sa):
    """The device against numbers that owe nothing to the oracle: scipy Radau + central differences (tests/golden/make_bruss_etd.py), 1408 steps over (0, 2.2)."""
    Gd = json.load(open(os.path.join(ROOT, "tests", "golden", "bruss_etd.json")))
    G = Gd["G"]; u0 = np.array(Gd["u0"])[None, :]; p = np.array(Gd["p"]); ts = np.array(Gd["ts"])
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (0.0, Gd["t1"]), p, (G, 0, 0, 0)), u0), sa.ETDRK4(), dt=0.0015625, saveat=ts,
                   sensealg=sa.InterpolatingAdjoint(), dgdu_discrete=sa.LsqShift(1.0), save_start=False)
    du0, dp = sa.adjoint_sensitivities(sol, sa.ETDRK4(), t=ts)
    assert rel(sol.u[0], np.array(Gd["sol"])) < 1e-6
    idx = Gd["du0_index"]
    assert np.max(np.abs(du0[0][idx] - np.array(Gd["du0"])) / np.abs(np.array(Gd["du0"]))) < 5e-5
    e = np.abs(dp - np.array(Gd["dp"])) / np.abs(np.array(Gd["dp"]))
    assert e[0] < 1e-6 and e[1] < 2e-6 and e[2] < 1e-3
    sol.engine.close()


def test_etdrk4_documented_horizon_runs_and_the_pullback_is_linear(sa):
    """configs[4] over the span the reference documents, (0, 11.5) with loss times 0:0.5:11.5 (docs/src/examples/pde/brusselator.md:115): 7360 exponential steps
    instead of 460 000 explicit ones; the oracle at this size takes minutes, so the check here is the linearity of the pullback and finiteness — parity at 32 x 32 is the
    test above, the bench prints the timing."""
    G = 32; dt = 0.0015625; S = 7360
    u0 = bruss_u0(G, 1); p = np.array([3.4, 1.0, 10.0])
    ts = 0.5 * np.arange(0, 24)
    rng = np.random.default_rng(1)
    d1 = rng.standard_normal((1, len(ts), 2 * G * G)); d2 = rng.standard_normal((1, len(ts), 2 * G * G))
    sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("bruss", u0[0], (0.0, S * dt), p, (G, 0, 0, 0)), u0), sa.ETDRK4(), dt=dt, saveat=ts, sensealg=sa.InterpolatingAdjoint())
    assert np.all(np.isfinite(sol.u)) and np.max(np.abs(sol.u)) < 50.0
    a1, b1 = sa.adjoint_sensitivities(sol, sa.ETDRK4(), dgdu_discrete=d1)
    a2, b2 = sa.adjoint_sensitivities(sol, sa.ETDRK4(), dgdu_discrete=d2)
    a3, b3 = sa.adjoint_sensitivities(sol, sa.ETDRK4(), dgdu_discrete=d1 + 2.0 * d2)
    assert np.all(np.isfinite(a1)) and rel(a3, a1 + 2.0 * a2) < 1e-9 and rel(b3, b1 + 2.0 * b2) < 1e-7
    assert sol.engine.stats()["workspace_bytes"] < 1.0e9
    sol.engine.close()


def test_etdrk4_is_refused_outside_its_family(sa):
    with pytest.raises(sa.HipadjError):
        sa.Engine("lorenz", "interpolating", 4, 0.0, 1.0, 0.01, save_times=[1.0], stepper=2)
    with pytest.raises(sa.HipadjError):
        sa.Engine("bruss", "backsolve", 1, 0.0, 1.0, 0.0125, save_times=[1.0], dims=(8, 0, 0, 0), stepper=2)
