"""The four-lanes-per-trajectory bodies (csrc/hipadj_quad_ts5.hpp) on the HOST: four threads per quad in lockstep, DPP quad_perm as a barrier exchange
(tests/emu/quad_emu.cpp), against the oracle — what `-m gpu` checks through the C ABI, checked here without a GPU for the arithmetic and the lane protocol:
a quad whose lanes stopped making the same quad_perm calls would deadlock (the device's lanes share a program counter; these threads do not)."""
import numpy as np
import pytest

import emu as E
import oracle as O
import quad_emu as Q


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def _case(seed=3, N=3, T=2.0):
    rng = np.random.default_rng(seed)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.3 * rng.standard_normal((N, 3))
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.array([0.0, 0.13, 0.5, 0.77, 1.0, 1.9, T])          # interior loss times, off any grid
    return u0, p, ts, rng.standard_normal((N, len(ts), 3))


@pytest.mark.parametrize("loss", ["cotangent", "lsq"])
@pytest.mark.parametrize("alg", ["interpolating", "backsolve", "gauss"])
def test_quad_bodies_match_the_oracle_on_the_host(alg, loss):
    u0, p, ts, delta = _case()
    ck = alg == "backsolve"
    kw = dict(loss_kind=0) if loss == "cotangent" else dict(loss_kind=1, loss_shift=2.0)
    cfg = E.make_config("lorenz", alg, len(u0), 0.0, 2.0, 0.0, ts, checkpointing=ck, p_shared=True, stepper=1, abstol=1e-9, reltol=1e-9, max_steps=4000, **kw)
    du0, dp, out, ns = Q.forward_adjoint(cfg, u0, p, delta if loss == "cotangent" else None)
    ref = O.Problem("LORENZ", alg=alg.upper(), stepper="TSIT5", t0=0.0, t1=2.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, checkpointing=ck,
                    **(dict(loss="COTANGENT") if loss == "cotangent" else dict(loss="LSQ_SHIFT", loss_shift=2.0)))
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta if loss == "cotangent" else None)
    # component-form arithmetic (du_0 = y sigma - sigma x, fused multiply-adds as written) against the oracle's plain form: equal step sequences up to the rounding of
    # the step-size factor; Lorenz amplifies those ulps over T = 2
    assert rel(out, rout) < 1e-10 and rel(du0, rdu0) < 1e-8 and rel(dp, rdp) < 1e-8
    assert ns.min() > 20


def test_one_component_gauss_instantiation_on_the_host():
    """hipadj_quad_ts5.hpp QuadNZ: on the DEVICE the one-component instantiation of the Gauss sweep returned a wrong lam (error growing with the step count on problems with
    interior loss times), so the product carries a dummy second component.  The same source compiled for the host with one component (-DHIPADJ_QUAD_GAUSS_NZ=1) agrees
    with the oracle and with the two-component build: the source is right, the defect is in the device code generation of that instantiation."""
    u0, p, ts, delta = _case(seed=5, N=2)
    cfg = E.make_config("lorenz", "gauss", len(u0), 0.0, 2.0, 0.0, ts, loss_kind=0, p_shared=True, stepper=1, abstol=1e-11, reltol=1e-11, max_steps=8000)
    a = Q.forward_adjoint(cfg, u0, p, delta, gauss_nz=2)
    b = Q.forward_adjoint(cfg, u0, p, delta, gauss_nz=1)
    ref = O.Problem("LORENZ", alg="GAUSS", stepper="TSIT5", t0=0.0, t1=2.0, dt=0.0, abstol=1e-11, reltol=1e-11, save_times=ts, loss="COTANGENT")
    rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, p, delta)
    print("nz2 vs oracle", rel(a[0], rdu0), rel(a[1], rdp), " nz1 vs oracle", rel(b[0], rdu0), rel(b[1], rdp), " nz1 vs nz2", rel(b[0], a[0]), rel(b[1], a[1]))
    assert rel(a[0], rdu0) < 1e-9 and rel(a[1], rdp) < 1e-9
    assert rel(b[0], rdu0) < 1e-9 and rel(b[1], rdp) < 1e-9


def test_fixed_step_forward_quad_on_the_host():
    """forward_quad_ev (hipadj_quad.hpp: the forward solve of the headline workload, one state component per lane of a quad, event knots for the save times): knots (u_k, f(u_k)),
    sol(ts) and y(T) against the oracle's RK4 solve and its right-hand side.  The component form is not expression-for-expression the plain form (du_0 = y sigma - sigma x,
    fused as written), so agreement is at roundoff amplified by Lorenz over T = 2, not bit for bit."""
    rng = np.random.default_rng(8)
    N, T, dt = 3, 2.0, 0.01
    u0 = np.array([1.0, 0.0, 0.0]) + 0.3 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.linspace(0.0, T, 21)
    cfg = E.make_config("lorenz", "interpolating", N, 0.0, T, dt, ts, loss_kind=1, loss_shift=2.0, p_shared=True)
    knots, out, yT = Q.forward_rk4(cfg, u0, p)
    ref = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=T, dt=dt, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    _, _, rout, _ = ref.adjoint_ensemble(u0, p)
    assert rel(out, rout) < 1e-11 and rel(yT, rout[:, -1, :]) < 1e-11
    assert np.array_equal(knots[:, 0, 0, :], u0) and rel(knots[:, ::10, 0, :], rout) < 1e-11          # the knots on the save grid are sol(ts)
    for i in range(N):
        for k in (0, 57, 200):
            assert rel(knots[i, k, 1], O.model_f("LORENZ", knots[i, k, 0], p)) < 1e-14                  # the slope stored with a knot is f at that knot


@pytest.mark.parametrize("p_shared", [True, False])
@pytest.mark.parametrize("model,omodel", [("lv", "LV"), ("lvt", "LVT")])
@pytest.mark.parametrize("alg", ["interpolating", "backsolve", "gauss"])
def test_lotka_volterra_quad_bodies_match_the_oracle_on_the_host(alg, model, omodel, p_shared):
    """QuadAdjLV (n = 2, np = 4: two lanes carry y and lam, all four a parameter sum; operands are broadcasts of lanes 0 and 1), plain and time-dependent (`fb`,
    test/Core3/adjoint.jl:8-12), with shared and with per-trajectory parameters."""
    rng = np.random.default_rng(17)
    N, T = 3, 3.0
    u0 = np.array([1.0, 1.0]) + 0.2 * rng.standard_normal((N, 2))
    p = np.array([1.5, 1.0, 3.0, 1.0]) * (1 + 0.05 * rng.standard_normal((N, 4)))
    if p_shared:
        p = p[0]
    ts = np.array([0.0, 0.4, 1.0, 1.7, 2.5, T])
    delta = rng.standard_normal((N, len(ts), 2))
    ck = alg == "backsolve"
    cfg = E.make_config(model, alg, N, 0.0, T, 0.0, ts, loss_kind=0, checkpointing=ck, p_shared=p_shared, stepper=1, abstol=1e-10, reltol=1e-10, max_steps=4000)
    du0, dp, out, ns = Q.forward_adjoint(cfg, u0, p, delta)
    ref = O.Problem(omodel, alg=alg.upper(), stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=1e-10, reltol=1e-10, save_times=ts, checkpointing=ck, loss="COTANGENT")
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    assert rel(out, rout) < 1e-11 and rel(du0, rdu0) < 1e-9 and rel(dp, rdp) < 1e-9


@pytest.mark.parametrize("alg", ["interpolating", "backsolve", "gauss"])
def test_quad_bodies_no_start_and_per_trajectory_parameters_on_the_host(alg):
    """`no_start` (the loss has no term at t0: src/concrete_solve.jl:600-640 drops the first save time) and per-trajectory parameters through the quad bodies."""
    u0, p, ts, delta = _case(seed=11, N=3)
    rng = np.random.default_rng(2)
    pp = p * (1 + 0.02 * rng.standard_normal((len(u0), 3)))
    ck = alg == "backsolve"
    cfg = E.make_config("lorenz", alg, len(u0), 0.0, 2.0, 0.0, ts, loss_kind=0, checkpointing=ck, p_shared=False, stepper=1, abstol=1e-9, reltol=1e-9, max_steps=4000, no_start=True)
    du0, dp, out, _ = Q.forward_adjoint(cfg, u0, pp, delta)
    ref = O.Problem("LORENZ", alg=alg.upper(), stepper="TSIT5", t0=0.0, t1=2.0, dt=0.0, abstol=1e-9, reltol=1e-9, save_times=ts, checkpointing=ck, loss="COTANGENT", no_start=True)
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, pp, delta)
    assert rel(out, rout) < 1e-10 and rel(du0, rdu0) < 1e-8 and rel(dp, rdp) < 1e-8
