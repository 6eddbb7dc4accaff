"""The exponential stepper of the PDE family (ORC_STEPPER_ETDRK4; HIPADJ_STEPPER_ETDRK4_FIXED, csrc/hipadj_field_etd.hpp) on the CPU side: the oracle's restatement of
ETDRK4 (Cox & Matthews 2002) against numbers that owe nothing to it — scipy's Radau on the Brusselator across the forcing switch (tests/golden/bruss_etd.json, made by
tests/golden/make_bruss_etd.py) — its order of convergence, the phi-functions, and the planner's rules for the new stepper."""
import json
import os

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "bruss_etd.json")))


def _problem(alg, dt, **kw):
    return O.Problem("BRUSS", alg=alg, stepper="ETDRK4", t0=0.0, t1=GOLD["t1"], dt=dt, save_times=np.array(GOLD["ts"]), dims=(GOLD["G"], 0, 0, 0),
                     loss="LSQ_SHIFT", loss_shift=1.0, quad_abstol=1e-11, quad_reltol=1e-11, **kw)


def test_forward_solution_converges_to_radau_at_high_order():
    u0 = np.array(GOLD["u0"]); p = np.array(GOLD["p"]); ref = np.array(GOLD["sol"])
    errs = []
    for dt in (0.0125, 0.00625, 0.003125):
        out, ns = _problem("INTERPOLATING", dt).forward(u0, p)
        assert ns == round(GOLD["t1"] / dt)
        errs.append(np.max(np.abs(out - ref)) / np.max(np.abs(ref)))
    assert errs[-1] < 5e-6
    # 4th order asymptotically; the non-smooth forcing patch and initial condition hold the observed order near 3 at these steps (stiff order reduction)
    assert errs[0] / errs[1] > 5.0 and errs[1] / errs[2] > 6.0


@pytest.mark.parametrize("alg", ["INTERPOLATING", "QUADRATURE", "GAUSS"])
def test_adjoint_gradient_converges_to_finite_differences_of_radau(alg):
    u0 = np.array(GOLD["u0"]); p = np.array(GOLD["p"])
    gdp = np.array(GOLD["dp"]); gdu = np.array(GOLD["du0"]); idx = GOLD["du0_index"]
    e_u, e_p = [], []
    for dt in (0.00625, 0.003125, 0.0015625):
        du0, dp, *_ = _problem(alg, dt).adjoint(u0, p)
        e_u.append(np.max(np.abs(du0[idx] - gdu) / np.abs(gdu)))
        e_p.append(np.abs(dp - gdp) / np.abs(gdp))
    assert e_u[-1] < 5e-5 and e_p[-1][0] < 1e-6 and e_p[-1][1] < 2e-6 and e_p[-1][2] < 1e-3
    assert e_u[0] / e_u[1] > 4.0 and e_u[1] / e_u[2] > 6.0          # the continuous adjoint integrated by the same scheme converges at its order


def test_step_on_a_component_without_linear_part_is_classic_rk4():
    """The gradient block of the Interpolating adjoint has M = 0: there ETDRK4 must be the classic RK4 (phi_k(0) = 1/k!) — with alpha = 0 the whole scheme is, and the
    exponential and the classic stepper of the oracle must agree to round-off."""
    u0 = np.array(GOLD["u0"]); p = np.array([3.4, 1.0, 0.0])
    ts = np.array([0.1, 0.2])
    kw = dict(t0=0.0, t1=0.2, dt=0.0125, save_times=ts, dims=(GOLD["G"], 0, 0, 0), loss="LSQ_SHIFT", loss_shift=1.0)
    a = O.Problem("BRUSS", alg="INTERPOLATING", stepper="ETDRK4", **kw).adjoint(u0, p)
    b = O.Problem("BRUSS", alg="INTERPOLATING", stepper="RK4", **kw).adjoint(u0, p)
    assert np.max(np.abs(a[0] - b[0])) <= 1e-12 * np.max(np.abs(b[0])) and np.max(np.abs(a[1][:2] - b[1][:2])) <= 1e-12 * np.max(np.abs(b[1]))


def test_exponential_stepper_is_refused_for_other_models():
    with pytest.raises(RuntimeError):
        O.Problem("LORENZ", alg="INTERPOLATING", stepper="ETDRK4", t0=0.0, t1=1.0, dt=0.01, save_times=np.array([1.0])).forward(np.ones(3), np.array([10.0, 28.0, 8 / 3]))


def test_planner_rules_for_the_exponential_stepper():
    """csrc/hipadj_plan.hpp through the host emulation's planner entry: the stepper is offered for the PDE family with Interpolating- / QuadratureAdjoint only."""
    import ctypes as C
    import emu as E

    def plan(model, alg, G):
        cfg = E.make_config(model, alg, 1, 0.0, 1.0, 0.0125, [0.5, 1.0], stepper=2)
        cfg.dims[0] = G
        nseg, nck, nq = C.c_int(), C.c_int(), C.c_int()
        b = (C.c_int * 64)()
        rc = E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq))
        return rc, E.lib().emu_last_error().decode()
    assert plan("bruss", "interpolating", 8)[0] == 0 and plan("bruss", "quadrature", 32)[0] == 0 and plan("bruss", "gauss", 16)[0] == 0
    rc, msg = plan("lorenz", "interpolating", 0)
    assert rc == -6 and "ETDRK4" in msg
