"""Which reference-held relation would notice a wrong [upstream-recall] constant?  

The oracle restates arithmetic of packages that are not vendored in the reference tree (OrdinaryDiffEq, DiffEqCallbacks, QuadGK).  Its constants are
data (oracle/adjoint_oracle.h: orc_test_set_recall); this file perturbs ONE at a time and evaluates the relations the REFERENCE's own tests hold for
this path, at the reference's tolerances:

  explicit   every sensealg == quadgk(lam' f_p) over a 1e-14 lambda solve, rtol 1e-9 (test/Core3/adjoint.jl:352-404; golden/explicit_integral.json)
  forwarddiff adjoint == ForwardDiff through the solver, rtol 1e-8 (test/Core3/adjoint.jl:691-705; golden/gradients.json, scipy standing in)
  literals   falling mass [-27.675, 0] atol 1e-2, exp.(p) rtol 1e-3 (test/Core7/physical_ode_regression.jl:42-51, test/Core1/sparse_adjoint.jl:32-33)

A perturbation that no relation notices is a constant whose value the restatement cannot defend: those rows head oracle/_ref/README.md as the first
fixtures to generate on a machine with Julia.  The expectations below are the measured outcome (deviation / tolerance), so the table stays true."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
R = dict(GAUSS_NODES_RK4=0, GAUSS_NODES_TSIT5=1, GK_TOL=2, QMAX=3, QMIN=4, GAMMA=5, BETA1=6, BETA2=7, PRESET_AT_INIT=8)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "explicit_integral.json")) as f:
        e = json.load(f)
    with open(os.path.join(HERE, "golden", "gradients.json")) as f:
        g = json.load(f)
    return e, g


def relations(gold, algs=("INTERPOLATING", "BACKSOLVE", "GAUSS", "GAUSS_KRONROD", "QUADRATURE")):
    """max over cases of deviation / tolerance, per relation and sensealg"""
    e, g = gold
    out = {}
    for alg in algs:
        worst = 0.0
        for case, model, loss in (("lvt", "LVT", "LSQ_SHIFT"), ("lorenz_T2", "LORENZ", "LSQ_SHIFT")):
            c = e[case]; ts = np.asarray(c["ts"])
            pr = O.Problem(model, alg=alg, checkpointing=(alg == "BACKSOLVE"), stepper="TSIT5", t0=c["tspan"][0], t1=c["tspan"][1], dt=0.0, abstol=1e-13, reltol=1e-13,
                           save_times=ts, loss=loss, loss_shift=2.0, quad_abstol=1e-13, quad_reltol=1e-12)
            du0, dp, _ = pr.adjoint(c["u0"], c["p"])
            tol = 1e-9 if alg != "BACKSOLVE" else 1e-7
            worst = max(worst, rel(dp, c["dp"]) / tol, rel(du0, c["du0"]) / tol)
        out[("explicit", alg)] = worst
        if alg != "GAUSS_KRONROD":
            c = g["lvt"]
            pr = O.Problem("LVT", alg=alg, stepper="TSIT5", t0=0, t1=10, dt=0.0, abstol=1e-12, reltol=1e-12, save_times=c["ts"], loss="LSQ_SHIFT", loss_shift=2.0,
                           quad_abstol=1e-12, quad_reltol=1e-12, checkpointing=(alg == "BACKSOLVE"))
            du0, dp, _ = pr.adjoint(c["u0"], c["p"])
            out[("forwarddiff", alg)] = max(rel(dp, c["dp"]), rel(du0, c["du0"])) / 1e-8
            c = g["lindiag"]
            pr = O.Problem("LINDIAG", alg=alg, stepper="TSIT5", t0=0, t1=1.0, dt=0.0, abstol=1e-6, reltol=1e-6, save_times=[1.0], loss="COTANGENT", quad_abstol=1e-6, quad_reltol=1e-6,
                           checkpointing=(alg == "BACKSOLVE"))
            _, dp, _ = pr.adjoint(c["u0"], c["p"], np.ones((1, 2)))
            out[("literals", alg)] = rel(dp, c["reference_literal"]) / 1e-3
    return out


PERTURBATIONS = [
    # (constant, perturbed value, relations that must notice (deviation > tolerance); empty = NOTHING the reference holds would notice)
    ("PRESET_AT_INIT", 0.0, {"explicit", "forwarddiff", "literals"}),
    ("GAUSS_NODES_TSIT5", 2.0, set()),        # measured: 0.02 x the tolerance of the tightest relation — at 1e-13 the steps are too short for the rule's order to show
    ("GAUSS_NODES_RK4", 3.0, set()),          # no reference test runs a fixed-step RK4 adjoint
    ("GK_TOL", 1e-3, set()),
    ("QMAX", 5.0, set()), ("QMIN", 0.5, set()), ("GAMMA", 0.8, set()), ("BETA1", 0.2, set()), ("BETA2", 0.0, set()),
]


def test_unperturbed_restatement_meets_every_relation(gold):
    O.lib().orc_test_set_recall(-1, C.c_double(0.0))
    r = relations(gold)
    assert max(r.values()) < 1.0, {k: v for k, v in r.items() if v >= 1.0}


@pytest.mark.parametrize("name,value,expect", PERTURBATIONS)
def test_which_relation_notices_a_perturbed_constant(gold, name, value, expect):
    L = O.lib()
    L.orc_test_set_recall.argtypes = [C.c_int, C.c_double]
    try:
        L.orc_test_set_recall(-1, 0.0)
        assert L.orc_test_set_recall(R[name], float(value)) == 0
        algs = ("GAUSS",) if name.startswith("GAUSS_NODES") else (("GAUSS_KRONROD",) if name == "GK_TOL" else ("INTERPOLATING", "BACKSOLVE", "GAUSS", "QUADRATURE"))
        r = relations(gold, algs)
    finally:
        L.orc_test_set_recall(-1, 0.0)
    noticed = {k[0] for k, v in r.items() if v > 1.0}
    print(f"[recall-sensitivity] {name} -> {value}: " + ", ".join(f"{k[0]}/{k[1]} {v:.2g}x tol" for k, v in sorted(r.items())))
    assert noticed == expect, (noticed, r)
