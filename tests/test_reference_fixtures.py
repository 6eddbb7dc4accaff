"""Oracle vs outputs of THE REFERENCE ITSELF (tests/golden/reference_fixtures.json, written by oracle/_ref/make_fixtures.jl on a machine
with Julia).  The file cannot be generated in this image (no Julia): until it exists these tests are SKIPPED and the oracle's parity
status stays "unpinned" (DESIGN.md §5); once it exists every [upstream-recall] assumption listed there is checked by a real run."""
import json
import os

import numpy as np
import pytest

import oracle as O

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")
CASES = json.load(open(PATH))["cases"] if os.path.exists(PATH) else []


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.skipif(bool(CASES), reason="fixtures present")
def test_reference_fixtures_are_absent_parity_unpinned():
    """Marker test: documents in every test run that the reference's own outputs are not available here."""
    assert not os.path.exists(PATH)
    assert os.path.exists(os.path.join(os.path.dirname(PATH), "..", "..", "oracle", "_ref", "make_fixtures.jl"))


@pytest.mark.skipif(not CASES, reason="tests/golden/reference_fixtures.json absent: run oracle/_ref/make_fixtures.jl with Julia (parity unpinned)")
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_the_reference(case):
    ts = np.asarray(case["ts"])
    pr = O.Problem(case["model"], alg=case["alg"], stepper=case["stepper"], t0=case["tspan"][0], t1=case["tspan"][1], dt=case["dt"],
                   abstol=case["abstol"], reltol=case["reltol"], save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0,
                   checkpointing=case["checkpointing"], checkpoints=case.get("checkpoints"), quad_abstol=case["quad_abstol"], quad_reltol=case["quad_reltol"])
    du0, dp, out = pr.adjoint(case["u0"], case["p"])
    if case["stepper"] == "TSIT5":
        _, nsteps = pr.forward(case["u0"], case["p"])
        assert nsteps == case["forward_steps"], f"step sequence differs: {case['targets']}"
    assert rel(out, np.asarray(case["out"])) < 1e-9
    assert rel(du0, case["du0"]) < 1e-6 and rel(dp, case["dp"]) < 1e-6, case["targets"]      # BASELINE.json north_star: rtol 1e-6
