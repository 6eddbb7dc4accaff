"""tests/golden/offgrid_ragged.json — an independent gradient for the fixed-step configurations of round 5: loss times off the step grid on a span that is not a multiple
of dt (time-dependent Lotka-Volterra, test/Core3/adjoint.jl:8-51's model), so that the oracle's reverse-step-list paths — with checkpointing, checkpoint lists, GaussKronrod,
Quadrature, Backsolve — are pinned to numerics that share nothing with them: the forward-sensitivity system integrated by scipy DOP853 at rtol = atol = 1e-13 (make_golden.py's
`gradient`) contracted with dl/du = u - 2 at the loss times.  The exact gradient does not depend on where checkpoints lie; the oracle's runs must converge to it as dt shrinks
(tests/test_oracle.py::test_offgrid_ragged_configurations_converge_to_the_independent_gradient).

    python tests/golden/make_offgrid_ragged.py        (needs scipy; a few seconds)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import gradient, lvt      # noqa: E402

if __name__ == "__main__":
    u0, p = [1.0, 1.0], [1.5, 1.0, 3.0, 1.0]
    tspan = (0.0, 1.505)
    ts = [0.137, 0.4, 0.40499, 1.2345, 1.502]
    du0, dp, us = gradient(lvt, u0, np.array(p), tspan, ts, lambda u, i: u - 2.0)
    out = dict(model="LVT", u0=u0, p=p, tspan=list(tspan), ts=ts, loss="sum_i |u(t_i) - 2|^2 / 2", du0=du0.tolist(), dp=dp.tolist(), out=us,
               method="scipy DOP853 forward sensitivities, rtol = atol = 1e-13 (tests/golden/make_golden.py gradient)")
    with open(os.path.join(HERE, "offgrid_ragged.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote offgrid_ragged.json", du0, dp)
