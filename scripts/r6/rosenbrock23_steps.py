#!/usr/bin/env python
"""Rosenbrock23 on the two stiff-stepper fixtures (tests/golden/stiff_adjoints.json), CPU oracle: accepted / rejected steps of the forward and the reverse pass and the error of
the gradient against the independent forward sensitivities, for both readings of the coefficient of dT in k3 (ORC_RECALL_ROS_K3_T: d = Shampine-Reichelt, 1 = the alternative).
Writes profiles/r6_rosenbrock23_steps.json.  CPU only."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import json, sys, numpy as np, ctypes as C
sys.path.insert(0, %r)
import oracle as O
g = json.load(open(%r))
case, tol, k3 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
O.lib().orc_test_set_recall(10, C.c_double(k3))
if case == "lv":
    c = g["lv"]
    pr = O.Problem("LV", alg="INTERPOLATING", stepper="ROS23", t0=0, t1=10.0, dt=0.0, abstol=tol, reltol=tol, save_times=c["ts"], loss="LSQ_DATA", loss_scale=2.0)
    du0, dp, _ = pr.adjoint(c["u0"], c["p"], np.array(c["target"]))
else:
    c = g["rober"]
    d = np.zeros((2, 3)); d[:, 2] = 1
    pr = O.Problem("ROBER", alg="INTERPOLATING", stepper="ROS23", t0=0, t1=100.0, dt=0.0, abstol=tol * 1e-2, reltol=tol, save_times=c["ts"], loss="COTANGENT")
    du0, dp, _ = pr.adjoint(c["u0"], c["p"], d)
print(json.dumps(dict(dp_rel=float(np.max(np.abs(dp - c["dp"]) / np.abs(c["dp"]))), du0_rel=float(np.max(np.abs(du0 - c["du0"]) / np.abs(c["du0"]))))))
""" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "stiff_adjoints.json"))


def main():
    rows = []
    for case in ("lv", "rober"):
        for tol in (1e-4, 1e-6, 1e-8):
            for k3, name in ((0.29289321881345247560, "d h dT"), (1.0, "h dT")):
                r = subprocess.run([sys.executable, "-c", CHILD, case, str(tol), str(k3)], capture_output=True, text=True, env=dict(os.environ, ORC_TRACE_STEPS="1"), check=True)
                st = {m[0]: dict(accepted=int(m[1]), rejected=int(m[2]), rhs=int(m[3])) for m in re.findall(r"orc integrate: (\w+) accepted (\d+) rejected (\d+) rhs (\d+)", r.stderr)}
                row = dict(case=case, tol=tol, k3_time_term=name, forward=st["forward"], reverse=st["reverse"], **json.loads(r.stdout.strip().splitlines()[-1]))
                rows.append(row); print(json.dumps(row))
    out = dict(what="Rosenbrock23, CPU oracle (oracle/adjoint_oracle.c), InterpolatingAdjoint; errors against tests/golden/stiff_adjoints.json (scipy forward sensitivities)", rows=rows)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r6_rosenbrock23_steps.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
