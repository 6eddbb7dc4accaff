#!/usr/bin/env python
"""The create-time race of DESIGN 7.2 with the load generated INSIDE the process: three host threads keep the device busy (Lorenz 10^4 reverse passes on their own handles) while the
main thread creates small LSQ_DATA handles and compares every gradient with the first one.  HIPADJ_CREATE_NO_DRAIN=1 = the library as it was.  One JSON line."""
import json, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
sa.build_extension()

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    stop = threading.Event()
    def load(seed):
        rng = np.random.default_rng(seed)
        N = 10000
        u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
        eng = sa.Engine("lorenz", "interpolating", N, 0.0, 10.0, 0.01, save_times=np.arange(0, 10.0 + 1e-9, 0.1), loss_kind=1, loss_shift=2.0)
        eng.forward(u0, p, want_out=False)
        while not stop.is_set():
            eng.adjoint(None)
        eng.close()
    th = [threading.Thread(target=load, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    rng = np.random.default_rng(9)
    N, T, dt = 66, 2.0, 0.01
    u0 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p = np.array([1.5, 1.0, 3.0, 1.0]); ts = np.array([0.503, 1.0, 1.777, 2.0])
    data = rng.standard_normal((N, len(ts), 2))
    ref, bad = None, 0
    try:
        for it in range(iters):
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem("lv", u0[0], (0, T), p), u0), sa.RK4(), dt=dt, saveat=ts, sensealg=sa.GaussAdjoint(), dgdu_discrete=sa.LsqData(data, 2.0))
            du0, dp = sa.adjoint_sensitivities(sol, sa.RK4(), t=ts, dgdu_discrete=sa.LsqData(data, 2.0))
            sol.engine.close()
            if ref is None:
                ref = (du0.copy(), dp.copy())
            elif not (np.array_equal(du0, ref[0]) and np.array_equal(dp, ref[1])):
                bad += 1
    finally:
        stop.set(); [t.join() for t in th]
    print(json.dumps(dict(no_drain=os.environ.get("HIPADJ_CREATE_NO_DRAIN") == "1", handles=iters, wrong=bad)))

if __name__ == "__main__":
    main()
