#!/usr/bin/env python
"""Are freshly created handles deterministic on a LOADED device?  (The create-time race of DESIGN 7.2 was silent on a quiet one.)  A dozen small configurations across the kernel
families and loss routes — each: create a handle, forward, reverse, close — are repeated `rounds` times while three host threads keep the device busy; every result is compared
bit for bit with the configuration's first one.  Prints one JSON line per configuration that ever differed and a summary line."""
import json, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scimlsensitivity_jl_amd as sa
sa.build_extension()
import user_models as UM

def configs():
    rng = np.random.default_rng(5)
    out = []
    N = 70
    u2 = np.array([1.0, 1.0]) + 0.05 * rng.standard_normal((N, 2)); p4 = np.array([1.5, 1.0, 3.0, 1.0])
    u3 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p3 = np.array([10.0, 28.0, 8 / 3])
    ts = np.linspace(0, 1.0, 6); tso = np.array([0.13, 0.5, 0.77, 1.0])
    d2 = rng.standard_normal((N, len(ts), 2)); d3 = rng.standard_normal((N, len(ts), 3)); d2o = rng.standard_normal((N, len(tso), 2))
    def lane(model, u0, p, alg, stepper, ts_, loss, delta=None, **kw):
        def run():
            sol = sa.solve(sa.EnsembleProblem(sa.ODEProblem(model, u0[0], (0, 1.0), p), u0), stepper, saveat=ts_, sensealg=alg, dgdu_discrete=loss, **kw)
            g = sa.adjoint_sensitivities(sol, stepper, t=ts_, dgdu_discrete=(loss if loss is not None else delta))
            o = None if sol.u is None else np.array(sol.u)
            sol.engine.close()
            return (o,) + tuple(g)
        return run
    out.append(("lv rk4 interpolating lsq_data", lane("lv", u2, p4, sa.InterpolatingAdjoint(), sa.RK4(), ts, sa.LsqData(d2, 2.0), dt=0.01)))
    out.append(("lv rk4 gauss offgrid lsq_data", lane("lv", u2, p4, sa.GaussAdjoint(), sa.RK4(), tso, sa.LsqData(d2o, 2.0), dt=0.01)))
    out.append(("lv rk4 backsolve cotangent", lane("lv", u2, p4, sa.BacksolveAdjoint(), sa.RK4(), ts, None, d2, dt=0.01)))
    out.append(("lorenz rk4 interpolating lsq_shift (one launch)", lane("lorenz", u3, p3, sa.InterpolatingAdjoint(), sa.RK4(), ts, sa.LsqShift(2.0), dt=0.01)))
    out.append(("lorenz rk4 quadrature cotangent", lane("lorenz", u3, p3, sa.QuadratureAdjoint(), sa.RK4(), ts, None, d3, dt=0.01)))
    out.append(("lorenz tsit5 interpolating lsq_data", lane("lorenz", u3, p3, sa.InterpolatingAdjoint(), sa.Tsit5(), ts, sa.LsqData(d3, 2.0), abstol=1e-8, reltol=1e-8)))
    out.append(("lorenz tsit5 gauss ckpt cotangent", lane("lorenz", u3, p3, sa.GaussAdjoint(checkpointing=True), sa.Tsit5(), ts, None, d3, abstol=1e-8, reltol=1e-8)))
    out.append(("lv rosenbrock23 interpolating lsq_data", lane("lv", u2, p4, sa.InterpolatingAdjoint(), sa.Rosenbrock23(), ts, sa.LsqData(d2, 2.0), abstol=1e-8, reltol=1e-8)))
    out.append(("lv rosenbrock23 quadrature cotangent", lane("lv", u2, p4, sa.QuadratureAdjoint(), sa.Rosenbrock23(), ts, None, d2, abstol=1e-8, reltol=1e-8)))
    m = UM.ring(4); fr = sa.DeviceFunction("ring4_det_load", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    ur = rng.uniform(0.3, 1.0, (N, 4)); pr = rng.uniform(0.4, 1.2, 5); dr = rng.standard_normal((N, len(ts), 4))
    out.append(("runtime ring4 rk4 interpolating lsq_data", lane(fr, ur, pr, sa.InterpolatingAdjoint(), sa.RK4(), ts, sa.LsqData(dr, 2.0), dt=0.01)))
    out.append(("runtime ring4 tsit5 backsolve cotangent", lane(fr, ur, pr, sa.BacksolveAdjoint(), sa.Tsit5(), ts, None, dr, abstol=1e-8, reltol=1e-8)))
    fw = sa.WideDeviceFunction.dense_chain("chain_det_load", (2, 32, 32, 2))
    pw = 0.3 * rng.standard_normal(fw.np); uw = rng.standard_normal((64, 2)); dw = rng.standard_normal((64, len(ts), 2))
    out.append(("dense chain 2-32-32-2 routed to the MFMA family, lsq_data", lane(fw, uw, pw, sa.GaussAdjoint(), sa.RK4(), ts, sa.LsqData(dw, 2.0), dt=0.01)))
    out.append(("dense chain 2-32-32-2 on the workgroup family, cotangent", lane(fw, uw[:24], pw, sa.InterpolatingAdjoint(), sa.RK4(), ts, None, dw[:24], dt=0.01, mfma=False)))
    return out

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    stop = threading.Event()
    def load(seed):
        rng = np.random.default_rng(seed)
        u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((10000, 3)); p = np.array([10.0, 28.0, 8 / 3])
        eng = sa.Engine("lorenz", "interpolating", 10000, 0.0, 10.0, 0.01, save_times=np.arange(0, 10.0 + 1e-9, 0.1), loss_kind=1, loss_shift=2.0)
        eng.forward(u0, p, want_out=False)
        while not stop.is_set():
            eng.adjoint(None)
        eng.close()
    cf = configs()
    ref = {name: run() for name, run in cf}            # quiet device
    th = [threading.Thread(target=load, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    bad = {name: 0 for name, _ in cf}
    try:
        for _ in range(rounds):
            for name, run in cf:
                r = run()
                same = all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(r, ref[name]))
                bad[name] += not same
    finally:
        stop.set(); [t.join() for t in th]
    for name, b in bad.items():
        if b:
            print(json.dumps(dict(configuration=name, differing_runs=b, of=rounds)))
    print(json.dumps(dict(no_drain=os.environ.get("HIPADJ_CREATE_NO_DRAIN") == "1", configurations=len(cf), rounds=rounds, configurations_that_differed=sum(1 for b in bad.values() if b))))

if __name__ == "__main__":
    main()
