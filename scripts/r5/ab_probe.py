#!/usr/bin/env python
"""GPU probes behind two items of VERDICT r4 (run on the GPU box after scripts/r5/ab_variants.sh built the variant libraries on the CPU container):

  python scripts/r5/ab_probe.py ticket     the composition tree's hand-off ticket: relaxed RMW + compiler barrier (shipped) vs an agent-scope acquire-release RMW
                                           (scripts/libhipadj_Tacqrel.so) — sustained ms per reverse pass at 10^4, 2500 and 1250 trajectories, alternating A/B/A/B
  python scripts/r5/ab_probe.py nz1        the one-component Gauss instantiation of the quad Tsit5 sweep (scripts/libhipadj_Tnz1.so, -DHIPADJ_QUAD_GAUSS_NZ=1), which came back
                                           wrong on the device in round 4 while the host build of the same source is exact: the canary — is it still wrong with this toolchain?
Each measurement runs in a child process with HIPADJ_LIBRARY pointing at the build under test; one JSON line per result."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child_ticket(n):
    import numpy as np
    import torch
    import scimlsensitivity_jl_amd as sa
    rng = np.random.default_rng(20240601)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.linspace(0.0, 10.0, 101)
    eng = sa.Engine("lorenz", "interpolating", n, 0.0, 10.0, 0.01, save_times=ts, loss_kind=1, loss_shift=2.0)
    dev = torch.device("cuda:0")
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((n, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, None)
        for _ in range(min(2000, int(4e6 / n))):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(200):
                eng.adjoint_dev(None, du0, dp)
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200)
    eng.synchronize()
    print(json.dumps(dict(n=n, ms_per_pass=best, dp=[float(x) for x in dp.cpu()], lib=os.environ.get("HIPADJ_LIBRARY", "shipped"))))
    eng.close()


def child_nz1():
    import numpy as np
    import oracle as O
    import scimlsensitivity_jl_amd as sa
    os.environ["HIPADJ_QUAD"] = "1"
    rng = np.random.default_rng(20240601)
    N, T = 37, 1.0
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.linspace(0.0, T, 11)
    out = []
    for tol in (1e-8, 1e-11):
        eng = sa.Engine("lorenz", "gauss", N, 0.0, T, 0.0, save_times=ts, stepper=1, abstol=tol, reltol=tol, p_shared=False, loss_kind=1, loss_shift=2.0)
        eng.forward(u0, np.tile(p, (N, 1)), want_out=False)
        du0, dp = eng.adjoint(None)
        eng.close()
        ref = O.Problem("LORENZ", alg="GAUSS", stepper="TSIT5", t0=0.0, t1=T, dt=0.0, abstol=tol, reltol=tol, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
        rdu0, rdp, _, _ = ref.adjoint_ensemble(u0, np.tile(p, (N, 1)))
        out.append(dict(tol=tol, rel_err_du0=float(np.max(np.abs(du0 - rdu0)) / np.max(np.abs(rdu0))), rel_err_dp=float(np.max(np.abs(dp - rdp)) / np.max(np.abs(rdp)))))
    print(json.dumps(dict(lib=os.environ.get("HIPADJ_LIBRARY", "shipped"), gauss_quad_vs_oracle=out)))


def run(args, lib):
    env = dict(os.environ)
    if lib:
        env["HIPADJ_LIBRARY"] = lib
    else:
        env.pop("HIPADJ_LIBRARY", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    if r.returncode != 0 or not lines:
        print(json.dumps(dict(error=r.stderr[-800:], args=args, lib=lib or "shipped")))
        return None
    print(lines[-1])
    return json.loads(lines[-1])


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "_ticket":
        child_ticket(int(sys.argv[2]))
    elif what == "_nz1":
        child_nz1()
    elif what == "ticket":
        var = os.path.join(ROOT, "scripts", "libhipadj_Tacqrel.so")
        for n in (10000, 2500, 1250):
            for rep in range(2):
                for lib in (None, var):
                    run(["_ticket", str(n)], lib)
    elif what == "nz1":
        run(["_nz1"], None)
        run(["_nz1"], os.path.join(ROOT, "scripts", "libhipadj_Tnz1.so"))
