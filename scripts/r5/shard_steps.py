#!/usr/bin/env python
"""Fixed cost vs per-step cost of a small shard's reverse pass: the 1250-trajectory Lorenz InterpolatingAdjoint sweep at 51 time segments for trajectories of 255 ... 4080 steps
(the segments grow with the trajectory, the wavefront count stays 1020), one-launch and three-launch form.  The intercept of ms(steps) is what a shard pays before its first
and after its last step (launch, prologue, first knot latency, composition tree, dp reduction); the slope is the sweep itself.   python scripts/r5/shard_steps.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(S, n=1250, seg=51):
    import numpy as np, torch
    import scimlsensitivity_jl_amd as sa
    dt = 0.01
    rng = np.random.default_rng(20240601)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
    T = S * dt
    eng = sa.Engine("lorenz", "interpolating", n, 0.0, T, dt, save_times=np.linspace(0.0, T, S // 5 + 1)[::2], loss_kind=1, loss_shift=2.0, time_segments=seg)
    dev = torch.device("cuda:0")
    tu0, tp = torch.tensor(u0, device=dev), torch.tensor(p, device=dev)
    du0, dp = torch.empty((n, 3), dtype=torch.float64, device=dev), torch.empty(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(st):
        eng.use_torch_stream()
        eng.forward_dev(tu0, tp, None)
        for _ in range(1500):
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
        eng.set_timing(1)
        for _ in range(3):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize(); eng.synchronize()
        s0 = eng.stats()
        for _ in range(50):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize(); eng.synchronize()
        s1 = eng.stats()
    print(json.dumps(dict(n=n, steps=S, segments=s1["time_segments"], launches=s1["launches_per_pass"], ms_per_pass=best,
                          main_kernel_ms=(s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / 50, fused=os.environ.get("HIPADJ_FUSED", "1"))))
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "_child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
        seg = int(sys.argv[2]) if len(sys.argv) > 2 else 51
        for fused in ("1", "0"):
            for S in ((255, 510, 1020, 2040, 4080) if seg == 51 else (260, 520, 1040, 2080, 4160)):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "_child", str(S), str(n), str(seg)], env=dict(os.environ, HIPADJ_FUSED=fused), capture_output=True, text=True)
                lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
                print(lines[-1] if lines else json.dumps(dict(steps=S, error=r.stderr[-300:])))
