"""Where a host-pointer gradient call (hipadj_forward + hipadj_adjoint with pageable arrays, the Julia binding's route) spends its time on BASELINE configs[1]
with a cotangent block of 24 MB: run under HIPADJ_HOST_TIMING=1 (stderr trace of upload_block), HIPADJ_NO_PINNED=1 (plain pageable copies) and
HIPADJ_HOST_COPY_THREADS=k to compare the staging variants.  One JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scimlsensitivity_jl_amd as sa
import bench
N = 10000
u0, p = bench.inputs(N)
ts = bench.save_times(); M = len(ts)
eng = sa.Engine("lorenz", "interpolating", N, 0.0, bench.T_FINAL, bench.DT, save_times=ts, loss_kind=0, p_shared=True)
delta = np.random.default_rng(3).standard_normal((N, M, 3))
eng.set_timing(0)
# what bench.py's process has that a bare host program has not: torch loaded (PROBE_TORCH=1), device work enqueued through a torch stream the handle adopted
# (PROBE_TSTREAM=1), the cotangents coming out of a torch CPU tensor (PROBE_TDELTA=1)
if os.environ.get("PROBE_TORCH") or os.environ.get("PROBE_TSTREAM") or os.environ.get("PROBE_TDELTA"):
    import torch
    x = torch.zeros(1 << 20, device="cuda"); x += 1; torch.cuda.synchronize()
    if os.environ.get("PROBE_TSTREAM"):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            eng.use_torch_stream()
    if os.environ.get("PROBE_TDELTA"):
        delta = torch.tensor(delta, device="cuda").cpu().numpy()
eng.forward(u0, p, want_out=True); eng.adjoint(delta)
R = 8
t0 = time.perf_counter()
for _ in range(R):
    eng.forward(u0, p, want_out=True)
t1 = time.perf_counter()
for _ in range(R):
    eng.adjoint(delta)
t2 = time.perf_counter()
for _ in range(R):
    eng.forward(u0, p, want_out=False)
t3 = time.perf_counter()
print(json.dumps(dict(variant=os.environ.get("VARIANT", "default"), forward_with_out_ms=(t1 - t0) / R * 1e3, adjoint_ms=(t2 - t1) / R * 1e3, forward_no_out_ms=(t3 - t2) / R * 1e3,
                      bytes_up_adjoint=delta.nbytes, pcie_bound_ms=delta.nbytes / 63e9 * 1e3)))
