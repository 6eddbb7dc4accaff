cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v12
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "routed" > gpurun_out/v12/gpu_tests.log 2>&1
tail -15 gpurun_out/v12/gpu_tests.log | cut -c1-300
python - <<'PY' 2> gpurun_out/v12/cross.err | tee gpurun_out/v12/cross.json
import json, sys, numpy as np
sys.path.insert(0, 'tests')
import bench, scimlsensitivity_jl_amd as sa
import torch
rows = bench.wide_rows(sa, None) if False else None
PY
python - <<'PY' 2> gpurun_out/v12/cross.err | tee gpurun_out/v12/cross.json
import json, sys, numpy as np
sys.path.insert(0, 'tests')
import scimlsensitivity_jl_amd as sa, scimlsensitivity_jl_amd.interface as _I
from test_gpu_parity import mlp_params
rng = np.random.default_rng(11)
def run(eng, u0, p, delta, reps):
    eng.forward(u0, p, want_out=False); eng.adjoint(delta); s0 = eng.stats()
    for _ in range(reps): eng.adjoint(delta)
    s1 = eng.stats()
    return (s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / reps, s1
Sx, Tx, Nx = 150, 1.5, 4096
tsx = np.linspace(0.0, Tx, 16); cross = []
for Hx in (32, 64, 128):
    funx = sa.WideDeviceFunction.dense_chain(f"bench_chain_{Hx}", (2, Hx, Hx, 2))
    px = mlp_params(2, Hx); u0x = rng.standard_normal((Nx, 2)); dx = rng.standard_normal((Nx, len(tsx), 2))
    row = dict(H=Hx)
    for name, mk, u0e, de in (("workgroup_per_trajectory", lambda: sa.Engine(funx.name, "gauss", Nx, 0.0, Tx, Tx / Sx, save_times=tsx), u0x, dx),
                               ("fp64_mfma", lambda: sa.Engine("mlp", "gauss", 1, 0.0, Tx, Tx / Sx, save_times=tsx, dims=(2, Hx, Nx, 0)), _I._to_columns(u0x), _I._to_columns(dx))):
        eng = mk(); ms, st = run(eng, u0e, px, de, 2)
        row[name] = dict(forward_ms=st["forward_ms_last"], reverse_ms=ms); eng.close()
    row["mfma_speedup_reverse"] = row["workgroup_per_trajectory"]["reverse_ms"] / row["fp64_mfma"]["reverse_ms"]
    cross.append(row)
print(json.dumps(cross))
PY
tail -3 gpurun_out/v12/cross.err
