cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v8
timeout 900 python -m pytest tests/test_gpu_etd.py -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/v8/gpu_tests.log 2>&1
tail -25 gpurun_out/v8/gpu_tests.log | cut -c1-300
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v8/smoke.log 2>&1
tail -12 gpurun_out/v8/smoke.log | cut -c1-200
python - <<'PY' > gpurun_out/v8/etd_rows.json 2> gpurun_out/v8/etd_rows.err
import json, sys, numpy as np
sys.path.insert(0, 'tests')
import scimlsensitivity_jl_amd as sa
from test_gpu_parity import bruss_u0
G, dte, Se = 32, 0.0015625, 7360
tsh = 0.5 * np.arange(0, 24); rng = np.random.default_rng(0); rows = []
for alg, N in (("quadrature", 1), ("gauss", 1), ("interpolating", 1), ("interpolating", 64)):
    eng = sa.Engine("bruss", alg, N, 0.0, Se * dte, dte, save_times=tsh, dims=(G, 0, 0, 0), stepper=2)
    u0 = bruss_u0(G, N); p = np.array([3.4, 1.0, 10.0]); d = rng.standard_normal((N, len(tsh), 2 * G * G))
    eng.forward(u0, p, want_out=False); eng.adjoint(d); s0 = eng.stats(); eng.adjoint(d); s1 = eng.stats()
    rows.append(dict(alg=alg, N=N, forward_ms=s1["forward_ms_last"], reverse_ms=s1["adjoint_ms_total"] - s0["adjoint_ms_total"], kernel_ms=s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"], GB=s1["workspace_bytes"] / 1e9))
    eng.close()
print(json.dumps(rows))
PY
cat gpurun_out/v8/etd_rows.json; tail -3 gpurun_out/v8/etd_rows.err
