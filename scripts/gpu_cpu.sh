#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ.setdefault("OMP_PROC_BIND","spread"); os.environ.setdefault("OMP_PLACES","threads")
import bench, numpy as np, oracle as O
u0,p=bench.inputs(40000); ts=np.arange(0,10.0001,0.1)
pr = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=10.0, dt=0.01, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
for nt in (1,32,64,128,256):
    n=min(40000, 156*nt)
    for rep in range(3):
        t0=time.perf_counter(); _,_,_,tm=pr.adjoint_ensemble(u0[:n],p,nthreads=nt,want_out=False); w=time.perf_counter()-t0
        print(nt, n, rep, 'wall %.3f rev %.3f fwd %.3f  traj/s(rev) %.0f'%(w, tm['reverse_s'], tm['forward_s'], n/tm['reverse_s']), flush=True)
PY
timeout 600 python bench.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_max_rel_du0_vs_oracle_sample']); print(d['cpu_baseline'])"
