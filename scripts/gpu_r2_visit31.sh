#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r2v31; mkdir -p $OUT; cd $REPO
ulimit -c 0
fmt='
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if "n" in d and "error" not in d: print("%s n=%d %-13s depth=%d segs=%-3s fwd %.3f main %.3f ms  err %.1e %.1e build %.1fs" % (d["model"], d["n"], d["alg"], d["depth"], d["time_segments"], d["forward_ms"] or 0, d["main_kernel_ms"], d["err_du0"], d["err_dp"], d["build_s"]))
    elif "compiler" not in d: print(l.strip()[:300])
'
PF_SIZES=lv,rober,2,3,4,5,6,8 PF_DEPTHS=0 PF_ALGS=interpolating,gauss timeout 900 python -X faulthandler scripts/bench_user_pf.py 2> $OUT/err.log | python -c "$fmt" | tee $OUT/user_defaults.log; grep -v amdgpu.ids $OUT/err.log | tail -5
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_full.log
