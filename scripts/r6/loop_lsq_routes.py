import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import scimlsensitivity_jl_amd as sa
sa.build_extension()
import test_gpu_device_loss as T
bad = 0
for it in range(int(sys.argv[1])):
    try:
        T.test_model_body_for_lsq_equals_the_builtin_kind(sa)
    except AssertionError as e:
        bad += 1; print("iteration", it, "FAILED", str(e)[:200], flush=True)
    if it % 10 == 5 and len(sys.argv) > 2:
        T.test_pipelined_host_transfers_from_concurrent_host_threads(sa)
print("iterations", sys.argv[1], "failures", bad)
