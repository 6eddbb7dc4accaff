#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/host
timeout 300 python scripts/bench_host_api.py 2>/dev/null | tee gpurun_out/host/host_api.json
