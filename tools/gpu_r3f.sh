#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3f_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3f_pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r3f_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > gpurun_out/r3f_bench.log 2> gpurun_out/r3f_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r3f_bench.log | cut -c1-120
