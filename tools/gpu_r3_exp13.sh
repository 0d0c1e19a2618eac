#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_gpu.log | tail -30
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
