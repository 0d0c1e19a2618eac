#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_backward.py -m gpu -q > gpurun_out/r3_pytest_bwd.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_bwd.log | tail -12
