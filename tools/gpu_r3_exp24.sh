#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_gpu.log | tail -6
bash tools/gpu_r3_exp12.sh
