#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_lanes.py -m gpu -q -x > gpurun_out/r3_lanes_pytest.log 2>&1
echo "lanes rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_lanes_pytest.log | tail -40
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --deselect tests/test_gpu_lanes.py > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_gpu.log | tail -40
for sh in 128,128,28,1,3 64,64,56,1,3; do for v in "X=0" "BTX_NO_TALL=1"; do echo "== $sh $v"; env $v BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_tall.log 2>&1
grep "==\|epilogue\|total\|A->B" gpurun_out/r3_ptrace_tall.log
