#!/bin/bash
# round-3 session 3: lanes tests, full gpu suite (full log), phase traces old tiles vs tall strips, A/B kbench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lanes.py -m gpu -q 2>&1 | tail -60 > gpurun_out/r3_lanes_pytest.log
echo "lanes rc=$?"; tail -12 gpurun_out/r3_lanes_pytest.log
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_gpu_lanes.py > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_gpu.log | tail -25
for sh in 128,128,28,1,3 256,256,14,1,3 64,64,56,1,3; do for v in "X=0" "BTX_NO_TALL=1"; do echo "== $sh $v"; env $v BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_tall.log 2>&1
cat gpurun_out/r3_ptrace_tall.log
for bs in 64 256; do
  BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 300 python tools/kbench.py --env - BTX_NO_TALL=1 --bs $bs --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_exp3_tall_ab.log 2>&1
cat gpurun_out/r3_exp3_tall_ab.log
