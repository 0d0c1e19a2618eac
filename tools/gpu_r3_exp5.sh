#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_lanes.py -m gpu -q -x > gpurun_out/r3_lanes_pytest.log 2>&1
echo "lanes rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_lanes_pytest.log | tail -5
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_gpu_lanes.py > gpurun_out/r3_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_pytest_gpu.log | tail -5
for sh in 128,128,28,1,3 64,64,56,1,3; do for v in "X=0" "BTX_NO_TALL=1"; do echo "== $sh $v"; env $v BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_tall.log 2>&1
grep "==\|epilogue\|total\|A->B\|prologue" gpurun_out/r3_ptrace_tall.log
for bs in 64 256 512; do
  BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 300 python tools/kbench.py --throughput-plan --env - BTX_NO_TALL=1 --bs $bs --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_exp5_tall_ab.log 2>&1
cat gpurun_out/r3_exp5_tall_ab.log
