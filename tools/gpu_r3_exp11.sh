#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
BTX_NO_TALL=1 BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 600 python -X faulthandler -m pytest tests/test_gpu_lanes.py tests/test_gpu_at_size.py -m gpu -q -x > gpurun_out/r3_lanes_pytest.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_lanes_pytest.log | tail -4
for bs in 256; do
  BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 300 python tools/kbench.py --throughput-plan --env BTX_NO_TALL=1 BTX_NO_TALL=1,BTX_TAPS_TUNE=1 BTX_NO_TALL=1,BTX_NO_PERSIST=1 - --bs $bs --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_exp11_persist_ab.log 2>&1
cat gpurun_out/r3_exp11_persist_ab.log
