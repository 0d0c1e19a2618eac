#!/bin/bash
# persistent kernel (taps3) after pinning tile scalars to SGPRs: A/B against contract_taps_kernel + parity
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BTX_NO_TALL=1
BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 400 python tools/kbench.py --throughput-plan --env BTX_PERSIST=1 - --bs 256 --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_exp15_persist.log
BTX_PERSIST=1 BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 600 python -m pytest tests/test_gpu_lanes.py -m gpu -q -x 2>&1 | tail -5 | tee -a gpurun_out/r3_exp15_persist.log
