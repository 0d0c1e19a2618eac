#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_contract.py -x -q -m gpu -k "fused_noise or explicit_noise or random_geometries" 2>&1 | tail -6 | cut -c1-400
export BTX_LIB=$PWD/build_variants/libbtx_tune.so
for sh in 64,128,56,2,3 128,256,28,2,3 256,512,14,2,3; do for v in "X=0" "BTX_NO_TAPS2=1"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done; done
