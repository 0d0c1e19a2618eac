#!/bin/bash
# A/B: swizzle of the patch on the output-row pixel count (new libbtx.so) vs on the patch pixel index (build_variants/libbtx_tune.so, built before)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_at_size.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
for sh in 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3; do
  echo -n "new "; timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"
  echo -n "old "; BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"
done
B="python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1 2; do
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new', d['value'], d['ms_per_step'])"
BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('old', d['value'], d['ms_per_step'])"
done
