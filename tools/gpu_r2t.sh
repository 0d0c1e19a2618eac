#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "stem or fused_resnet18" 2>&1 | tail -5 | cut -c1-400
python tools/stem_bench.py 64 2>&1 | grep -v amdgpu.ids | tail -6
BTX_LIB=$PWD/build_variants/libbtx_trace.so python tools/stem_trace.py Flipout 2>&1 | grep -v amdgpu.ids | tail -17
