#!/bin/bash
# where the prologue of contract_taps_kernel goes (sub-stamps, trace build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for sh in 64,64,56,1,3 128,128,28,1,3; do for v in "BTX_TAPS_TUNE=128" "BTX_TAPS_TUNE=0"; do echo "== $sh $v"; env $v BTX_NO_TALL=1 BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_prologue.log 2>&1
cat gpurun_out/r3_ptrace_prologue.log
