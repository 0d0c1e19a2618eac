#!/bin/bash
# sustained-load phase timers: does the shader clock differ between the persistent and the plain kernel?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BTX_NO_TALL=1 BTX_LIB=$PWD/build_variants/libbtx_trace.so
for sh in 64,64,56,1,3 128,128,28,1,3; do for v in "BTX_PERSIST=1" "X=0"; do echo "== $sh $v (600 warm launches)"; env $v timeout 300 python tools/gpu_diag.py trace --throughput-plan --bs 256 --warm 600 --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_sustained.log 2>&1
cat gpurun_out/r3_ptrace_sustained.log
