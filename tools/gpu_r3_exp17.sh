#!/bin/bash
# sectioned kernarg reads in contract_taps_kernel: prologue stamps + kbench against the committed library (libbtx.so in-tree = old)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for sh in 64,64,56,1,3; do for v in "BTX_TAPS_TUNE=128" "BTX_TAPS_TUNE=0"; do echo "== $sh $v"; env $v BTX_NO_TALL=1 BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave \|column 7"; done; done > gpurun_out/r3_ptrace_prologue2.log 2>&1
cat gpurun_out/r3_ptrace_prologue2.log
for lib in build_variants/libbtx_tune.so bayesian_torch_amd/libbtx.so; do echo "## $lib"
BTX_LIB=$PWD/$lib timeout 400 python tools/kbench.py --throughput-plan --env - --bs 256 --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3_exp17_kbench.log
