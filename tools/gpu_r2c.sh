#!/bin/bash
# round-2 session C: phase timers of the tap-unrolled kernel, kernel-only durations, split-K slot A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for sh in 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3; do echo "== $sh"; BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v "amdgpu.ids\|wave "; done > gpurun_out/r2c_ptrace.log 2>&1; echo "ptrace rc=$?"
cat gpurun_out/r2c_ptrace.log
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r2c_ks" -o ks -- python "$R/tools/kbench.py" --rounds 2 --env - > "$R/gpurun_out/r2c_ks.log" 2>&1; cd "$R"
find gpurun_out/r2c_ks -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220 | head -12
for v in 256 512; do for sh in 256,256,14,1,3 512,512,7,1,3 128,128,28,1,3; do echo -n "SLOTS4=$v "; BTX_SLOTS4=$v timeout 120 python tools/kbench.py --rounds 2 --env - --shapes $sh 2>&1 | grep -E "TFLOP|rror"; done; done > gpurun_out/r2c_slots.log 2>&1; cat gpurun_out/r2c_slots.log
