#!/bin/bash
# what the K loop of contract_taps_kernel pays for: ablation builds (results are wrong by construction; time only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BTX_NO_TALL=1
for v in tune abl4 abl16 abl2 abl22; do echo "## $v"
BTX_LIB=$PWD/build_variants/libbtx_$v.so timeout 300 python tools/kbench.py --throughput-plan --env - --bs 256 --rounds 2 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3_exp25_ablation.log
