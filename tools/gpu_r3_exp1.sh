#!/bin/bash
# round-3 experiment 1: grid-size scaling of the existing kernels (batch 64 / 128 / 256 / 512 = stand-in for 1/2/4/8 MC lanes per launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ rocminfo | grep -m2 gfx; nproc; } > gpurun_out/box.txt 2>&1
for bs in 64 128 256 512; do
  echo "== bs $bs"
  timeout 300 python tools/kbench.py --env - --bs $bs --rounds 3 --reps 10 --shapes all 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_exp1_bs_scaling.log 2>&1
echo done
tail -50 gpurun_out/r3_exp1_bs_scaling.log
