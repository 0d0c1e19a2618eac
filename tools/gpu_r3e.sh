#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backward.py -q -m gpu 2>&1 | tail -4 | cut -c1-300
export TMPDIR=/tmp
R="$PWD"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r3e_train" -o train -- python "$R/tools/train_profile.py" > "$R/gpurun_out/r3e_train.log" 2>&1; echo "rc=$?"
grep "ms_per_step" "$R/gpurun_out/r3e_train.log" | cut -c1-60,200-330
