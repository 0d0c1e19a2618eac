#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r3d_train" -o train -- python "$R/tools/train_profile.py" > "$R/gpurun_out/r3d_train.log" 2>&1; echo "rc=$?"
grep "ms_per_step" "$R/gpurun_out/r3d_train.log" | cut -c1-300
