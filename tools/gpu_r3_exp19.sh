#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r3_exp15.sh
bash tools/gpu_r3_exp18.sh 2>&1 | grep "==\|prologue\|K loops\|store sides\|total\|ss:\|tiles"
