#!/bin/bash
# round-3 session 2: parity of the lanes + tall-strip changes, then kbench at bs 64/256
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lanes.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r3_lanes_pytest.log
echo "lanes rc=$?"; tail -30 gpurun_out/r3_lanes_pytest.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_lanes.py 2>&1 | tail -30 > gpurun_out/r3_pytest_gpu.log
tail -15 gpurun_out/r3_pytest_gpu.log
for bs in 64 256; do
  echo "== bs $bs"
  timeout 300 python tools/kbench.py --env - --bs $bs --rounds 3 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3_exp2_tall.log 2>&1
cat gpurun_out/r3_exp2_tall.log
