#!/bin/bash
# round-2: fused stem + max-pool kernel: parity tests, then the bench with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "stem or fused_resnet18" 2>&1 | tail -15
B="python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1; do
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fused-pool', d['value'], d['ms_per_step'])"
timeout 300 $B --no-stem-pool 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('two-launch', d['value'], d['ms_per_step'])"
done
