#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1 2; do
for L in 4 8; do
timeout 300 $B --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fused-pool lanes $L', d['value'], d['ms_per_step'])"
timeout 300 $B --no-stem-pool --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('two-launch lanes $L', d['value'], d['ms_per_step'])"
done
done
