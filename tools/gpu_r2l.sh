#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1 2; do for L in 8 12 16 24; do
timeout 300 $B --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lanes $L', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
