#!/bin/bash
# A/B: residual requested ahead of epilogue stage 1 (new libbtx.so) vs inside stage 2 (build_variants/libbtx_tune.so, built before)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_at_size.py tests/test_gpu_contract.py -x -q -m gpu 2>&1 | tail -2 | cut -c1-300
B="python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1 2 3; do
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new', d['value'], d['ms_per_step'])"
BTX_LIB=$PWD/build_variants/libbtx_tune.so timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('old', d['value'], d['ms_per_step'])"
done
