#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2z_pytest.log
export BTX_LIB=$PWD/build_variants/libbtx_tune.so
B="python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
for i in 1 2; do
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('taps2', d['value'], d['ms_per_step'])"
BTX_NO_TAPS2=1 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('dma  ', d['value'], d['ms_per_step'])"
done
