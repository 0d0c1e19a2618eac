#!/bin/bash
# round-2 A/B: K-groups vs plain 4-wave blocks when several MC samples are in flight (throughput plan), lanes sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export BTX_LIB=$PWD/build_variants/libbtx_tune.so
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing"
run() { echo "== $*"; env "$@" timeout 200 $B --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
for L in 3 4 6; do
  run X=0
  run BTX_NO_KG=1
done
L=4; run BTX_NO_KG=1 BTX_SLOTS4=512
L=3; run BTX_SLOTS4=512
