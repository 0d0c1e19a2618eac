#!/bin/bash
# round-2: kernel-trace stats of the bench command under the throughput plan without K-groups, 4 samples in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r2r_prof" -o bench -- python "$R/bench.py" --steps 12 --warmup 4 --no-cpu-baseline --no-extras --no-traffic --no-launch-timing > "$R/gpurun_out/r2r_prof.log" 2>&1; echo "prof rc=$?"
cd "$R"
for L in 4 5 8; do timeout 300 python bench.py --steps 24 --warmup 4 --lanes $L --no-cpu-baseline --no-extras --no-traffic --no-launch-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lanes', $L, d['value'], d['ms_per_step'])"; done
