#!/bin/bash
# round-2 closing session: the bench line (twice), rocprofv3 kernel-trace stats of the bench command, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 300 python __graft_entry__.py smoke > gpurun_out/r2v_smoke.log 2>&1; echo "smoke rc=$?"
for i in 1 2; do timeout 900 python bench.py > gpurun_out/r2v_bench$i.log 2> gpurun_out/r2v_bench$i.err; echo "bench$i rc=$?"; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r2v_prof" -o bench -- python "$R/bench.py" --steps 12 --warmup 4 --no-cpu-baseline --no-extras --no-traffic > "$R/gpurun_out/r2v_prof.log" 2>&1; echo "prof rc=$?"
cd "$R"
