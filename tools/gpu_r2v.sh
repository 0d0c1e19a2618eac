#!/bin/bash
# round-2 closing session: all GPU tests, the bench line, rocprofv3 kernel-trace stats of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2v_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2v_pytest.log
timeout 900 python bench.py > gpurun_out/r2v_bench.log 2> gpurun_out/r2v_bench.err; echo "bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r2v_prof" -o bench -- python "$R/bench.py" --steps 12 --warmup 4 --no-cpu-baseline --no-extras --no-traffic > "$R/gpurun_out/r2v_prof.log" 2>&1; echo "prof rc=$?"
cd "$R"
