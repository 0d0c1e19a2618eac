#!/bin/bash
# round-2 session A: parity of the tap-unrolled kernel + A/B timing + bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2a_pytest.log
timeout 600 python tools/kbench.py --rounds 3 > gpurun_out/r2a_kbench.log 2>&1; echo "kbench rc=$?"; cat gpurun_out/r2a_kbench.log | grep -v amdgpu.ids
timeout 300 python tools/kbench.py --rounds 3 --typ Reparameterization > gpurun_out/r2a_kbench_rep.log 2>&1; cat gpurun_out/r2a_kbench_rep.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r2a_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2a_bench.log | cut -c1-600
BTX_NO_TAPS=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-launch-timing > gpurun_out/r2a_bench_notaps.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2a_bench_notaps.log | cut -c1-300
