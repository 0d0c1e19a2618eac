#!/bin/bash
# hang watch: the full bench line several times in a row (one run of the round aborted with "GPU Hang" while the
# LDS-counter group barriers were in stem_pool_kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 600 python bench.py > gpurun_out/r2x_bench$i.log 2> gpurun_out/r2x_bench$i.err; echo "full bench $i rc=$? $(grep -i "hang\|exception" gpurun_out/r2x_bench$i.err | head -2 | tr '\n' ' ') $(tail -1 gpurun_out/r2x_bench$i.log | cut -c1-80)"
done
