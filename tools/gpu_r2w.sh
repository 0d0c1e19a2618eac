#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r2w_$name.log 2> gpurun_out/r2w_$name.err; echo "$name rc=$? $(tail -c 300 gpurun_out/r2w_$name.err | tr '\n' ' ' | cut -c1-200)"; }
run head --no-extras --no-traffic --no-cpu-baseline
run traffic --no-extras --no-cpu-baseline
run extras --no-traffic --no-cpu-baseline
