#!/bin/bash
# pure kernel durations (rocprofv3 kernel trace) of the persistent vs the plain tap-unrolled kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$PWD"; cd /tmp
export BTX_NO_TALL=1 BTX_LIB=$R/build_variants/libbtx_tune.so
for v in 1 0; do
  if [ $v = 1 ]; then export BTX_PERSIST=1; else unset BTX_PERSIST; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_kt_persist$v -o kt -- python $R/tools/kbench.py --throughput-plan --env - --bs 256 --rounds 2 --reps 10 --shapes 64,64,56,1,3 128,128,28,1,3 > $R/gpurun_out/r3_kt_persist$v.log 2>&1
  echo "persist=$v rc=$?"
  f=$(ls $R/gpurun_out/r3_kt_persist$v/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-200
done
