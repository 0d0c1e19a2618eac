#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_lanes.py tests/test_gpu_rng_kl.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r3_lanes_pytest.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3_lanes_pytest.log | tail -5
for cfg in "--lanes 4 --lane-mode launch"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --steps 24 --warmup 3 --no-extras --no-cpu-baseline --no-traffic $cfg > gpurun_out/r3_bench_$tag.log 2>&1
  echo "== $cfg rc=$?"
  tail -1 gpurun_out/r3_bench_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('value %.1f ms/step %.4f' % (d['value'], d['ms_per_step']))
print('roofline frac %.4f (contraction only %.4f) sampling_us %.1f lanes/launch %d e2e %.4f' % (r['frac'], r['frac_contraction_only'], r['sampling_us_per_launch'], r['mc_samples_per_launch'], r['frac_e2e']))
"
done 2>&1 | tee gpurun_out/r3_bench_lanes_summary2.txt
