#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python bench.py --steps 24 --warmup 3 --no-traffic > gpurun_out/r3_bench_full.log 2>&1
echo "bench rc=$?"
tail -1 gpurun_out/r3_bench_full.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('value %.1f ms/step %.4f runs %s' % (d['value'], d['ms_per_step'], ['%.3f' % v for v in d['ms_per_step_runs']]))
print('roofline frac %.4f (contraction only %.4f) sampling_us %.1f lanes/launch %d e2e %.4f' % (r['frac'], r['frac_contraction_only'], r['sampling_us_per_launch'], r['mc_samples_per_launch'], r['frac_e2e']))
print('parity', d.get('logits_rel_l2_vs_unfused_f32'), 'kl_rel', d.get('kl_rel_err'))
for k, v in d.get('extra', {}).items(): print(k, json.dumps(v)[:400])
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
" 2>&1 | tee gpurun_out/r3_bench_full_summary.txt
tail -5 gpurun_out/r3_bench_full.log | cut -c1-300
