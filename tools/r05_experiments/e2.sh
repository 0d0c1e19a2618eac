R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
export BTX_LIB=build_variants/libbtx_tune.so
timeout 300 python tools/kbench.py --bs 1280 --throughput-plan --shapes 64,64,56,1,3 512,512,7,1,3 --env - BTX_TAPS_TUNE=256 --rounds 5 --reps 10 2>&1 | grep Flipout > $O/kbench.txt
cat $O/kbench.txt
B="--steps 20 --warmup 20 --no-extras --no-cpu-baseline --no-traffic --no-launch-timing --no-sustain"
for i in 1 2; do
for T in 0 256; do
  BTX_TAPS_TUNE=$T python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune $T', d['value'], d['ms_per_step_runs'])" >> $O/bench_ab.txt
done; done
cat $O/bench_ab.txt
