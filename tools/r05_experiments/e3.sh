R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "lanes or mc or accumulate or contract or golden or fused_noise or nccl" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5
export BTX_LIB=build_variants/libbtx_tune.so
timeout 300 python tools/kbench.py --bs 2048 --throughput-plan --shapes 64,256,56,1,1 64,64,56,1,1 256,64,56,1,1 --env - BTX_NO_DMA_PW=1 --rounds 5 --reps 10 2>&1 | grep Flipout > $O/kbench.txt
cat $O/kbench.txt
B="--arch resnet50 --moped --batch 128 --steps 16 --warmup 16 --lanes 16 --repeats 3 --no-extras --no-traffic --no-cpu-baseline --no-launch-timing --no-sustain"
for i in 1 2; do
for T in 0 1; do
  if [ $T = 1 ]; then export BTX_NO_DMA_PW=1; else unset BTX_NO_DMA_PW; fi
  python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 nopw=$T', d['value'], d['ms_per_step_runs'])" >> $O/bench_ab.txt
done; done
unset BTX_NO_DMA_PW; unset BTX_LIB
cat $O/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $R/bench.py --steps 20 --warmup 20 --no-extras --no-cpu-baseline --no-traffic --no-launch-timing --no-sustain > /dev/null 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $R/tools/trace_report.py $db --sequence 60 | grep -i "accumulate\|dma_kernel\|splitk" | tail -6
rm -rf $O/prof
