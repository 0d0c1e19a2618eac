R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --no-traffic --no-launch-timing --no-sustain"
for L in 20 10; do
  rocprofv3 --kernel-trace -d $O/prof$L -o t -- python $R/bench.py --steps 40 --warmup 20 --lanes $L $B > /dev/null 2>&1
  db=$(find $O/prof$L -name "*.db" | head -1)
  python $R/tools/trace_report.py $db --sequence 140 > $O/seq$L.txt
  rm -rf $O/prof$L
done
