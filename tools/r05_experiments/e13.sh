O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py --graph 20 > $O/train.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/trace_report.py $db > $O/train_trace.txt
rm -rf $O/prof
grep replays $O/train.log; head -45 $O/train_trace.txt | cut -c1-150; tail -1 $O/train_trace.txt
