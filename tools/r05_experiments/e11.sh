cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
python tools/train_profile.py 2>&1 | tail -1 | grep -o "'ms_per_step': [0-9.]*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py > $O/train.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/trace_report.py $db > $O/train_trace.txt
rm -rf $O/prof
head -24 $O/train_trace.txt | cut -c1-130; tail -1 $O/train_trace.txt
