# E15: parity-major pixel order for the data gradient of stride-2 convolutions (only the taps of the tile's parity class are walked)
O=$GRAFT_REPO_ROOT/gpurun_out/r5v; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout -k 5 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/pytest.log
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-420) > $O/train.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py --graph 20 > $O/train_prof.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); timeout -k 5 60 python $GRAFT_REPO_ROOT/tools/trace_report.py $db --sequence 480 > $O/train_trace.txt
rm -rf $O/prof
tail -4 $O/pytest.log; cat $O/train.log; grep replays $O/train_prof.log; head -14 $O/train_trace.txt | cut -c1-150; grep "^TOTAL" $O/train_trace.txt
