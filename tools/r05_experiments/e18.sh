# E13: weight gradient through chunk slabs + the all-taps kernel: parity tests, per-layer times, the captured training step and its trace
O=$GRAFT_REPO_ROOT/gpurun_out/r5r; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout -k 5 200 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -6) > $O/pytest.log
(timeout -k 5 120 python tools/wgrad_bench.py 2>&1 | tail -16) > $O/wgrad_bench.txt
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-700) > $O/train.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/train_profile.py --graph 20 > $O/train_prof.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); timeout -k 5 60 python $GRAFT_REPO_ROOT/tools/trace_report.py $db --sequence 560 > $O/train_trace.txt
rm -rf $O/prof
tail -3 $O/pytest.log; cat $O/wgrad_bench.txt; cat $O/train.log; grep replays $O/train_prof.log; head -24 $O/train_trace.txt | cut -c1-150; grep "^TOTAL" $O/train_trace.txt
