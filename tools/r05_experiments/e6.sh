R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -k "batchnorm" > $O/pytest_bn.log 2>&1; grep -E "passed|failed|Error" $O/pytest_bn.log | tail -5
python tools/train_profile.py 2>&1 | tail -1 | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $R/tools/train_profile.py > $O/train.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $R/tools/trace_report.py $db > $O/train_trace.txt
rm -rf $O/prof
grep "bn_" $O/train_trace.txt | cut -c1-140
