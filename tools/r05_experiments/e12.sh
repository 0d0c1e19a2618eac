cd $GRAFT_REPO_ROOT; O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -s -k "captured or batchnorm or dgrad" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|captured vs|raise|rror:" $O/pytest.log | tail -8
python tools/train_profile.py 2>&1 | tail -1 | grep -o "'ms_per_step': [0-9.]*\|'ms_per_step_eager': [0-9.]*\|'hipgraph': [^,]*"
