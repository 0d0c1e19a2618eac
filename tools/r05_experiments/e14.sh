cd $GRAFT_REPO_ROOT; O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -6
python tools/train_profile.py --graph 20 2>&1 | grep replays
