# E14 (cont.): the data gradient of strided convolutions takes its operands from btx_dgrad_weights too (no eps tensor, no flips)
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout -k 5 300 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -8) > $O/pytest.log
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-420) > $O/train.log
tail -4 $O/pytest.log; cat $O/train.log
