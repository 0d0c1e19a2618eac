# E14 (cont.): nn.MaxPool2d under autograd through btx_maxpool2d_cl_train / _bwd
O=$GRAFT_REPO_ROOT/gpurun_out/r5z; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout -k 5 300 python -m pytest tests/test_gpu_backward.py -x -q -k "maxpool or batchnorm or captured or readme" 2>&1 | tail -6) > $O/pytest.log
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-420) > $O/train.log
tail -3 $O/pytest.log; cat $O/train.log
