# E15 (cont.): parity-major classes padded to whole tiles, class-minor tile order, multiply-shift stride division
O=$GRAFT_REPO_ROOT/gpurun_out/r5y; mkdir -p $O
cd $GRAFT_REPO_ROOT
(BTX_LIB=build_variants/libbtx_tune.so timeout -k 5 150 python tools/dgrad_bench.py 2>&1 | tail -8) > $O/dgrad_bench.txt
(timeout -k 5 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_contract.py -x -q 2>&1 | tail -5) > $O/pytest.log
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-420) > $O/train.log
cat $O/dgrad_bench.txt; tail -3 $O/pytest.log; cat $O/train.log
