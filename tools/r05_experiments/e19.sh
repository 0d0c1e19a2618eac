# E13 (cont.): two-step run-ahead in the tap-per-workgroup kernel, num_batches_tracked folded into the BN statistics launch
O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout -k 5 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
(timeout -k 5 120 python tools/wgrad_bench.py 2>&1 | tail -14) > $O/wgrad_bench.txt
(timeout -k 5 200 python bench.py --train-step-only 2>&1 | tail -1 | cut -c1-420) > $O/train.log
tail -3 $O/pytest.log; cat $O/wgrad_bench.txt; cat $O/train.log
