cd $GRAFT_REPO_ROOT
export BTX_LIB=build_variants/libbtx_tune.so
python tools/r05_experiments/e8.py 2>&1 | tail -1
BTX_NO_DMA=1 NO_PRESAMPLE=1 python tools/r05_experiments/e8.py 2>&1 | tail -1
BTX_NO_DMA=1 python tools/r05_experiments/e8.py 2>&1 | tail -1
python tools/r05_experiments/e8.py 2>&1 | tail -1
