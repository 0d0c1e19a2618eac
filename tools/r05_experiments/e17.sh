cd $GRAFT_REPO_ROOT; O=gpurun_out/r5o; mkdir -p $O
export BTX_LIB=build_variants/libbtx_tune.so
timeout 300 python tools/kbench.py --bs 1280 --throughput-plan --shapes 64,64,56,1,3 512,512,7,1,3 --env - BTX_TALL_MIN=1.0 BTX_TALL_MIN=99 --rounds 5 --reps 10 2>&1 | grep Flipout > $O/kbench_tall.txt
cat $O/kbench_tall.txt
