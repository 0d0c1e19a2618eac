R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
export BTX_LIB=build_variants/libbtx_tune.so
timeout 600 python tools/kbench.py --bs 2048 --throughput-plan --shapes 256,1024,14,1,1 128,512,28,1,1 512,2048,7,1,1 512,128,28,1,1 1024,256,14,1,1 256,128,56,1,1 --env - BTX_NO_GEMM8=1 BTX_NO_GEMM8=1,BTX_NO_DMA_PW=1 --rounds 5 --reps 10 2>&1 | grep Flipout > $O/kbench_g8.txt
cat $O/kbench_g8.txt
