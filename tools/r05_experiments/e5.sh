R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
export BTX_LIB=build_variants/libbtx_ptr.so
for s in 64,256,56,1,1 256,64,56,1,1; do
 echo "== $s bs 2048 PW"; timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape $s --bs 2048 --warm 20 2>&1 | grep -v "wave  " | head -14
 echo "== $s bs 2048 generic"; BTX_NO_DMA_PW=1 timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape $s --bs 2048 --warm 20 2>&1 | grep -v "wave  " | head -14
done > $O/trace.txt 2>&1
cat $O/trace.txt
