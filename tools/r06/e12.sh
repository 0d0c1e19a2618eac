# E12 (round 6): Reparameterization prologue without the patch-pixel table (no LDS round trip, no workgroup barrier in front of the patch DMAs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e12; mkdir -p $O
cd $R
S="64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3"
for i in 1 2; do
for V in nodm tune; do
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --typ Reparameterization --bs 1024 --throughput-plan --shapes $S --env - --rounds 5 --reps 10 2>&1 | grep Reparam | sed "s/^/$V /" >> $O/reparam.txt
done; done
BTX_LIB=build_variants/libbtx_trace.so BTX_TAPS_TUNE=128 timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape 64,64,56,1,3 --typ Reparameterization --bs 256 2>&1 | grep -v "^   wave\|amdgpu" >> $O/trace.txt
BTX_LIB=build_variants/libbtx_trace.so timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape 64,64,56,1,3 --typ Reparameterization --bs 256 2>&1 | grep -v "^   wave\|amdgpu" >> $O/trace.txt
cat $O/*.txt
