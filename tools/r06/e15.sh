# E15 (round 6): stage 2 of the staged store side with branch-free bf16 stores (buffer stores, out-of-range offset for pixels that do not exist)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e15; mkdir -p $O
cd $R
for i in 1 2; do
for V in pred tune; do
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --typ Reparameterization --bs 1024 --throughput-plan --shapes 64,64,56,1,3 128,128,28,1,3 64,128,56,2,3 --env - --rounds 5 --reps 10 2>&1 | grep Reparam | sed "s/^/$V /" >> $O/ab.txt
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --bs 1280 --throughput-plan --shapes 64,64,56,1,3 128,128,28,1,3 --env - --rounds 5 --reps 10 2>&1 | grep Flipout | sed "s/^/$V /" >> $O/ab.txt
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --bs 2048 --throughput-plan --shapes 64,256,56,1,1 256,64,56,1,1 64,64,56,1,1 64,128,56,2,1 --env - --rounds 5 --reps 10 2>&1 | grep Flipout | sed "s/^/$V /" >> $O/ab.txt
done; done
cat $O/ab.txt
