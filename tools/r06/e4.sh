# E4 (round 6): prologue (one batch of kernarg loads, one LDS round trip for the patch table), one-barrier store side of the wide tile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e4; mkdir -p $O
cd $R
S="64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3"
for i in 1 2; do
for V in tune flat; do
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --typ Reparameterization --bs 1024 --throughput-plan --shapes $S --env - BTX_NO_TALL=1 --rounds 5 --reps 10 2>&1 | grep Reparam | sed "s/^/$V /" >> $O/reparam.txt
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --bs 1280 --throughput-plan --shapes $S --env - --rounds 5 --reps 10 2>&1 | grep Flipout | sed "s/^/$V /" >> $O/flipout.txt
done; done
cat $O/*.txt
