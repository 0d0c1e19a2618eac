# E1 (round 6): what the LDS bank conflicts of the activation-fragment reads cost (upper bound: -DBTX_PT_NOJUMP, wrong results),
# the wide Reparameterization tile A/B, and the parked persistent kernel on Reparameterization (not power-limited)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e1; mkdir -p $O
cd $R
S="64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3"
for i in 1 2; do
for V in tune nojump; do
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --bs 1280 --throughput-plan --shapes $S --env - --rounds 5 --reps 10 2>&1 | grep Flipout | sed "s/^/$V /" >> $O/conflicts_flipout.txt
  BTX_LIB=build_variants/libbtx_$V.so timeout 300 python tools/kbench.py --typ Reparameterization --bs 1024 --throughput-plan --shapes $S --env - BTX_NO_WIDE=1 --rounds 5 --reps 10 2>&1 | grep Reparam | sed "s/^/$V /" >> $O/conflicts_reparam.txt
done; done
BTX_LIB=build_variants/libbtx_tune.so timeout 300 python tools/kbench.py --typ Reparameterization --bs 1024 --throughput-plan --shapes $S --env BTX_NO_WIDE=1 BTX_NO_WIDE=1,BTX_PERSIST=1,BTX_NO_TALL=1 BTX_NO_WIDE=1,BTX_NO_TALL=1 --rounds 5 --reps 10 2>&1 | grep Reparam > $O/persist_reparam.txt
cat $O/*.txt
