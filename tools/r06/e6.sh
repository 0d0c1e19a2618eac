# E6 (round 6): ablations of stem_pool_kernel (trace builds, wrong results): where a K pass spends its time.  Measured on the round-5
# step layout: F="-DBTX_TUNING -DBTX_PT_TRACE -DBTX_STEM_PASS3=0 -DBTX_STEM_NB=2"; bash tools/build_variants.sh trace "$F" spa1 "$F -DBTX_SP_ABL=1" spa2 "$F -DBTX_SP_ABL=2" spa4 "$F -DBTX_SP_ABL=4"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e6; mkdir -p $O
cd $R
for V in trace spa1 spa2 spa4; do
  for T in Flipout Reparameterization; do
    echo "== $V $T" >> $O/abl.txt
    BTX_LIB=build_variants/libbtx_$V.so timeout 120 python tools/stem_trace.py $T 2>&1 | grep -v "wave \|amdgpu" >> $O/abl.txt
  done
done
cat $O/abl.txt
