# E5 (round 6): stem + pool kernel after (F1) staging constants read up front, (F2) two-phase sign copy, (F3) explicit end-of-stage
# LDS wait in the K passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e5; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu -k "stem or Stem or pool" 2>&1 | tail -4 > $O/tests.txt
python tools/stem_bench.py 64 > $O/stem_bench.txt 2>&1
python tools/stem_bench.py 1280 >> $O/stem_bench.txt 2>&1
for T in Flipout Reparameterization; do
  BTX_LIB=build_variants/libbtx_trace.so timeout 120 python tools/stem_trace.py $T > $O/stem_trace_$T.txt 2>&1
done
cat $O/*.txt
