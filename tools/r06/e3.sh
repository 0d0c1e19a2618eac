# E3 (round 6): where the time of a Reparameterization 56x56 tile and of the stem + pool phases goes (trace build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e3; mkdir -p $O
cd $R
export BTX_LIB=build_variants/libbtx_trace.so
for T in Reparameterization Flipout; do
  echo "== 64,64,56,1,3 $T bs 256" >> $O/taps_trace.txt
  timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape 64,64,56,1,3 --typ $T --bs 256 2>&1 | grep -v "^   wave" >> $O/taps_trace.txt
  echo "== 64,64,56,1,3 $T bs 256 BTX_TAPS_TUNE=128" >> $O/taps_trace.txt
  BTX_TAPS_TUNE=128 timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape 64,64,56,1,3 --typ $T --bs 256 2>&1 | grep -v "^   wave" >> $O/taps_trace.txt
  echo "== 128,128,28,1,3 $T bs 1024" >> $O/taps_trace.txt
  timeout 120 python tools/gpu_diag.py trace --prec bf16 --shape 128,128,28,1,3 --typ $T --bs 1024 2>&1 | grep -v "^   wave" >> $O/taps_trace.txt
  timeout 120 python tools/stem_trace.py $T > $O/stem_trace_$T.txt 2>&1
done
cat $O/*.txt
