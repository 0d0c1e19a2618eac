# E16 (round 6): stem + pool kernel, store role staging of both channel halves in one call (BTX_STEM_PAIR): constants read once
# bash tools/build_variants.sh pair "-DBTX_TUNING" nopair "-DBTX_TUNING -DBTX_STEM_PAIR=0"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e16; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu -k "stem or Stem or pool" 2>&1 | tail -4 > $O/tests.txt
for i in 1 2; do
for V in nopair pair; do
  echo "== $V" >> $O/stem_bench.txt
  BTX_LIB=build_variants/libbtx_$V.so python tools/stem_bench.py 64 2>&1 | grep -v amdgpu >> $O/stem_bench.txt
  BTX_LIB=build_variants/libbtx_$V.so python tools/stem_bench.py 1280 2>&1 | grep -v amdgpu >> $O/stem_bench.txt
done; done
cat $O/*.txt
