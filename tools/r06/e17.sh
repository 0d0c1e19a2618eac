# E17 (round 6): instruction-cache counters of the tap-unrolled kernels (133 KB / 76 KB / 135 KB of code against a 64-KB instruction cache)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for T in Flipout Reparameterization; do
for S in 64,64,56,1,3 128,128,28,1,3 512,512,7,1,3; do
  d=$O/pmc_${T}_$(echo $S | tr , _)
  timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $d -o pmc -- python $R/tools/gpu_diag.py one --typ $T --throughput-plan --prec bf16 --iters 6 --bs 1280 --shape $S > $d.log 2>&1
  echo "== $T $S" >> $O/icache.txt
  (cd $R && python tools/pmc_report.py "$d/pmc_results.db" --kernel taps 2>&1 | tail -12) >> $O/icache.txt
  rm -rf $d
done; done
cat $O/icache.txt
