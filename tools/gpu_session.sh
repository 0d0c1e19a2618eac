#!/bin/bash
# One GPU-box session: everything it learns lands in gpurun_out/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{ ls /root/reference 2>&1 | head -3; rocminfo | grep -m2 gfx; nproc; lscpu | grep -m1 "Model name"; free -g | head -2; } > gpurun_out/box.txt 2>&1
for step in "$@"; do
  case "$step" in
    smoke)  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" ;;
    parity) timeout 1200 python tools/gpu_diag.py parity > gpurun_out/diag_parity.log 2>&1; echo "parity rc=$?" ;;
    pytest) timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" ;;
    pytest_all) timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" ;;
    perf)   timeout 900 python tools/gpu_diag.py perf --iters 5 > gpurun_out/diag_perf.log 2>&1; echo "perf rc=$?" ;;
    bench)  timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log ;;
    bench_nofuse) timeout 900 python bench.py --steps 10 --warmup 2 --no-fuse --no-cpu-baseline > gpurun_out/bench_nofuse.log 2>&1; echo "bench_nofuse rc=$?"; tail -1 gpurun_out/bench_nofuse.log ;;
    bench_diag) timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --per-step > gpurun_out/bench_diag1.log 2>&1; timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --per-step --no-launch-timing > gpurun_out/bench_diag2.log 2>&1; timeout 900 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --per-step > gpurun_out/bench_diag3.log 2>&1; echo "bench_diag rc=$?" ;;
    bench_f32) timeout 900 python bench.py --steps 5 --warmup 1 --prec f32 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; echo "bench_f32 rc=$?"; tail -1 gpurun_out/bench_f32.log ;;
    counters) rocprofv3 -L > gpurun_out/counters.txt 2>&1; echo "counters rc=$?" ;;
    pmc)    R="$PWD"; cd /tmp
            i=0
            for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
                       "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
                       "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum"; do
              i=$((i+1))
              timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$R/gpurun_out/pmc$i" -o pmc -- python "$R/tools/gpu_diag.py" one --prec bf16 --iters 6 > "$R/gpurun_out/pmc$i.log" 2>&1; echo "pmc$i rc=$?"
            done; cd "$R" ;;
    ablate) for d in 0 1 2 4 8 16 17 18 19 27 31; do echo "== BTX_DBG=$d"; BTX_DBG=$d timeout 300 python tools/gpu_diag.py timeone --prec bf16 --iters 20; done > gpurun_out/ablate.log 2>&1; echo "ablate rc=$?" ;;
    variants) for f in build_variants/libbtx_*.so; do for sh in 64,64,56,1,3 256,256,14,1,3; do echo -n "$(basename $f) "; BTX_LIB=$PWD/$f timeout 300 python tools/gpu_diag.py timeone --prec bf16 --iters 30 --shape $sh 2>&1 | grep shape; done; done > gpurun_out/variants.log 2>&1; echo "variants rc=$?" ;;
    patchab) for sh in 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3; do for v in "X=0" "BTX_PATCH_NW=8" "BTX_NO_PATCH=1"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done; done > gpurun_out/patchab.log 2>&1; echo "patchab rc=$?" ;;
    dmaab) for sh in 3,64,224,2,7 64,128,56,2,3 64,128,56,2,1 128,256,28,2,3 256,512,14,2,3 256,512,14,2,1; do for v in "X=0" "BTX_DMA_NW=8"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py timeone --prec bf16 --iters 30 --shape $sh 2>&1 | grep -E "shape|rror"; done; done > gpurun_out/dmaab.log 2>&1; echo "dmaab rc=$?" ;;
    pytest_contract) timeout 900 python -m pytest tests/test_gpu_contract.py -m gpu -q -x > gpurun_out/pytest_contract.log 2>&1; echo "pytest_contract rc=$?"; tail -15 gpurun_out/pytest_contract.log ;;
    ptrace) for sh in 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3 64,128,56,2,3 64,128,56,2,1; do echo "== $sh"; BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape $sh 2>&1 | grep -v amdgpu.ids; done > gpurun_out/ptrace.log 2>&1; echo "ptrace rc=$?" ;;
    ptrace1) for bs in 16 32 64; do echo "== bs $bs"; BTX_LIB=$PWD/build_variants/libbtx_trace.so timeout 300 python tools/gpu_diag.py trace --prec bf16 --shape 64,64,56,1,3 --bs $bs 2>&1 | grep -v "amdgpu.ids\|wave "; done > gpurun_out/ptrace1.log 2>&1; echo "ptrace1 rc=$?" ;;
    ksab) for sh in 256,256,14,1,3 512,512,7,1,3 128,256,28,2,3 256,512,14,2,3 256,512,14,2,1 128,128,28,1,3; do for v in 256 384 512 768; do echo -n "SLOTS4=$v "; BTX_SLOTS4=$v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done; done > gpurun_out/ksab.log 2>&1; echo "ksab rc=$?" ;;
    gtimes) for sh in 3,64,224,2,7 64,64,56,1,3 64,128,56,2,3 64,128,56,2,1 128,128,28,1,3 128,256,28,2,3 128,256,28,2,1 256,256,14,1,3 256,512,14,2,3 256,512,14,2,1 512,512,7,1,3; do timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done > gpurun_out/gtimes.log 2>&1; echo "gtimes rc=$?" ;;
    stemab) for v in "X=0" "BTX_STEM_NW=8" "BTX_NO_STEM=1"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape 3,64,224,2,7 2>&1 | grep -E "shape|rror"; done > gpurun_out/stemab.log 2>&1; echo "stemab rc=$?" ;;
    kprof_stem) R="$PWD"; cd /tmp; timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/kprof" -o kp -- python "$R/tools/gpu_diag.py" one --prec bf16 --iters 10 --shape 3,64,224,2,7 > /dev/null 2>&1; cd "$R"; echo "kprof rc=$?" ;;
    gvariants) for f in build_variants/libbtx_*.so; do for sh in 64,64,56,1,3 256,256,14,1,3; do echo -n "$(basename $f) "; BTX_LIB=$PWD/$f timeout 300 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep shape; done; done > gpurun_out/gvariants.log 2>&1; echo "gvariants rc=$?" ;;
    mi4ab) for sh in 64,64,56,1,3 128,128,28,1,3 256,256,14,1,3 512,512,7,1,3; do for v in "X=0" "BTX_PATCH_MI=4"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done; done > gpurun_out/mi4ab.log 2>&1; echo "mi4ab rc=$?"; BTX_PATCH_MI=4 timeout 900 python -m pytest tests/test_gpu_contract.py -m gpu -q -x 2>&1 | tail -3 ;;
    redab) for sh in 256,256,14,1,3 512,512,7,1,3 256,512,14,2,3 256,512,14,2,1; do for v in "X=0" "BTX_NO_FUSED_REDUCE=1"; do echo -n "$v "; env $v timeout 120 python tools/gpu_diag.py gtime --prec bf16 --shape $sh 2>&1 | grep -E "shape|rror"; done; done > gpurun_out/redab.log 2>&1; echo "redab rc=$?" ;;
    kstats) R="$PWD"; cd /tmp; for v in main a31 a63 a1 a4; do lib="$R/build_variants/libbtx_$v.so"; [ $v = main ] && lib="$R/bayesian_torch_amd/libbtx.so"; BTX_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/ks_$v" -o ks -- python "$R/tools/gpu_diag.py" one --prec bf16 --iters 10 > /dev/null 2>&1; echo "== $v"; find "$R/gpurun_out/ks_$v" -name "*kernel_stats.csv" | head -1 | xargs cat | grep -E "patch|presample|Name" | cut -c1-200; done > "$R/gpurun_out/kstats.log" 2>&1; cd "$R"; echo "kstats rc=$?" ;;
    prof)   cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof.log" 2>&1; echo "prof rc=$?"; cd "$OLDPWD" ;;
  esac
done
tail -5 gpurun_out/smoke.log 2>/dev/null
