#!/bin/bash
# Phase timers and ablations of contract_gemm8_kernel on a GPU box -> gpurun_out/profiles/r04_gemm8_phase_timers.txt
# needs:  T="-DBTX_TUNING -DBTX_PT_TRACE -DBTX_EP_TRACE"
#         tools/build_variants.sh g8t "$T" g8t_s1 "$T -DBTX_EP_TRACE2" g8t_skel "$T -DBTX_PT_ABL=534" g8t_nomask "$T -DBTX_PT_ABL=16" tune "-DBTX_TUNING"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles
O=gpurun_out/profiles/r04_gemm8_phase_timers.txt
G="python tools/gpu_diag.py g8trace --prec bf16 --bs 512 --warm 40"
F='clock|waves|gap|g8trace'
{
echo "# contract_gemm8_kernel (csrc/btx_contract_gemm8.h): per-wave phase timers (s_memtime, shader cycles), one traced launch after"
echo "# 40 warm-up launches, batch 512, bf16.  tools/g8_phase_report.sh; libraries built by tools/build_variants.sh with"
echo "# -DBTX_TUNING -DBTX_PT_TRACE -DBTX_EP_TRACE (+ the switch named per section)."
echo "# 'full store side' = BN affine + bf16 residual + ReLU (a ResNet50 expand convolution); 'bare' = no per-channel constants."
echo
echo "## the shipped schedule"
for s in 256,1024,14,1,1 128,512,28,1,1 512,2048,7,1,1; do BTX_LIB=build_variants/libbtx_g8t.so timeout 120 $G --shape $s 2>&1 | grep -E "$F"; done
for s in 512,128,28,1,1 1024,256,14,1,1; do BTX_LIB=build_variants/libbtx_g8t.so timeout 120 $G --shape $s --bare 2>&1 | grep -E "$F"; done
echo
echo "## stage 1 of the store side split (-DBTX_EP_TRACE2): columns 'prologue' | 'K loop' | 'store stage 1' hold"
echo "## head (parameters, residual requests, L2 touch) | body (fold, affine, LDS writes) | LDS drain"
BTX_LIB=build_variants/libbtx_g8t_s1.so timeout 120 $G --shape 256,1024,14,1,1 2>&1 | grep -E "$F"
echo
echo "## the direct (register-resident) store side, BTX_G8_DIRECT=1: the whole store side is reported as stage 2 (+ drain)"
BTX_G8_DIRECT=1 BTX_LIB=build_variants/libbtx_g8t.so timeout 120 $G --shape 256,1024,14,1,1 2>&1 | grep -E "$F"
echo
echo "## K-loop ablations on 512 -> 128 at 28x28 (bare): -DBTX_PT_ABL=16 no s_in masks; =534 MFMAs + barriers + delta reads only"
for v in g8t g8t_nomask g8t_skel; do echo "# $v"; BTX_LIB=build_variants/libbtx_$v.so timeout 120 $G --shape 512,128,28,1,1 --bare 2>&1 | grep -E "clock|waves 0"; done
echo
echo "## layer call, staged vs direct store side and gemm8 vs contract_dma_kernel (tools/kbench.py, batch 512, hipGraph of 20 calls)"
BTX_LIB=build_variants/libbtx_tune.so timeout 400 python tools/kbench.py --bs 512 --shapes 256,1024,14,1,1 512,128,28,1,1 1024,256,14,1,1 128,512,28,1,1 512,2048,7,1,1 256,512,56,2,1 --env - BTX_G8_DIRECT=1 BTX_NO_GEMM8=1 2>&1 | grep -E "Flipout"
} > $O 2>&1
echo wrote $O
