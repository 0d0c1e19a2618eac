#!/usr/bin/env python3
"""Regenerates profiles/ from the last GPU session's gpurun_out/ (steps: prof, pmc, gtimes, ptrace of tools/gpu_session.sh)."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def run(cmd):
    return subprocess.run(cmd, shell=True, capture_output=True, text=True).stdout


hdr = ("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline   (1 MI355X, ROCm 7.2; 12 pre-warm +\n"
       "# 2 warm-up + 5 timed graph replays + 5 eager re-issues with per-launch events + capture warm-ups; tools/trace_report.py)\n")
open("profiles/r01_bench_kernel_trace_stats.txt", "w").write(hdr + run("python tools/trace_report.py gpurun_out/prof/bench_results.db"))
hdr = ("# rocprofv3 --pmc <set> --kernel-trace -- python tools/gpu_diag.py one --prec bf16 --iters 6   (ResNet18 layer1: 3x3 conv 64->64,\n"
       "# 56x56, batch 64, Flipout, bf16: the most frequent launch of the bench step; one --pmc pass per counter set; tools/pmc_report.py)\n"
       "# counters slow the kernel down (the plain kernel trace has it at ~47 us)\n")
pm = run("python tools/pmc_report.py 'gpurun_out/pmc*/pmc_results.db' --kernel patch")
open("profiles/r01_pmc_layer1_conv3x3_flipout_bf16.txt", "w").write(hdr + pm + run("python tools/pmc_report.py 'gpurun_out/pmc1/pmc_results.db' --kernel presample"))
hdr = ("# GPU time of ONE layer call (sampling pre-pass + contraction + split-K reduce + the input packing of the stem),\n"
       "# 20 calls captured in a hipGraph and replayed (tools/gpu_diag.py gtime): no host launch overhead in the number.\n"
       "# shape = cin,cout,hw,stride,k ; Flipout, bf16 activations + bf16 MFMA, batch 64; TFLOP/s = 2*2*M*N*K / time\n")
open("profiles/r01_per_layer_resnet18_bs64.txt", "w").write(hdr + open("gpurun_out/gtimes.log").read())
if os.path.exists("gpurun_out/ptrace.log"):
    hdr = ("# per-wave phase timers (s_memtime, ~2.4 GHz ticks; -DBTX_PT_TRACE build, tools/gpu_diag.py trace): mean/min/max over all waves\n"
           "# of one launch.  A->B = DMA issue + fragment reads + MFMA issue, B->C = timer read + vmcnt wait, C->D = barrier wait (summed over\n"
           "# the K stages of the block).  shape = cin,cout,hw,stride,k ; Flipout bf16 batch 64\n")
    tr = "".join(l for l in open("gpurun_out/ptrace.log") if " wave " not in l and "kernel span" not in l and "column 7" not in l)
    open("profiles/r01_phase_timers.txt", "w").write(hdr + tr)
f = float(re.search(r"FETCH_SIZE\s+([0-9.e+]+)", pm).group(1))
w = float(re.search(r"WRITE_SIZE\s+([0-9.e+]+)", pm).group(1))
j = {"hbm_bytes_per_launch": (2 * f + w) * 1024.0,
     "launch": "contract_patch_kernel<bf16,Flipout,4 waves> on ResNet18 layer1 3x3 conv, bs 64 (the most frequent launch of the bench step)",
     "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
     "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md HBM "
             "section); separate --pmc passes, rocprofv3",
     "algorithmic_bytes_per_launch": 51675136}
j["ratio"] = j["hbm_bytes_per_launch"] / j["algorithmic_bytes_per_launch"]
json.dump(j, open("profiles/pmc_traffic.json", "w"), indent=1)
print(open("profiles/pmc_traffic.json").read())
