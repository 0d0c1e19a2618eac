#!/usr/bin/env python3
"""One command per file under profiles/ (round 3 on; the files of the current round carry PREFIX).  Run the collecting steps on a GPU box, e.g.

    gpurun --timeout 2400 -- 'python tools/refresh_profiles.py bench trace pmc'     # writes gpurun_out/profiles/*
    python tools/refresh_profiles.py install                                        # here: gpurun_out/profiles/* -> profiles/

steps (each writes the files named in its docstring, with the command that produced them in the header):
  bench       profiles/<PREFIX>_bench_line.json (+ _summary.txt)            python bench.py (the driver's default invocation)
  trace       profiles/<PREFIX>_bench_kernel_trace_stats.txt, _lanes1.txt   rocprofv3 --kernel-trace of the bench command
  pmc         profiles/<PREFIX>_pmc_taps_lanes.txt                          SQ counters of contract_taps_kernel, many-tiles regime
  cfg3        profiles/<PREFIX>_cfg3_kernel_trace_stats.txt, _cfg3_pmc_taps.txt, _cfg3_bench_line.json   BASELINE cfg3 (Reparameterization):
                                                                            kernel trace of its bench command, SQ counters of its wide kernel
  phase       profiles/<PREFIX>_phase_timers.txt, r03_phase_timers_sustained.txt   block phase timers (trace build)
  ablation    profiles/<PREFIX>_kloop_ablation.txt                          what the K loop pays for (ablation builds)
  ubench      profiles/<PREFIX>_mfma_mix_ubench.txt                         tools/ubench/mfma_mix.hip
  persistent  profiles/<PREFIX>_persistent_kbench.txt                       persistent kernel A/B (tuning build)
  tall        profiles/<PREFIX>_tall_tiles_ab.txt                           tall-strip tiles A/B
  power       profiles/<PREFIX>_power_probe.txt                             socket power / clock under sustained launches
  train       profiles/<PREFIX>_train_step_captured_trace.txt               kernel trace of the captured training step
  wgrad       profiles/<PREFIX>_wgrad_bench.txt, _dgrad_bench.txt           weight-gradient paths / stride-2 data gradients per layer
The measurement builds (build_variants/libbtx_{tune,trace,abl*}.so) are made by tools/build_variants.sh when missing.
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "profiles")
ENV = dict(os.environ, TMPDIR="/tmp")
PREFIX = "r06"  # file-name prefix of the round being measured
SHAPES = ["64,64,56,1,3", "128,128,28,1,3", "256,256,14,1,3", "512,512,7,1,3"]


def sh(cmd, env=None, timeout=1500, cwd=ROOT):
    r = subprocess.run(cmd, shell=True, capture_output=True, text=True, env=dict(ENV, **(env or {})), timeout=timeout, cwd=cwd)
    return "\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l) + "\n"


def write(name, header, body):
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, name), "w").write("".join("# " + l + "\n" for l in header.strip().splitlines()) + body)
    print("wrote", os.path.join("gpurun_out/profiles", name))


def variant(name, flags):
    so = os.path.join(ROOT, "build_variants", "libbtx_%s.so" % name)
    if not os.path.exists(so):
        print(sh("bash tools/build_variants.sh %s '%s'" % (name, flags), timeout=3000))
    return so


def step_bench():
    out = sh("timeout -k 5 600 python bench.py --steps 20 --warmup 5", timeout=2400)  # the driver's invocation
    line = [l for l in out.splitlines() if l.startswith("{")][-1]  # the compact line (<= 6 KB), LAST on stdout
    c = json.loads(line)
    os.makedirs(OUT, exist_ok=True)
    json.dump(c, open(os.path.join(OUT, PREFIX + "_bench_line.json"), "w"), indent=1)
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_detail.json")))  # the full result of the same run
    json.dump(d, open(os.path.join(OUT, PREFIX + "_bench_detail.json"), "w"), indent=1)
    r = d["roofline"]
    body = "final stdout line: %d bytes\n" % len(line)
    body += "value %.1f %s  ms/step %.4f  = the median of the settled half of %d back-to-back regions (%.2f s); first five %s; all regions %s\n" % (
        d["value"], d["unit"], d["ms_per_step"], d["timed_regions"], d["timed_seconds"], ["%.3f" % v for v in d["ms_per_step_runs"]],
        json.dumps(d.get("ms_per_step_all_regions")))
    body += "sustained (2 s of replays): %s\n" % json.dumps(d.get("sustained"))
    body += "roofline.frac %.4f (contraction only %.4f)  e2e %.4f  sampling %.1f us per %d-lane launch\n" % (
        r["frac"], r["frac_contraction_only"], r["frac_e2e"], r["sampling_us_per_launch"], r["mc_samples_per_launch"])
    body += "traffic (%s): %s\n" % (r["traffic_source"], json.dumps({k: round(v["ratio"], 2) for k, v in (r["traffic_detail"] or {}).get("layers", {}).items()}))
    body += "parity: logits rel-L2 %s  KL rel err %s\n" % (d.get("logits_rel_l2_vs_unfused_f32"), d.get("kl_rel_err"))
    for row in r["per_launch"]:
        body += "  %-46s %8.1f us  %7.1f TFLOP/s  %5.2f TB/s  %s %.3f\n" % (row["launch"], row["us"], row["tflops"], row["tbs"], row["bound"], row["frac_incl_sampling"])
    for k, v in d.get("extra", {}).items():
        body += "%s: %s\n" % (k, json.dumps(v)[:600])
    body += "cpu_baseline: %s\n" % json.dumps(d.get("cpu_baseline"))[:400]
    write(PREFIX + "_bench_summary.txt", "python bench.py --steps 20 --warmup 5   (the driver's invocation; 1 MI355X; the compact JSON line is profiles/" + PREFIX +
          "_bench_line.json, the full result profiles/" + PREFIX + "_bench_detail.json)", body)


def step_trace():
    # --steps / --warmup are multiples of the lane count (20 = the driver's --steps 20): every contract_taps_kernel launch of
    # the traced run carries the same number of MC sample lanes, so the trace's average IS the bench's avg_launch_us
    tags = (("stats", ""), ("lanes1", " --lanes 1"))
    if os.environ.get("BTX_TRACE_STATS_ONLY"):  # (a short GPU budget: the 20-lane trace only)
        tags = tags[:1]
    for tag, extra in tags:
        d = os.path.join(ROOT, "gpurun_out", "r6_kt_" + tag)
        shutil.rmtree(d, ignore_errors=True)
        cmd = "python %s/bench.py --steps 20 --warmup 20 --no-extras --no-cpu-baseline --no-traffic --no-sustain%s" % (ROOT, extra)
        sh("timeout -k 5 400 rocprofv3 --kernel-trace --stats -d %s -o kt -- %s" % (d, cmd), cwd="/tmp", timeout=1200)
        db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        body = sh("python tools/trace_report.py %s" % db[0]) if db else "(no trace written)\n"
        write(PREFIX + "_bench_kernel_trace_%s.txt" % tag,
              "cd /tmp && rocprofv3 --kernel-trace --stats -- %s   (1 MI355X; tools/trace_report.py on the result)\n"
              "%s" % (cmd.replace(ROOT + "/", ""), "lanes 1: one MC sample per launch — the dominant kernel's average here is a single-sample launch"
                      if extra else "the bench's default for --steps 20: 20 MC samples per launch (lanes), the per-launch timing replays included"), body)
        shutil.rmtree(d, ignore_errors=True)  # the raw trace (tens of MB) stays on the box: gpurun merges at most 64 MiB back


def step_train():
    """profiles/<PREFIX>_train_step_captured_trace.txt: per-kernel totals and one replay's kernel sequence of autograd.GraphedTrainStep"""
    d = os.path.join(ROOT, "gpurun_out", "r5_train_trace")
    shutil.rmtree(d, ignore_errors=True)
    log = sh("timeout -k 5 240 rocprofv3 --kernel-trace -d %s -o t -- python %s/tools/train_profile.py --graph 20" % (d, ROOT), cwd="/tmp", timeout=400)
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    body = sh("python tools/trace_report.py %s --sequence 420" % db[0]) if db else "(no trace written)\n"
    rep = [l for l in log.splitlines() if "replays" in l]
    write(PREFIX + "_train_step_captured_trace.txt",
          "cd /tmp && rocprofv3 --kernel-trace -- python tools/train_profile.py --graph 20   (autograd.GraphedTrainStep on dnn_to_bnn(ResNet18) Flipout bs 64,\n"
          "bf16, hip_batchnorm with fused residual / ReLU: 3 warm-up steps + 20 replays = 23 steps in the trace; %s under the profiler;\n"
          "tools/trace_report.py: per-kernel totals, then the last 420 dispatches = one replay in launch order)" % (rep[-1].strip() if rep else "?"), body)
    shutil.rmtree(d, ignore_errors=True)


def step_wgrad():
    """profiles/<PREFIX>_wgrad_bench.txt, _dgrad_bench.txt: the weight-gradient paths and the stride-2 data gradients, layer by layer"""
    lib = variant("tune", "-DBTX_TUNING")
    write(PREFIX + "_wgrad_bench.txt", "BTX_LIB=build_variants/libbtx_tune.so python tools/wgrad_bench.py --ablate   (1 MI355X; columns: btx_contract_wgrad with f32\n"
          "atomics | btx_contract_wgrad_ws on the tap-per-workgroup kernel | btx_contract_wgrad_ws as shipped (all-taps kernel on the 3x3/s1 rows))",
          sh("python tools/wgrad_bench.py --ablate", env={"BTX_LIB": lib}, timeout=400))
    write(PREFIX + "_dgrad_bench.txt", "BTX_LIB=build_variants/libbtx_tune.so python tools/dgrad_bench.py   (1 MI355X; autograd._data_grad_hip of the strided ResNet18\n"
          "convolutions: transposed launch in raster order (BTX_NO_PAR_MAJOR=1) | parity-major order as shipped)",
          sh("python tools/dgrad_bench.py", env={"BTX_LIB": lib}, timeout=300))


def derived(rep):
    """MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); issued FLOP = SQ_INSTS_MFMA x 32768;
    LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
    import re
    v = {k: float(x) for k, x in re.findall(r"^\s+([A-Z_]+)\s+([0-9.e+]+)\s*$", rep, re.M)}
    try:
        return ("derived: MFMA busy %.1f %% of the SIMD-cycles of the launch; issued %.1f GFLOP; LDS bank-conflict cycles %.1f %% of "
                "SQ_LDS_IDX_ACTIVE; shader clock %.2f GHz (GRBM_GUI_ACTIVE / 8 / kernel time)\n" % (
                    100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0), v["SQ_INSTS_MFMA"] * 32768 / 1e9,
                    100.0 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
                    v["GRBM_GUI_ACTIVE"] / 8.0 / (float(re.findall(r"avg ([0-9.]+) us", rep)[-1]) * 1e3)))
    except Exception as e:  # noqa
        return "derived: (%s)\n" % e


def step_pmc():
    sets = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA",
            "GRBM_GUI_ACTIVE"]
    body = ""
    for shp in SHAPES[:2]:
        for i, s in enumerate(sets):
            d = os.path.join(ROOT, "gpurun_out", "r6_pmc%d_%s" % (i, shp.replace(",", "_")))
            shutil.rmtree(d, ignore_errors=True)
            sh("rocprofv3 --pmc %s --kernel-trace -d %s -o pmc -- python %s/tools/gpu_diag.py one --throughput-plan --prec bf16 --iters 6 "
               "--bs 1280 --shape %s" % (s, d, ROOT, shp), cwd="/tmp", timeout=400)
        rep = sh("python tools/pmc_report.py 'gpurun_out/r6_pmc*_%s/pmc_results.db' --kernel taps" % shp.replace(",", "_"))
        body += "== %s\n" % shp + derived(rep) + rep
        for d in glob.glob(os.path.join(ROOT, "gpurun_out", "r6_pmc*_%s" % shp.replace(",", "_"))):
            shutil.rmtree(d, ignore_errors=True)
    write(PREFIX + "_pmc_taps_lanes.txt",
          "rocprofv3 --pmc <set> --kernel-trace -- python tools/gpu_diag.py one --throughput-plan --prec bf16 --iters 6 --bs 1280 --shape <s>\n"
          "(one pass per counter set, collected THIS round into fresh gpurun_out/r6_pmc* directories; batch 1280 = the tiles of the bench's 20 MC sample lanes; contract_taps_kernel; tools/pmc_report.py)", body)


def step_cfg3():
    """BASELINE cfg3 — dnn_to_bnn(ResNet18) Reparameterization bs 64, 16 MC samples as lanes: the bench command's own line and
    per-launch table, its rocprofv3 kernel trace, and the SQ counters of contract_taps_kernel<bf16, Reparameterization, WIDE>"""
    cmd = "python %s/bench.py --type Reparameterization --steps 16 --warmup 16 --no-extras --no-cpu-baseline --no-traffic" % ROOT
    out = sh("timeout -k 5 400 " + cmd, timeout=900)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(OUT, PREFIX + "_cfg3_bench_line.json"), "w"), indent=1)
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_detail.json")))
    r = d["roofline"]
    body = "value %.1f %s  ms/step %.4f  (settled half of %d regions over %.2f s)  sustained %s\n" % (
        d["value"], d["unit"], d["ms_per_step"], d["timed_regions"], d["timed_seconds"], json.dumps(d.get("sustained")))
    body += "dominant kernel (13 stride-1 3x3 launches, sampling share included) %.4f of the bf16 MFMA peak (contraction only %.4f), e2e %.4f\n" % (
        r["frac"], r["frac_contraction_only"], r["frac_e2e"])
    for row in r["per_launch"]:
        body += "  %-46s %8.1f us  %7.1f TFLOP/s  %5.2f TB/s  %s %.3f\n" % (row["launch"], row["us"], row["tflops"], row["tbs"], row["bound"], row["frac_incl_sampling"])
    write(PREFIX + "_cfg3_per_launch.txt", cmd.replace(ROOT + "/", "") + "   (1 MI355X; per launch: 16 MC sample lanes, re-issued 10x in a hipGraph between HIP events)", body)
    dd = os.path.join(ROOT, "gpurun_out", "r6_kt_cfg3")
    shutil.rmtree(dd, ignore_errors=True)
    sh("timeout -k 5 400 rocprofv3 --kernel-trace --stats -d %s -o kt -- %s --no-sustain" % (dd, cmd), cwd="/tmp", timeout=1200)
    db = glob.glob(os.path.join(dd, "**", "*.db"), recursive=True)
    write(PREFIX + "_cfg3_kernel_trace_stats.txt", "cd /tmp && rocprofv3 --kernel-trace --stats -- %s --no-sustain   (tools/trace_report.py on the result)" % cmd.replace(ROOT + "/", ""),
          sh("python tools/trace_report.py %s" % db[0]) if db else "(no trace written)\n")
    shutil.rmtree(dd, ignore_errors=True)
    sets = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA",
            "GRBM_GUI_ACTIVE"]
    body = ""
    for shp in SHAPES[:3]:
        for i, s_ in enumerate(sets):
            d_ = os.path.join(ROOT, "gpurun_out", "r6_pmc3_%d_%s" % (i, shp.replace(",", "_")))
            shutil.rmtree(d_, ignore_errors=True)
            sh("rocprofv3 --pmc %s --kernel-trace -d %s -o pmc -- python %s/tools/gpu_diag.py one --typ Reparameterization --throughput-plan --prec bf16 "
               "--iters 6 --bs 1024 --shape %s" % (s_, d_, ROOT, shp), cwd="/tmp", timeout=400)
        rep = sh("python tools/pmc_report.py 'gpurun_out/r6_pmc3_*_%s/pmc_results.db' --kernel taps" % shp.replace(",", "_"))
        body += "== %s\n" % shp + derived(rep) + rep
        for d_ in glob.glob(os.path.join(ROOT, "gpurun_out", "r6_pmc3_*_%s" % shp.replace(",", "_"))):
            shutil.rmtree(d_, ignore_errors=True)
    write(PREFIX + "_cfg3_pmc_taps.txt",
          "rocprofv3 --pmc <set> --kernel-trace -- python tools/gpu_diag.py one --typ Reparameterization --throughput-plan --prec bf16 --iters 6 --bs 1024 --shape <s>\n"
          "(one pass per counter set, collected this round into fresh r6_pmc3_* directories; batch 1024 = the tiles of cfg3's 16 MC sample lanes;\n"
          "56x56: the narrow tile (one n-tile), 28x28 / 14x14: contract_taps_kernel<bf16, Reparameterization, WIDE>; tools/pmc_report.py)", body)


def step_phase():
    lib = variant("trace", "-DBTX_PT_TRACE -DBTX_TUNING")
    body = ""
    for shp in SHAPES[:2]:
        for v in ("X=0", "BTX_NO_TALL=1", "BTX_TAPS_TUNE=128 BTX_NO_TALL=1"):
            body += "== %s %s\n" % (shp, v) + "".join(
                l + "\n" for l in sh("env %s python tools/gpu_diag.py trace --prec bf16 --shape %s" % (v, shp), env={"BTX_LIB": lib}).splitlines()
                if " wave " not in l and "column 7" not in l)
    write(PREFIX + "_phase_timers.txt", "BTX_LIB=build_variants/libbtx_trace.so python tools/gpu_diag.py trace --prec bf16 --shape <s>  (batch 64, one launch;\n"
          "X=0: tall-strip tiles where the plan takes them, BTX_NO_TALL=1: plain tiles, BTX_TAPS_TUNE=128: prologue sub-stamps)", body)
    body = ""
    for shp in SHAPES[:2]:
        for v in ("BTX_PERSIST=1", "X=0"):
            body += "== %s %s (600 warm launches)\n" % (shp, v) + "".join(
                l + "\n" for l in sh("env %s BTX_NO_TALL=1 python tools/gpu_diag.py trace --throughput-plan --bs 256 --warm 600 --prec bf16 --shape %s" % (v, shp),
                                     env={"BTX_LIB": lib}).splitlines() if " wave " not in l and "column 7" not in l)
    write(PREFIX + "_phase_timers_sustained.txt", "BTX_LIB=build_variants/libbtx_trace.so [BTX_PERSIST=1] BTX_NO_TALL=1 python tools/gpu_diag.py trace --throughput-plan --bs 256\n"
          "--warm 600 --prec bf16 --shape <s>: block phase timers and the shader clock (s_memtime / s_memrealtime) under sustained load,\n"
          "persistent kernel (contract_taps3_kernel) against contract_taps_kernel.  Reading: per tile the persistent kernel needs fewer\n"
          "cycles (56x56: (K loops + store sides) / 7 tiles ~ 29k against ~35k for a one-tile block) but runs at a lower clock under the\n"
          "higher duty (1.9 against 2.2 GHz): the launch takes the same time (r03_persistent_kbench.txt, r03_persistent_ab.txt)", body)


def kbench(lib, envs, shapes, bs=256, env=None):
    return sh("python tools/kbench.py --throughput-plan --env %s --bs %d --rounds 3 --reps 10 --shapes %s" % (" ".join(envs), bs, " ".join(shapes)),
              env=dict({"BTX_LIB": lib}, **(env or {})), timeout=900)


def step_ablation():
    body = ""
    for name, flags in (("tune", "-DBTX_TUNING"), ("abl4", "-DBTX_TUNING -DBTX_PT_ABL=4"), ("abl16", "-DBTX_TUNING -DBTX_PT_ABL=16"),
                        ("abl2", "-DBTX_TUNING -DBTX_PT_ABL=2"), ("abl22", "-DBTX_TUNING -DBTX_PT_ABL=22")):
        body += "## %s\n" % name + kbench(variant(name, flags), ["-"], [SHAPES[0], SHAPES[1], SHAPES[3]], env={"BTX_NO_TALL": "1"})
    write(PREFIX + "_kloop_ablation.txt", "BTX_NO_TALL=1 BTX_LIB=build_variants/libbtx_<v>.so python tools/kbench.py --throughput-plan --env - --bs 256 ...\n"
          "builds with -DBTX_PT_ABL=<bits>: 4 no weight/patch DMA in the K loop, 16 no s_in masks, 2 no LDS fragment reads, 22 all three\n"
          "(results wrong by construction; time only).  Reading (128->128, 28x28): no DMA -19 %, no masks -7 %, no fragment reads -27 %,\n"
          "none of the three: the floor set by MFMA issue, barriers, prologue and store side (118 GFLOP at ~2.1 GHz = 54 us of matrix time)", body)


def step_ubench():
    sh("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_mix tools/ubench/mfma_mix.hip", timeout=900)
    write(PREFIX + "_mfma_mix_ubench.txt", "hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_mix tools/ubench/mfma_mix.hip && tools/ubench/mfma_mix\n"
          "The instruction mix of one K-stage of the Flipout tap kernel (LDS fragment reads, s_in masks, mean + delta MFMAs, a barrier; NO\n"
          "global traffic) per register-tile shape, 12 x 50 ms launches per line, two rounds; TFLOP/s from HIP events (the cycles/stage\n"
          "column is block 0's s_memtime delta: only meaningful for the 1-block/CU lines).\n"
          "Reading: A, the mix of contract_taps_kernel, sustains ~1.75 PFLOP/s (0.70 of the nominal bf16 peak) at 2.03 GHz — what the K loop\n"
          "could deliver if weight/patch DMA, real patch addressing and tile prologue/store side were free (the kernel: 0.85-1.05).  The\n"
          "512-register 1-wave/SIMD shapes with AGPR accumulators: 4x2 tiles 0.93x, 2x4 tiles 1.03x of A (a single wave per SIMD exposes\n"
          "the mask VALU and LDS latency a second wave hides).  Masks cost 10 % (D vs A), half the fragment reads another 8 % (E vs D).\n"
          "G/H/I add the kernel's DMA streams: weight tiles from L2 (8 KiB per block-stage) cost 4-5 %; streaming another 4 KiB per\n"
          "block-stage from HBM (about twice the kernel's own activation traffic) costs 14 % and pulls the clock from 1.97 to 1.65 GHz:\n"
          "at the package power limit HBM bytes are paid for in clock, whether or not their latency is hidden.",
          sh("tools/ubench/mfma_mix", timeout=600))


def step_persistent():
    lib = variant("tune", "-DBTX_TUNING")
    write(PREFIX + "_persistent_kbench.txt", "BTX_NO_TALL=1 BTX_LIB=build_variants/libbtx_tune.so python tools/kbench.py --throughput-plan --env BTX_PERSIST=1 - --bs 256 ...\n"
          "(contract_taps3_kernel against contract_taps_kernel; '== variant0: True' = bit-identical results)",
          sh("python tools/kbench.py --throughput-plan --env BTX_PERSIST=1 - --bs 256 --rounds 3 --reps 10 --shapes %s" % " ".join(SHAPES),
             env={"BTX_LIB": lib, "BTX_NO_TALL": "1"}, timeout=900))


def step_tall():
    lib = variant("tune", "-DBTX_TUNING")
    body = ""
    for bs in (64, 256, 512):
        body += "## batch %d\n" % bs + kbench(lib, ["-", "BTX_NO_TALL=1"], SHAPES, bs=bs)
    write(PREFIX + "_tall_tiles_ab.txt", "BTX_LIB=build_variants/libbtx_tune.so python tools/kbench.py --throughput-plan --env - BTX_NO_TALL=1 --bs <b> ...\n"
          "(tall-strip tiles where the plan takes them against plain tiles)", body)


def step_power():
    """profiles/<PREFIX>_power_probe.txt: socket power and sclk (rocm-smi, every ~0.3 s) while 60 000 back-to-back launches of one
    layer run — contract_taps_kernel against the persistent kernel; needs the tuning build"""
    import threading
    import time
    lib = variant("tune", "-DBTX_TUNING")
    body = sh("rocm-smi --showmaxpower 2>/dev/null | grep -i 'max graphics'")
    for shp in SHAPES[:2]:
        for tag, env in (("contract_taps_kernel", {}), ("persistent (BTX_PERSIST=1)", {"BTX_PERSIST": "1"})):
            samples, done = [], []

            def poll():
                while not done:
                    o = sh("rocm-smi --showpower --showclocks", timeout=20)
                    try:
                        w = float([l for l in o.splitlines() if "Socket Graphics" in l][0].split(":")[-1])
                        c = float([l for l in o.splitlines() if "sclk" in l][0].split("(")[1].split("Mhz")[0])
                        samples.append((w, c))
                    except Exception:  # noqa
                        pass
                    time.sleep(0.3)
            th = threading.Thread(target=poll)
            th.start()
            out = sh("python tools/gpu_diag.py timeone --throughput-plan --prec bf16 --bs 256 --iters 60000 --shape %s" % shp,
                     env=dict({"BTX_LIB": lib, "BTX_NO_TALL": "1"}, **env), timeout=400)
            done.append(1)
            th.join()
            busy = [(w, c) for w, c in samples if w > 500]
            body += "%s  %-28s %s" % (shp, tag, [l for l in out.splitlines() if "us / launch" in l][-1].split(":")[-1])
            if busy:
                body += "   busy samples %d: mean power %.0f W (max %.0f), mean sclk %.0f MHz\n" % (
                    len(busy), sum(w for w, _ in busy) / len(busy), max(w for w, _ in busy), sum(c for _, c in busy) / len(busy))
            else:
                body += "   (no busy samples)\n"
    write(PREFIX + "_power_probe.txt", "BTX_LIB=build_variants/libbtx_tune.so BTX_NO_TALL=1 [BTX_PERSIST=1] python tools/gpu_diag.py timeone --throughput-plan\n"
          "--prec bf16 --bs 256 --iters 60000 --shape <s>, with `rocm-smi --showpower --showclocks` polled meanwhile (1 MI355X).\n"
          "Reading: both kernels run at 92-99 % of the 1400 W package limit with the clock throttled to 1.86-2.03 GHz (peak 2.4 GHz;\n"
          "boxes of the pool differ by a few percent): the operating point is set by power, and a kernel that keeps the matrix pipe\n"
          "busier per cycle (the persistent one) gets fewer cycles per second.  tools/ubench/mfma_mix draws 1260-1340 W at 2.3-2.4 GHz.", body)


def step_install():
    n = 0
    for f in sorted(glob.glob(os.path.join(OUT, "*"))):
        shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
        n += 1
    print("installed %d file(s) into profiles/" % n)


if __name__ == "__main__":
    steps = sys.argv[1:] or ["install"]
    for s in steps:
        globals()["step_" + s]()
