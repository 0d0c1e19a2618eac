#!/usr/bin/env python3
"""(needs a library built with -DBTX_TUNING: `bash tools/build_variants.sh tune "-DBTX_TUNING"`, then
BTX_LIB=build_variants/libbtx_tune.so — the shipped libbtx.so ignores the A/B environment variables)

In-process A/B of kernel variants on the ResNet18 / ResNet50 convolution shapes: GPU time of one layer call (all its
kernels), `reps` calls captured in a hipGraph and replayed (no host launch overhead), interleaved rounds.

usage: python tools/kbench.py [--shapes s1 ...] [--env "A=1,B=2" ...] [--rounds 3] [--typ Flipout] [--prec bf16]
  a shape is cin,cout,hw,stride,k ; an --env entry is a comma-separated list of VAR=VALUE set for that variant
  ("-" = no variables).  Only variables the library re-reads per call (BTX_NO_TAPS, ...) can be toggled in-process.
"""
import argparse
import os
import sys
import warnings

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

RN18_S1 = ["64,64,56,1,3", "128,128,28,1,3", "256,256,14,1,3", "512,512,7,1,3"]
RN18_ALL = ["3,64,224,2,7", "64,64,56,1,3", "64,128,56,2,3", "64,128,56,2,1", "128,128,28,1,3", "128,256,28,2,3",
            "128,256,28,2,1", "256,256,14,1,3", "256,512,14,2,3", "256,512,14,2,1", "512,512,7,1,3"]


def capture(layer, x, reps):
    dev = x.device
    side = torch.cuda.Stream(dev)
    with torch.no_grad():
        with torch.cuda.stream(side):
            for i in range(3):
                layer._forward_hip(x, sample_idx=i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                y = layer._forward_hip(x, sample_idx=i)
        g.replay()
        torch.cuda.synchronize()
    return g, y


def timeit(g, reps, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=RN18_S1)
    ap.add_argument("--env", nargs="*", default=["-", "BTX_NO_TAPS=1"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--typ", default="Flipout")
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--throughput-plan", action="store_true", help="BTX_FLAG_CONCURRENT on every launch: the K split of launches that share the GPU / carry MC sample lanes")
    a = ap.parse_args()
    if a.shapes == ["all"]:
        a.shapes = RN18_ALL
    from bayesian_torch_amd import layers as L
    from bayesian_torch_amd import functional as BF
    BF._CONCURRENT = bool(a.throughput_plan)
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if a.prec == "bf16" else torch.float32
    for sh in a.shapes:
        cin, cout, hw, stride, k = [int(v) for v in sh.split(",")]
        torch.manual_seed(0)
        layer = getattr(L, "Conv2d" + a.typ)(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
        layer.precision = a.prec
        x = torch.randn(a.bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
        graphs, outs = [], []
        for ev in a.env:
            kv = [] if ev == "-" else [e.split("=", 1) for e in ev.split(",")]
            for k_, v_ in kv:
                os.environ[k_] = v_
            g, y = capture(layer, x, a.reps)
            for k_, _ in kv:
                del os.environ[k_]
            graphs.append(g)
            outs.append(y.float().clone())
        ho = outs[0].shape[2]
        fl = 2.0 * a.bs * ho * ho * cout * cin * k * k * (2 if a.typ == "Flipout" else 1)
        res = [[] for _ in a.env]
        for _ in range(a.rounds):
            for i, g in enumerate(graphs):
                res[i].append(timeit(g, a.reps))
        for i, ev in enumerate(a.env):
            us = sorted(res[i])
            same = bool(torch.equal(outs[i], outs[0]))
            print("%-16s %-5s %-7s %-28s min %7.1f med %7.1f us  %7.1f TFLOP/s (min)  == variant0: %s" % (
                sh, a.prec, a.typ[:7], ev, us[0], us[len(us) // 2], fl / (us[0] * 1e-6) / 1e12, same), flush=True)


if __name__ == "__main__":
    main()
