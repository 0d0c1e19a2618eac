#!/usr/bin/env python3
"""Shader clock and package power per KERNEL: each layer call of the benched step (batch = 20 lanes x 64 images as one batch,
throughput plan) replayed back to back from a hipGraph for ~1.2 s while the amdgpu hwmon files of the device are polled
(bench.ClockPowerSampler).  Tells a launch that runs at the package power limit (time = energy / cap: only fewer joules per tile
help) from one that leaves power on the table (bound by issue / LDS / latency: shorter critical paths help).

usage: python tools/power_probe.py [--bs 1280] [--seconds 1.2] [--prec bf16] [shape ...]   shape = cin,cout,hw,stride,k[,pool]"""
import argparse
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

DEFAULT = ["3,64,224,2,7,pool", "64,64,56,1,3", "64,128,56,2,3", "64,128,56,2,1", "128,128,28,1,3", "128,256,28,2,3", "256,256,14,1,3",
           "256,512,14,2,3", "512,512,7,1,3"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=DEFAULT)
    ap.add_argument("--bs", type=int, default=1280)
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from bayesian_torch_amd import layers as L
    from bayesian_torch_amd import functional as BF
    BF._CONCURRENT = True
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if a.prec == "bf16" else torch.float32
    print("%-22s %10s %9s %9s %8s %8s" % ("shape (bs %d)" % a.bs, "us/call", "TFLOP/s", "of peak", "GHz", "W"))
    for sh in a.shapes:
        f = sh.split(",")
        cin, cout, hw, stride, k = [int(v) for v in f[:5]]
        pool = len(f) > 5 and f[5] == "pool"
        bs = a.bs if not pool else a.bs // 20 * 20 // 20  # the stem's input is shared by the lanes: run it with lanes below
        torch.manual_seed(0)
        layer = L.Conv2dFlipout(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
        layer.precision = a.prec
        if pool:
            import bayesian_torch_amd as bt
            x = torch.randn(64, cin, hw, hw, device=dev).to(act)
            scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            lanes = a.bs // 64
            bt.set_sample_lanes(layer, list(range(lanes)), batch=64)
            call = lambda i: layer.forward_fused(x, scale, shift, None, True, pool=True)  # noqa: E731
            m_rows = lanes * 64 * (hw // stride) ** 2
        else:
            x = torch.randn(a.bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
            call = lambda i: layer._forward_hip(x, sample_idx=i)  # noqa: E731
            m_rows = a.bs * (hw // stride) ** 2
        side = torch.cuda.Stream(dev)
        with torch.no_grad():
            with torch.cuda.stream(side):
                for i in range(2):
                    call(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(a.reps):
                    call(i)
            g.replay()
            torch.cuda.synchronize()
            n = 0
            with bench.ClockPowerSampler(0) as smp:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                while time.perf_counter() - t0 < a.seconds:
                    for _ in range(4):
                        g.replay()
                        n += a.reps
                    torch.cuda.synchronize()
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
        s = smp.summary()
        fl = 2.0 * 2 * m_rows * cout * cin * k * k
        tf = fl / (us * 1e-6) / 1e12
        print("%-22s %10.1f %9.0f %9.3f %8s %8s" % (sh, us, tf, tf / bench.MFMA_PEAK_TFLOPS[a.prec],
                                                    "%.3f" % s["shader_clock_ghz"] if s["shader_clock_ghz"] else "-",
                                                    "%.0f" % s["package_w"] if s["package_w"] else "-"), flush=True)
        del g


if __name__ == "__main__":
    main()
