"""Weight-gradient launches of ResNet18 (batch 64, bf16 activations, Flipout) one by one: the f32-atomics path of rounds 2-5, the
chunk-slab path of the same kernel, and the all-taps kernel (csrc/btx_wgrad_taps.h) — time per call (HIP events, whole call:
memsets / kernel / slab reduction) and agreement between the paths.

    BTX_LIB=build_variants/libbtx_tune.so python tools/wgrad_bench.py [--iters 20] [--sweep]

The tuning build reads BTX_WGRAD_NO_TAPS3 / BTX_WGRAD_SLAB_WGS / BTX_WGRAD_T3_WGS per call; the shipped library ignores them
(then the "slab, tap-per-workgroup" column repeats the all-taps numbers on the 3x3/s1 rows)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import _lib, functional as BF  # noqa: E402

SHAPES = [  # (label, cin, cout, H, k, stride)
    ("3x3 s1   64->64  56", 64, 64, 56, 3, 1), ("3x3 s1 128->128  28", 128, 128, 28, 3, 1),
    ("3x3 s1 256->256  14", 256, 256, 14, 3, 1), ("3x3 s1 512->512   7", 512, 512, 7, 3, 1),
    ("3x3 s2  64->128  56", 64, 128, 56, 3, 2), ("3x3 s2 128->256  28", 128, 256, 28, 3, 2),
    ("3x3 s2 256->512  14", 256, 512, 14, 3, 2), ("1x1 s2  64->128  56", 64, 128, 56, 1, 2),
    ("1x1 s2 128->256  28", 128, 256, 28, 1, 2), ("1x1 s2 256->512  14", 256, 512, 14, 1, 2),
]


def run(op, x, dy, iters, atomics, env):
    for k in ("BTX_WGRAD_NO_TAPS3", "BTX_WGRAD_SLAB_WGS", "BTX_WGRAD_T3_WGS", "BTX_WGRAD_T3_ABL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    BF.WGRAD_ATOMICS = atomics
    w_shape = (op.out_channels, op.in_channels) + tuple(op.kernel[1:])
    call = lambda: BF.wgrad_hip(_lib.KIND_FLIPOUT, x, dy, op, 1234, 5, 7, w_shape, raw=True)  # noqa: E731
    out = call()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        t0.record()
        for _ in range(iters):
            call()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / iters)
    BF.WGRAD_ATOMICS = False
    return best, out[0].clone(), out[1].clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sweep", action="store_true", help="workgroup-count sweeps of the slab paths")
    ap.add_argument("--ablate", action="store_true", help="all-taps kernel without its slab stores / MFMA section (time only)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("# us per btx_contract_wgrad* call, batch %d, bf16, Flipout (min of 3 x %d calls)" % (a.batch, a.iters))
    print("%-22s %10s %10s %10s   %s" % ("layer", "atomics", "slab/tap", "all-taps", "max rel diff vs atomics (mu, delta)"))
    tot = [0.0, 0.0, 0.0]
    for label, cin, cout, hw, k, s in SHAPES:
        op = BF.OpDesc(2, cin, cout, k, s, k // 2)
        ho = op.out_spatial((1, hw, hw))[1]
        x = torch.randn(a.batch, cin, hw, hw, device=dev).relu_().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = (torch.randn(a.batch, cout, ho, ho, device=dev) * 0.01).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ta, ma, da = run(op, x, dy, a.iters, True, {})
        ts, ms, ds = run(op, x, dy, a.iters, False, {"BTX_WGRAD_NO_TAPS3": "1"})
        tt, mt, dt = run(op, x, dy, a.iters, False, {})
        rel = lambda p, q: float((p - q).abs().max() / q.abs().max())  # noqa: E731
        print("%-22s %10.1f %10.1f %10.1f   slab %.1e %.1e | all-taps %.1e %.1e" %
              (label, ta, ts, tt, rel(ms, ma), rel(ds, da), rel(mt, ma), rel(dt, da)))
        tot = [tot[0] + ta, tot[1] + ts, tot[2] + tt]
        if a.sweep:
            row = []
            for wgs in (256, 512, 1024):
                row.append("%d: %.1f" % (wgs, run(op, x, dy, a.iters, False, {"BTX_WGRAD_NO_TAPS3": "1", "BTX_WGRAD_SLAB_WGS": str(wgs)})[0]))
            if k == 3 and s == 1:
                for wgs in (128, 256, 512):
                    row.append("T3 %d: %.1f" % (wgs, run(op, x, dy, a.iters, False, {"BTX_WGRAD_T3_WGS": str(wgs)})[0]))
            print("    sweep (target workgroups: us)  " + "  ".join(row))
        if a.ablate and k == 3 and s == 1:
            row = ["%s: %.1f" % (nm, run(op, x, dy, a.iters, False, {"BTX_WGRAD_T3_ABL": v})[0])
                   for nm, v in (("no slab stores", "1"), ("no MFMA section", "2"), ("neither", "3"))]
            print("    all-taps ablation (us, incl. the slab reduction launch)  " + "  ".join(row))
    print("%-22s %10.1f %10.1f %10.1f" % ("sum", tot[0], tot[1], tot[2]))


if __name__ == "__main__":
    main()
