#!/usr/bin/env python3
"""GPU diagnostics: (1) parity table of the fused kernels vs the CPU oracle over many shapes/modes (never stops at the
first failure), (2) per-layer kernel time / TFLOP/s on the ResNet18 bs64 shapes with HIP events.

usage: python tools/gpu_diag.py [parity] [perf] [--prec bf16,f32] [--iters 10]
"""
import argparse
import os
import sys
import traceback
import warnings

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parity():
    from helpers import load_golden, case_geometry, oracle_forward, rel_l2
    from test_gpu_contract import FUSED_CASES, _run_fused, _layer_from_meta, _noise_from
    dev = torch.device("cuda:0")
    g = load_golden()
    print("== explicit-noise (reference goldens, GEN kernels) ==")
    for prec in ("f32", "bf16"):
        for name, (meta, d) in g["cases"].items():
            try:
                layer = _layer_from_meta(meta, dev)
                layer.precision = prec
                x = torch.from_numpy(d["x"]).to(dev)
                with torch.no_grad():
                    out = layer._forward_hip(x, noise=_noise_from(d, dev), sample_idx=0).float().cpu().numpy()
                torch.cuda.synchronize()
                print("  %-5s %-28s rel-L2 vs reference %.3e  %s" % (prec, name, rel_l2(out, d["out"]),
                                                                     "" if np.isfinite(out).all() else "NON-FINITE"))
            except Exception as e:  # noqa
                print("  %-5s %-28s EXC %s" % (prec, name, repr(e)[:200]))
    print("== fused-noise (in-kernel Philox / hash) vs oracle fed with materialised noise ==")
    for prec, act in (("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16")):
        for cls, kw, xshape in FUSED_CASES:
            try:
                layer, x, out, geo, a = _run_fused(cls, kw, xshape, prec, act, dev)
                torch.cuda.synchronize()
                o = out.float().cpu().numpy()
                ref = oracle_forward(geo, a["x"], a["mu_w"], a["rho_w"], a["mu_b"], a["rho_b"], a["eps_w"], a["eps_b"],
                                     a["sign_in"], a["sign_out"], bf16=(prec == "bf16"))
                # split the error: mean part only (set signs/eps irrelevant) is not separable here; print max-abs too
                print("  %-4s/%-4s %-34s x%-18s rel-L2 %.3e  maxabs %.3e / %.3e" % (
                    prec, act, cls, str(xshape), rel_l2(o, ref), np.abs(o - ref).max(), np.abs(ref).max()))
            except Exception as e:  # noqa
                print("  %-4s/%-4s %-34s x%-18s EXC %s" % (prec, act, cls, str(xshape), repr(e)[:300]))
                traceback.print_exc(limit=2)


RN18 = [  # (Cin, Cout, H, stride, k, count in ResNet18)
    (3, 64, 224, 2, 7, 1), (64, 64, 56, 1, 3, 4), (64, 128, 56, 2, 3, 1), (64, 128, 56, 2, 1, 1), (128, 128, 28, 1, 3, 3),
    (128, 256, 28, 2, 3, 1), (128, 256, 28, 2, 1, 1), (256, 256, 14, 1, 3, 3), (256, 512, 14, 2, 3, 1),
    (256, 512, 14, 2, 1, 1), (512, 512, 7, 1, 3, 3)]


def perf(precs, iters, bs=64):
    from bayesian_torch_amd import layers as L
    dev = torch.device("cuda:0")
    peak = {"bf16": 2500.0, "f32": 157.3}
    for typ in ("Flipout", "Reparameterization"):
        for prec in precs:
            act = torch.bfloat16 if prec == "bf16" else torch.float32
            tot_ms, tot_fl = 0.0, 0.0
            print("== %s prec=%s act=%s bs=%d ==" % (typ, prec, act, bs))
            for cin, cout, hw, stride, k, cnt in RN18:
                torch.manual_seed(0)
                cls = getattr(L, "Conv2d" + typ)
                layer = cls(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
                layer.precision = prec
                x = torch.randn(bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
                with torch.no_grad():
                    for _ in range(2):
                        y = layer._forward_hip(x, sample_idx=0)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(iters):
                        y = layer._forward_hip(x, sample_idx=i)
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                ho = y.shape[2]
                fl = 2.0 * bs * ho * ho * cout * cin * k * k * (2 if typ == "Flipout" else 1)
                tf = fl / (ms * 1e-3) / 1e12
                tot_ms += ms * cnt
                tot_fl += fl * cnt
                print("  cin%4d cout%4d hw%4d s%d k%d  x%d : %8.1f us  %7.1f TFLOP/s  (%4.1f%% of %s peak)" % (
                    cin, cout, hw, stride, k, cnt, ms * 1e3, tf, 100 * tf / peak[prec], prec))
            print("  ALL 20 convs: %.3f ms / MC sample, %.1f TFLOP/s = %.1f%% of peak" % (
                tot_ms, tot_fl / (tot_ms * 1e-3) / 1e12, 100 * tot_fl / (tot_ms * 1e-3) / 1e12 / peak[prec]))


def one(prec, iters, cin=64, cout=64, hw=56, stride=1, k=3, typ="Flipout", bs=64):
    """one layer shape, `iters` launches — the target of rocprofv3 runs"""
    from bayesian_torch_amd import layers as L
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(0)
    layer = getattr(L, "Conv2d" + typ)(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
    layer.precision = prec
    x = torch.randn(bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        layer._forward_hip(x, sample_idx=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            layer._forward_hip(x, sample_idx=i)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def lanes_run(prec, iters, cin=64, cout=64, hw=56, stride=1, k=3, typ="Flipout", bs=64, lanes=8):
    """one layer shape the way the bench's MC loop runs it: `lanes` MC samples per launch (their own activations), the
    weights sampled for all lanes by one pre-pass launch with the mean tiles cached after the first call (skip_mu) —
    `iters` x (sampling launch + contraction launch): the target of the rocprofv3 --pmc traffic runs of bench.py"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd import layers as L, rng as R
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(0)
    bt.manual_seed(1)
    layer = getattr(L, "Conv2d" + typ)(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
    layer.precision = prec
    bt.assign_layer_ids(layer)
    x = torch.randn(lanes * bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
    cache = {}
    with torch.no_grad():
        sd = bt.set_sample_lanes(layer, list(range(lanes)), batch=bs)
        layer._forward_hip(x)  # records the input shape the pre-pass needs
        R._presample(layer, 0, cache=cache)
        layer._forward_hip(x)
        torch.cuda.synchronize()
        for i in range(iters):
            bt.set_sample_lanes(layer, [100 + i * lanes + l for l in range(lanes)], batch=bs, sample_dev=sd)
            R._presample(layer, 100 + i * lanes, cache=cache, skip_mu=True)
            layer._forward_hip(x)
    torch.cuda.synchronize()


def gtime(prec, cin=64, cout=64, hw=56, stride=1, k=3, typ="Flipout", bs=64, reps=20):
    """GPU time of one layer call (all its kernels) with the host out of the picture: `reps` calls captured in a
    hipGraph, replayed and timed with events"""
    from bayesian_torch_amd import layers as L
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(0)
    layer = getattr(L, "Conv2d" + typ)(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
    layer.precision = prec
    x = torch.randn(bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (3 * reps) * 1e3
    ho = y.shape[2]
    fl = 2.0 * bs * ho * ho * cout * cin * k * k * (2 if typ == "Flipout" else 1)
    return us, fl / (us * 1e-6) / 1e12


def trace(prec, cin=64, cout=64, hw=56, stride=1, k=3, typ="Flipout", bs=64, warm=3):
    """per-wave phase timings of the patch kernel (needs a libbtx built with -DBTX_PT_TRACE, see BTX_LIB)"""
    import os
    import numpy as np
    from bayesian_torch_amd import layers as L
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(0)
    layer = getattr(L, "Conv2d" + typ)(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(dev)
    layer.precision = prec
    x = torch.randn(bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
    buf = torch.zeros(1 << 22, dtype=torch.int32, device=dev)
    with torch.no_grad():
        for i in range(warm):  # many launches: the traced one runs at the clock the GPU settles to under this load
            layer._forward_hip(x, sample_idx=i)
        os.environ["BTX_TRACE_PTR"] = hex(buf.data_ptr())
        layer._forward_hip(x, sample_idx=7)
        torch.cuda.synchronize()
        del os.environ["BTX_TRACE_PTR"]
    t = buf.cpu().numpy().view(np.uint32).reshape(-1, 8)
    t = t[t[:, 5] != 0]
    if os.environ.get("BTX_TRACE_DUMP"):
        np.save(os.environ["BTX_TRACE_DUMP"], t)
    print("waves traced: %d" % len(t))
    if not len(t):
        return
    names = ["prologue", "A->B issue+mma", "B->C vmcnt|100MHz ticks", "C->D barrier", "epilogue", "total"]
    if os.environ.get("BTX_PERSIST"):  # contract_taps3_kernel: sums over the tiles a workgroup walks (column 7: tiles)
        names = ["prologue", "K loops (sum)", "100MHz ticks", "-", "store sides (sum)", "total"]
        if int(os.environ.get("BTX_TAPS_TUNE", "0")) & 128:
            names = ["ss: in-flight landed", "ss: addresses+constants+barrier", "ss: combine", "ss: pair+residual+stores", "ss: next tile+barrier", "tiles"]
    elif int(os.environ.get("BTX_TAPS_TUNE", "0")) & 128:  # prologue sub-stamps, cumulative from the kernel's first instruction
        names = ["patch DMAs issued", "sign keys derived", "role index math", "sign words written", "patch landed+barrier", "first fragments (K loop starts)"]
    if t[:, 2].astype(np.float64).mean() > 0 and t[:, 3].astype(np.float64).mean() == 0:
        print("  shader clock while the blocks ran: %.3f GHz (s_memtime / s_memrealtime)" % (
            t[:, 5].astype(np.float64).mean() / t[:, 2].astype(np.float64).mean() * 0.1))
    for i, n in enumerate(names):
        c = t[:, i].astype(np.float64)
        print("  %-16s mean %9.0f  min %9.0f  max %9.0f  (clock ticks)" % (n, c.mean(), c.min(), c.max()))
    t0 = t[:, 6].astype(np.int64)
    t0 = (t0 - t0.min()) & 0xffffffff
    end = t0 + t[:, 5]
    print("  kernel span %d ticks; start-time histogram (8 bins): %s" % (end.max(), np.histogram(t0, bins=8)[0].tolist()))
    print("  column 7 (patch: HW_ID, dma: geometry part of the prologue) mean %.0f" % t[:, 7].astype(np.float64).mean())
    for w in range(0, min(len(t), 16)):
        print("   wave %3d: %s start %d" % (w, t[w, :6].tolist(), int(t0[w])))


def g8trace(prec, cin=256, cout=1024, hw=14, stride=1, k=1, typ="Flipout", bs=512, warm=30, full=True):
    """per-wave phase timings of contract_gemm8_kernel (-DBTX_PT_TRACE -DBTX_EP_TRACE build): prologue / K loop / store stage 1 /
    store stage 2, and per CU the gap between the end of a workgroup and the start of the next one.  full: BN affine + residual
    + ReLU in the store side (a ResNet50 expand convolution); else a bare layer."""
    import os
    import numpy as np
    from bayesian_torch_amd import layers as L
    dev = torch.device("cuda:0")
    act = torch.bfloat16 if prec == "bf16" else torch.float32
    torch.manual_seed(0)
    layer = getattr(L, "Conv2d" + typ)(cin, cout, k, stride=stride, padding=0, bias=False).to(dev)
    layer.precision = prec
    x = torch.randn(bs, cin, hw, hw, device=dev).to(act).contiguous(memory_format=torch.channels_last)
    ho = (hw - 1) // stride + 1
    ep = None
    if full:
        ep = dict(scale=torch.rand(cout, device=dev) + 0.5, shift=torch.randn(cout, device=dev), relu=True,
                  residual=torch.randn(bs, cout, ho, ho, device=dev).to(act).contiguous(memory_format=torch.channels_last))
    buf = torch.zeros(1 << 22, dtype=torch.int32, device=dev)
    with torch.no_grad():
        for i in range(warm):
            layer._forward_hip(x, sample_idx=i, epilogue=ep)
        os.environ["BTX_TRACE_PTR"] = hex(buf.data_ptr())
        layer._forward_hip(x, sample_idx=7, epilogue=ep)
        torch.cuda.synchronize()
        del os.environ["BTX_TRACE_PTR"]
    t = buf.cpu().numpy().view(np.uint32).reshape(-1, 8, 8)
    t = t[t[:, 0, 5] != 0]
    print("g8trace %d->%d %dx%d s%d bs %d %s: workgroups traced: %d" % (cin, cout, hw, hw, stride, bs, "full store side" if full else "bare", len(t)))
    if not len(t):
        return
    f = t.astype(np.float64)
    ghz = f[:, :, 5].mean() / f[:, :, 2].mean() * 0.1
    print("  shader clock %.3f GHz;  K stages %d" % (ghz, cin // 32))
    for g in (0, 1):
        w = f[:, 4 * g:4 * g + 4, :]
        print("  waves %d-%d: prologue %6.0f | K loop %6.0f (%.0f / stage) | store stage 1 %6.0f | stage 2 %6.0f | drain %5.0f | total %6.0f cycles" % (
            4 * g, 4 * g + 3, w[:, :, 0].mean(), w[:, :, 1].mean(), w[:, :, 1].mean() / (cin // 32), w[:, :, 3].mean(), w[:, :, 4].mean(),
            (w[:, :, 5] - w[:, :, 0] - w[:, :, 1] - w[:, :, 3] - w[:, :, 4]).mean(), w[:, :, 5].mean()))
    start = t[:, :, 6].astype(np.int64)
    base = start.min()
    start = (start - base) & 0xffffffff
    end = start + t[:, :, 5].astype(np.int64)
    bs_, be_ = start.min(axis=1), end.max(axis=1)
    key = (t[:, 0, 7] >> 8) & 0xfff  # XCC | SE | SH | CU
    gaps, per = [], []
    for kk in np.unique(key):
        idx = np.where(key == kk)[0]
        o = idx[np.argsort(bs_[idx])]
        per.append(len(o))
        gaps.extend((bs_[o[1:]] - be_[o[:-1]]).tolist())
    gaps = np.array(gaps, dtype=np.float64)
    print("  CUs seen %d, workgroups per CU %.1f; workgroup time (first wave start -> last wave end) mean %.0f; gap to the next workgroup of the CU: mean %.0f  median %.0f  min %.0f  max %.0f cycles" % (
        len(per), np.mean(per), (be_ - bs_).mean(), gaps.mean() if len(gaps) else 0, np.median(gaps) if len(gaps) else 0,
        gaps.min() if len(gaps) else 0, gaps.max() if len(gaps) else 0))
    print("  kernel span %.0f cycles = %.1f us" % (be_.max(), be_.max() / ghz / 1e3))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["parity", "perf"])
    ap.add_argument("--prec", default="bf16,f32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shape", default="64,64,56,1,3")
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=8)
    ap.add_argument("--throughput-plan", action="store_true")
    ap.add_argument("--bare", action="store_true", help="g8trace: no BN affine / residual / ReLU in the store side")
    ap.add_argument("--typ", default="Flipout", help="one / timeone / trace: Flipout | Reparameterization")
    a = ap.parse_args()
    if a.throughput_plan:
        from bayesian_torch_amd import functional as _BF
        _BF._CONCURRENT = True
    if "lanes" in a.what:
        c = [int(v) for v in a.shape.split(",")]
        lanes_run(a.prec.split(",")[0], a.iters, *c, bs=a.bs, lanes=a.lanes)
    if "one" in a.what or "timeone" in a.what:
        c = [int(v) for v in a.shape.split(",")]
        us = one(a.prec.split(",")[0], a.iters, *c, typ=a.typ, bs=a.bs)
        if "timeone" in a.what:
            print("shape %s: %.1f us / launch" % (a.shape, us))
    if "gtime" in a.what:
        c = [int(v) for v in a.shape.split(",")]
        us, tf = gtime(a.prec.split(",")[0], *c, bs=a.bs)
        print("shape %s bs %d: %.1f us / call  %.1f TFLOP/s" % (a.shape, a.bs, us, tf))
    if "g8trace" in a.what:
        c = [int(v) for v in a.shape.split(",")]
        g8trace(a.prec.split(",")[0], *c, bs=a.bs, warm=a.warm, full=not a.bare)
    if "trace" in a.what:
        c = [int(v) for v in a.shape.split(",")]
        trace(a.prec.split(",")[0], *c, typ=a.typ, bs=a.bs, warm=a.warm)
    if "parity" in a.what:
        parity()
    if "perf" in a.what:
        perf(a.prec.split(","), a.iters)
