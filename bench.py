#!/usr/bin/env python3
"""bench.py — MC-samples/sec of the variational-layer forward hot path on MI355X.

Workload (BASELINE.json metric / configs[3], per-GPU shard): dnn_to_bnn(ResNet18) Flipout, 224x224, batch 64,
synthetic input, reference init draws (torch.manual_seed(0)), MC sample s keyed (seed=2024, sample_idx=s).
A "step" = one Monte-Carlo sample: one stochastic forward of the whole converted model (one weight-sampling launch,
21 contraction launches with eval-BN / residual / ReLU folded into their stores, the pooling ops between them) + the
on-device accumulation of the predictive statistics, replayed as ONE hipGraph per sample (--no-graph: eager launches).  N>1: every rank runs its own K samples (weak scaling, sample indices interleaved by rank),
then ONE RCCL all-reduce of the packed statistics inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the contraction launches (dominant kernel: btx::contract_patch_kernel, Flipout): algorithmic FLOPs of every
                launch (2*2*M*N*K per Flipout launch) / its HIP-event duration on the launch stream, vs
                the dense MFMA peak of the contraction dtype.
  cpu_baseline  oracle/bt_ref.py (the reference's ATen op chain) timed on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
PREWARM_STEPS = 12
KL_KNOWN = 55.67487335205078  # reference get_kl_loss(dnn_to_bnn(resnet18)), default init, seed 0


def build_model(typ, device, act_dtype, fuse=True):
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models.resnet import resnet18
    torch.manual_seed(0)
    m = resnet18()
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type=typ,
                          moped_enable=False, moped_delta=0.5))
    m = m.to(device).eval()
    if act_dtype == torch.bfloat16:
        # activations (and the stock BN layers) in bf16; the variational parameters stay f32 (the kernels read
        # mu/rho as f32 and sample in f32)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.to(torch.bfloat16)
    bt.assign_layer_ids(m)
    if fuse:
        # eval-mode BN (+ residual + ReLU) folded into the store of the contraction kernels
        from bayesian_torch_amd.models.fuse import fuse_resnet
        fuse_resnet(m)
    return m


def cpu_baseline(typ, bs, budget_s=25.0):
    """the reference's op chain (oracle/bt_ref.py) on the host cores: 1 warm-up + as many timed MC forwards as fit"""
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.models.resnet import resnet18
    from oracle import bt_ref
    # all cores up to 32: the op chain is dominated by serial RNG fills and small convs, and on the 256-thread GPU-box
    # host an uncapped run measured 10x slower than 8 threads of the build container (0.024 vs 0.22 MC-samples/s)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = resnet18()
    bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type=typ,
                          moped_enable=False, moped_delta=0.5))
    m = bt_ref.convert_for_baseline(m).eval()
    torch.manual_seed(1234)
    x = torch.randn(bs, 3, 224, 224)
    with torch.no_grad():
        t0 = time.time()
        m(x)
        warm = time.time() - t0
        n, t0 = 0, time.time()
        while True:
            m(x)
            n += 1
            el = time.time() - t0
            if el + el / n > budget_s - warm or n >= 5:
                break
    return {"value": n / el, "unit": "MC-samples/s", "cores": cores, "kind": "port",
            "sample": "%d timed MC forwards (+1 warm-up) of ResNet18-%s bs%d 224^2 f32, oracle/bt_ref.py ATen op chain, "
                      "%d of %d host threads" % (n, typ, bs, cores, os.cpu_count() or 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--type", default="Flipout", choices=["Flipout", "Reparameterization"])
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--act", default=None, choices=["bf16", "f32"], help="activation dtype (default = --prec)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-fuse", action="store_true", help="keep BatchNorm/ReLU/residual as separate torch ops")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of every MC sample from Python instead of "
                    "replaying one captured hipGraph per sample")
    ap.add_argument("--lanes", type=int, default=3, help="MC samples evaluated concurrently (one stream each) inside one "
                    "hipGraph replay; independent noise, identical results to one at a time")
    ap.add_argument("--no-presample", action="store_true", help="sample the weights per layer launch instead of once per MC sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--per-step", action="store_true", help="diagnostic: print host time of every timed step to stderr")
    args = ap.parse_args()
    act = args.act or args.prec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import bayesian_torch_amd as bt
    from bayesian_torch_amd import functional as BF
    from bayesian_torch_amd import mc
    bt.manual_seed(2024)
    bt.set_precision(args.prec)
    act_dtype = torch.bfloat16 if act == "bf16" else torch.float32
    model = build_model(args.type, dev, act_dtype, fuse=not args.no_fuse)
    torch.manual_seed(1234)
    x = torch.randn(args.batch, 3, 224, 224).to(dev).to(act_dtype)

    with torch.no_grad():
        kl_t = bt.get_kl_loss(model)
    kl = float(kl_t)
    graphed = None
    if args.no_graph:
        packed = torch.zeros(mc.packed_numel(args.batch, 1000), dtype=torch.float32, device=dev)

        def step(s_global):
            bt.set_sample_index(model, s_global, presample=not args.no_presample)
            logits = model(x)
            mc.accumulate(packed, logits, kl)
    else:
        # one MC sample (weight sampling + 21 fused contractions + pooling + accumulation) = one hipGraph replay;
        # the sample index is a device word the kernels read when they run (BtxRng.sample_idx_dev)
        rest_graphs = {}
        try:
            graphed = mc.GraphedMC(model, x, kl=kl, lanes=max(1, args.lanes))
            # ragged last groups (steps / warm-up not a multiple of the lane count): one smaller graph per remainder size
            rest_graphs = {r: mc.GraphedMC(model, x, kl=kl, lanes=r)
                           for r in sorted({args.steps % graphed.lanes, args.warmup % graphed.lanes,
                                            PREWARM_STEPS % graphed.lanes, 1 if args.per_step else 0} - {0})}
        except Exception as e:  # a runtime that cannot capture: measure the eager path rather than nothing
            print("bench: hipGraph capture failed (%s: %s) - falling back to eager launches" % (type(e).__name__, e),
                  file=sys.stderr)
            graphed = None
            for m_ in model.modules():
                if hasattr(m_, "_btx_sample_dev"):
                    m_._btx_sample_dev = None
            torch.cuda.synchronize(dev)
    if graphed is not None:
        packed = graphed.packed
        lanes = graphed.lanes

        def run_steps(indices):
            """MC samples `indices`, `lanes` at a time (one hipGraph replay per group), the ragged rest one by one"""
            full = len(indices) // lanes * lanes
            for i in range(0, full, lanes):
                if lanes == 1:
                    graphed.run(indices[i])
                else:
                    graphed.run_many(indices[i:i + lanes])
            rest = indices[full:]
            if rest:
                gr = rest_graphs[len(rest)]
                gr.run(rest[0]) if gr.lanes == 1 else gr.run_many(rest)
    elif not args.no_graph:
        packed = torch.zeros(mc.packed_numel(args.batch, 1000), dtype=torch.float32, device=dev)

        def step(s_global):
            bt.set_sample_index(model, s_global, presample=not args.no_presample)
            logits = model(x)
            mc.accumulate(packed, logits, kl)

    if graphed is None:
        def run_steps(indices):
            for i in indices:
                step(i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        # Initialisation, untimed and in addition to --warmup: the HIP runtime grows an internal pool once after
        # ~500 kernel launches (a single 40-60 ms host stall, measured with --per-step); run past it.
        run_steps([20_000_000 + w * world + rank for w in range(PREWARM_STEPS)])
        run_steps([10_000_000 + w * world + rank for w in range(args.warmup)])
        if world > 1:
            dist.all_reduce(packed)  # warm the communicator too
        packed.zero_()
        if graphed is not None:
            for gr in rest_graphs.values():
                gr.packed.zero_()
        if not args.no_launch_timing and graphed is None:
            BF.enable_launch_timing(True)
        barrier()
        t0 = time.perf_counter()
        if args.per_step:
            for k in range(args.steps):
                ts = time.perf_counter()
                run_steps([k * world + rank])
                print("step %d: host %.3f ms (launch only, no sync)" % (k, 1e3 * (time.perf_counter() - ts)), file=sys.stderr)
        else:
            run_steps([k * world + rank for k in range(args.steps)])
        if graphed is not None:
            for gr in rest_graphs.values():
                packed.add_(gr.packed)  # the ragged rest of the last group
        if world > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        barrier()
        elapsed = time.perf_counter() - t0
    stats = packed.clone()
    if graphed is not None:
        graphed.close()
        for gr in rest_graphs.values():
            gr.close()
        if not args.no_launch_timing:
            # Kernel durations for the roofline: HIP events cannot bracket the nodes of a replayed graph, so the same
            # launches are issued once more eagerly, with an event pair around each, right after the timed region.
            scratch = torch.zeros_like(packed)
            with torch.no_grad():
                BF.enable_launch_timing(True)
                for k in range(min(args.steps, 10)):
                    # a ~2 ms spin kernel in front of every step lets the host run ahead: without it the GPU would wait for
                    # the next launch INSIDE an event pair (eager issue of a step takes longer than its kernels)
                    torch.cuda._sleep(5_000_000)
                    bt.set_sample_index(model, k * world + rank, presample=not args.no_presample)
                    mc.accumulate(scratch, model(x), kl)
                torch.cuda.synchronize(dev)
    log = BF.launch_log() or []
    BF.enable_launch_timing(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)

    # roofline of the dominant kernel from the HIP events of the timed region
    roofline = None
    if log:
        log_steps = args.steps if graphed is None else min(args.steps, 10)
        total_flops = sum(f for _, f, _, _ in log)
        total_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in log)
        achieved = total_flops / (total_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.prec]
        per = {}
        for tag, f, e0, e1 in log:
            d = per.setdefault(tag, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += f
            d[2] += e0.elapsed_time(e1)
        layers = sorted(({"launch": t, "n": v[0], "avg_us": 1e3 * v[2] / v[0], "tflops": v[1] / (v[2] * 1e-3) / 1e12}
                         for t, v in per.items()), key=lambda r: -r["avg_us"] * r["n"])
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic,
                    "kernel": "btx_contract_fwd_ex launches <%s,%s,%s>: contract_patch_kernel (13 of 21 launches, the "
                              "dominant kernel; `traffic` is its layer1 launch), contract_dma_kernel, contract_stem_kernel, "
                              "incl. their split-K reduce" % (args.prec, act, args.type),
                    "launches": len(log), "avg_launch_us": 1e3 * total_ms / len(log),
                    "kernel_time_frac_of_step": (total_ms * 1e-3 / log_steps) / (elapsed / args.steps),
                    "algorithmic_gflop_per_step": total_flops / log_steps / 1e9,
                    "measured": "HIP events around every launch, " + (
                        "timed region" if graphed is None else "eager re-issue of the same launches after the timed "
                        "region (the timed region replays one hipGraph per MC sample)"),
                    "top_launches": layers[:6]}

    if rank == 0:
        u = mc.unpack(stats, args.batch, 1000)
        assert abs(float(u["samples"]) - args.steps * world) < 0.5, "work was skipped inside the timed region"
        assert torch.isfinite(u["mean_prob"]).all()
        out = {
            "metric": "MC-samples/sec (Bayesian-ResNet18, 224^2, bs=64)", "value": args.steps * world / elapsed,
            "unit": "MC-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "dnn_to_bnn(ResNet18) %s, 224x224, batch %d, %d MC samples per GPU, default init "
                                   "(seed 0), activations %s, eval-BN/ReLU/residual %s, %s" % (
                                       args.type, args.batch, args.steps, act,
                                       "as torch ops" if args.no_fuse else "folded into the kernel epilogue",
                                       "eager launches" if graphed is None else
                                       "hipGraph replay, %d MC samples in flight (one stream each)" % graphed.lanes),
                       "global_batch": args.batch * world, "parallelism": "mc-sample-shard x%d" % world},
            "image_samples_per_s": args.batch * args.steps * world / elapsed,
            "kl": kl, "kl_rel_err": abs(kl - KL_KNOWN) / KL_KNOWN,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.type, args.batch)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
