"""the training step of bench.py (extra.train_step) — target of `rocprofv3 --kernel-trace`.
   python tools/train_profile.py            bench.run_train_step (eager steps, captured replays, the f32 parity step)
   python tools/train_profile.py --graph N  N replays of autograd.GraphedTrainStep only (per-step kernel composition)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if "--graph" in sys.argv:
    n = int(sys.argv[sys.argv.index("--graph") + 1])
    import bayesian_torch_amd as bt
    from bayesian_torch_amd.autograd import GraphedTrainStep
    from bayesian_torch_amd.models.fuse import hip_batchnorm
    dev = torch.device("cuda:0")
    bt.manual_seed(2024)
    bt.set_precision("bf16")
    model = bench.build_model("Flipout", dev, torch.bfloat16, fuse=False).train()
    hip_batchnorm(model)
    torch.manual_seed(1234)
    x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
    y = torch.randint(0, 1000, (64,), device=dev)
    gs = GraphedTrainStep(model, x, y)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(n):
        gs.run(i)
    torch.cuda.synchronize()
    print("replays %d: %.3f ms per step" % (n, 1e3 * (time.perf_counter() - t0) / n))
else:
    print(bench.run_train_step(torch.device("cuda:0"), steps=4))
