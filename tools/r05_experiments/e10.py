"""E10: two hipGraphs of L/2 lanes replayed concurrently on two streams vs one graph of L lanes (fills the launch tails?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import bayesian_torch_amd as bt
from bayesian_torch_amd import mc

dev = torch.device("cuda:0")
bt.manual_seed(2024); bt.set_precision("bf16")
model = bench.build_model("Flipout", dev, torch.bfloat16)
torch.manual_seed(1234)
x = torch.randn(64, 3, 224, 224).to(dev).to(torch.bfloat16)
kl = float(bt.get_kl_loss(model))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


g1 = mc.GraphedMC(model, x, kl=kl, lanes=L, static_input=True)
t1 = timeit(lambda: g1.run_many(list(range(L))))
print("one graph of %d lanes: %.3f ms per replay -> %.1f MC-samples/s" % (L, t1 * 1e3, L / t1))
ref = g1.packed.clone(); g1.close()
streams = [torch.cuda.Stream(dev) for _ in range(G)]
gs = []
for s in streams:
    gs.append(mc.GraphedMC(model, x, kl=kl, lanes=L // G, static_input=True, capture_stream=s))


def both():
    cur = torch.cuda.current_stream(dev)
    for k, (g, s) in enumerate(zip(gs, streams)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            g.run_many(list(range(k * (L // G), (k + 1) * (L // G))))
    for s in streams:
        cur.wait_stream(s)


t2 = timeit(both)
print("%d graphs of %d lanes on %d streams: %.3f ms per replay -> %.1f MC-samples/s  (%.1f %%)" % (G, L // G, G, t2 * 1e3, L / t2, 100 * (t1 / t2 - 1)))
