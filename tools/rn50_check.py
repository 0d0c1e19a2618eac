import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_torch_amd as bt
from bayesian_torch_amd import mc
from bayesian_torch_amd.models.resnet import resnet50
from bayesian_torch_amd.models.fuse import fuse_resnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = resnet50()
bt.dnn_to_bnn(m, dict(prior_mu=0.0, prior_sigma=1.0, posterior_mu_init=0.0, posterior_rho_init=-3.0, type="Flipout", moped_enable=False, moped_delta=0.5))
m = m.to(dev).eval()
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d): mod.to(torch.bfloat16)
bt.assign_layer_ids(m); bt.manual_seed(1); bt.set_precision("bf16")
x = torch.randn(64, 3, 224, 224, device=dev).to(torch.bfloat16)
with torch.no_grad():
    bt.set_sample_index(m, 0); ref = m(x).float()
    n = fuse_resnet(m)
    bt.set_sample_index(m, 0); y = m(x).float()
    print("fused blocks", n, "rel err fused vs unfused", float((y - ref).norm() / ref.norm()), "finite", bool(torch.isfinite(y).all()))
    g = mc.GraphedMC(m, x, kl=0.0)
    for s in range(5): g.run(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(20): g.run(100 + s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("ResNet50-Flipout bs64 224 bf16: %.3f ms / MC sample = %.1f MC-samples/s" % (dt * 1e3, 1 / dt))
