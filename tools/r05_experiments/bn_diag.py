import sys; sys.path.insert(0,'.')
import torch
from bayesian_torch_amd.models.fuse import hip_batchnorm
dev=torch.device('cuda:0'); torch.manual_seed(5)
C=64
x=(torch.randn(8,C,56,56,device=dev)*1.7+0.3); x[:, :4]+=300.0
x=x.contiguous(memory_format=torch.channels_last)
bn=torch.nn.BatchNorm2d(C,momentum=1.0).to(dev); ref=torch.nn.BatchNorm2d(C,momentum=1.0).to(dev)
hip_batchnorm(bn); bn.train(); ref.train()
y=bn(x); yr=ref(x)
xd=x.double().permute(1,0,2,3).reshape(C,-1)
m64=xd.mean(1); v64=xd.var(1,unbiased=True)
print("mean  hip err %.3e  torch err %.3e"%(float((bn.running_mean.double()-m64).abs().max()), float((ref.running_mean.double()-m64).abs().max())))
print("var   hip rel err %.3e  torch rel err %.3e"%(float(((bn.running_var.double()-v64)/v64).abs().max()), float(((ref.running_var.double()-v64)/v64).abs().max())))
y64=(xd - m64[:,None])/torch.sqrt(xd.var(1,unbiased=False)[:,None]+1e-5)
y64=y64.reshape(C,8,56,56).permute(1,0,2,3)
print("y     hip rel-L2 %.3e  torch rel-L2 %.3e"%(float((y.double()-y64).norm()/y64.norm()), float((yr.double()-y64).norm()/y64.norm())))
