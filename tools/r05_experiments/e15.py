import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
r = bench.run_mlp_config(torch.device("cuda:0"))
print({k: v for k, v in r.items() if k != "workload"})
