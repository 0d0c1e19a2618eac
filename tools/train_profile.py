"""the training step of bench.py (extra.train_step) a few times — target of `rocprofv3 --kernel-trace --stats`"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(bench.run_train_step(torch.device("cuda:0"), steps=4))
