"""configs[2] / configs[3] conv launches alone (rocprofv3): python tools/run_conv.py lenet|cifar [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd.nn import NormalConv2d
from bayesianneuralnetworks_amd import _mc
dev = torch.device("cuda:0"); S = 8
B, C, O, HW, k, s, p = {"lenet": (1024, 64, 64, 6, 3, 2, 1), "cifar": (256, 128, 128, 4, 3, 1, 1)}[sys.argv[1]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
layer = NormalConv2d(C, O, k, stride=s, padding=p).to(dev)
x = torch.randn(S * B, C, HW, HW, device=dev)
bnn.set_compute("bf16")
with torch.no_grad(), _mc.McContext(S, B, 0):
    for _ in range(n):
        layer(x)
torch.cuda.synchronize()
