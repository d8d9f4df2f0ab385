"""Layer-2 dense GEMM launches alone (for rocprofv3 --pmc / --kernel-trace): python tools/run_dense_l2.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
dev = torch.device("cuda:0"); S, B = 8, 512
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mw, rw, mb, rb = [t.to(dev) for t in bench.posteriors(0)[1]]
pre = ops.draw_layers([(mw, rw, mb, rb, DrawKey(1, 1, 0, S, 0, gen=1), DrawKey(1, 2, 0, S, 0, gen=1))], S)[0]
h = torch.randn(S, B, 1200, device=dev).relu_().bfloat16()
for _ in range(n):
    ops._dense_raw(h, B * 1200, B, pre, 1200, True, torch.bfloat16)
    ops.draw_layers([(mw, rw, mb, rb, DrawKey(1, 1, 0, S, 0, gen=1), DrawKey(1, 2, 0, S, 0, gen=1))], S)
torch.cuda.synchronize()
