"""The real draw kernel on ONE 2000 x 1200 tensor (the shape tools/ubench_draw.hip's (D) times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
S = 8
for rows, cols in ((2000, 1200), (2000, 1216), (1200, 784), (1200, 1200)):
    mu = torch.randn(rows, cols, device=dev) * 0.05; rho = torch.full((rows, cols), -2.0, device=dev)
    L = [(mu, rho, None, None, DrawKey(1, 1, 0, S, 0), None)]
    us = bench._graph_time(lambda: ops.draw_layers(L, S), dev)
    # the same launch without the allocation inside the graph: call the C entry on preallocated outputs
    print("%d x %d: %.2f us  (%.2f us per M scalars)" % (rows, cols, us, us / (rows * cols / 1e6)))
