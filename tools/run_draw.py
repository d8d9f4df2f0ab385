"""A few eager draw launches of the BASELINE net (for rocprofv3 passes).   usage: python tools/run_draw.py [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0), DrawKey(1, 2 * i + 2, 0, S, 0)) for i, (mw, rw, mb, rb) in enumerate(post)]
for _ in range(12):
    pre = ops.draw_layers(layers, S)
torch.cuda.synchronize()
