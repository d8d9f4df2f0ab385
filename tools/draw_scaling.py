"""Draw launch time against the number of MC samples: the slope is the per-sample VALU work, the intercept is launch +
posterior loads + softplus + ramp / tail.   usage: python tools/draw_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
for S in (1, 2, 4, 8, 16, 32):
    layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=int(os.environ.get("GEN", "1"))), DrawKey(1, 2 * i + 2, 0, S, 0, gen=int(os.environ.get("GEN", "1")))) for i, (mw, rw, mb, rb) in enumerate(post)]
    us = bench._graph_time(lambda: ops.draw_layers(layers, S), dev)
    print("S = %2d: %.2f us" % (S, us))
