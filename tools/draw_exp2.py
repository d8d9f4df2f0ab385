"""Which part of the BASELINE draw launch costs what: weights only / biases only / one tensor at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
S = 8
def K(i): return DrawKey(1, i, 0, S, 0, gen=1)
full = [(mw, rw, mb, rb, K(2 * i + 1), K(2 * i + 2)) for i, (mw, rw, mb, rb) in enumerate(post)]
nob = [(mw, rw, None, None, K(2 * i + 1), None) for i, (mw, rw, mb, rb) in enumerate(post)]
print("all 6 tensors      : %.2f us" % bench._graph_time(lambda: ops.draw_layers(full, S), dev))
print("3 weights, no bias : %.2f us" % bench._graph_time(lambda: ops.draw_layers(nob, S), dev))
for i in range(3):
    print("layer %d weight only: %.2f us" % (i, bench._graph_time(lambda: ops.draw_layers(nob[i:i + 1], S), dev)))
    print("layer %d w + b      : %.2f us" % (i, bench._graph_time(lambda: ops.draw_layers(full[i:i + 1], S), dev)))
