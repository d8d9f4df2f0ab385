"""fp32-mode layer 2 (512 x 1200 x 1200, 8 samples) on three-plane operands alone: python tools/run_x3.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
dev = torch.device("cuda:0")
S, B = 8, 512
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
for li, K in ((0, 784), (1, 1200)):
    mw, rw, mb, rb = post[li]
    pre = ops.draw_layers([(mw, rw, mb, rb, DrawKey(1, 1, 0, S, 0), DrawKey(1, 2, 0, S, 0))], S, x3=True)[0]
    h = torch.randn(S * B, K, device=dev).relu_()
    xp = ops.split_x3(h).view(3, S, B, -1)
    us = bench._graph_time(lambda: ops._dense_raw_x3(xp, False, B, pre, K, True, li == 0), dev)
    print("fp32-mode layer %d (512 x %d x 1200 x 8): %.2f us = %.1f fp32-equivalent TFLOP/s" % (li + 1, K, us, 2.0 * S * B * K * 1200 / us / 1e6))
