"""Wide dense launches at S = 1 (the configs[4] layer: 4096 x 4096, batch 4096) with and without the XCD-contiguous, grouped tile
order (BNN_DENSE_XCD=0 = plain order): bf16 operands and the fp32 parity mode's three-plane operands.
usage: [BNN_DENSE_XCD=0] python tools/wide_xcd_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
dev = torch.device("cuda:0")
tag = "plain order" if os.environ.get("BNN_DENSE_XCD") == "0" else "XCD-contiguous groups"
for (S, M, N, K) in ((1, 4096, 4096, 4096), (3, 2048, 2048, 2048), (1, 4096, 1200, 4096)):
    mw = torch.randn(N, K, device=dev) * 0.02; rw = torch.full((N, K), -2.0, device=dev)
    mb = torch.zeros(N, device=dev); rb = torch.full((N,), -2.0, device=dev)
    lay = [(mw, rw, mb, rb, DrawKey(1, 1, 0, S, 0), DrawKey(1, 2, 0, S, 0))]
    pre = ops.draw_layers(lay, S)[0]
    xb = torch.randn(M, K, device=dev).bfloat16()
    us = bench._graph_time(lambda: ops._dense_raw(xb, 0, M, pre, K, True, torch.bfloat16), dev, reps=5, iters=5)
    fl = 2.0 * S * M * N * K
    print("%-22s bf16 %d x (%d x %d x %d): %8.1f us = %6.1f TFLOP/s" % (tag, S, M, N, K, us, fl / us / 1e6))
    pre3 = ops.draw_layers(lay, S, x3=True)[0]
    xp = ops.split_x3(torch.randn(M, K, device=dev))
    us = bench._graph_time(lambda: ops._dense_raw_x3(xp, True, M, pre3, K, True, False), dev, reps=5, iters=5)
    print("%-22s x3   %d x (%d x %d x %d): %8.1f us = %6.1f fp32-equivalent TFLOP/s" % (tag, S, M, N, K, us, fl / us / 1e6))
    del pre, pre3
