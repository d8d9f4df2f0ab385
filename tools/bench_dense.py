"""Timing of the pieces of the draw-once path (csrc/bnn_dense.hip) at the BASELINE shapes and at 4096^3.
usage: python tools/bench_dense.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench

lib = _lib.load(); dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S, B = 8, 512
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0), DrawKey(1, 2 * i + 2, 0, S, 0)) for i, (mw, rw, mb, rb) in enumerate(post)]


def t(fn, n=iters):
    """us per call, launches replayed from a HIP graph of 10 calls (eager Python calls are host-bound at these sizes)"""
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    reps = max(1, n // 10)
    return bench._time_launches(g.replay, dev, reps, warm=2) * 1e3 / 10


us = t(lambda: ops.draw_layers(layers, S))
nsc = sum(p[0].numel() + p[2].numel() for p in post)
print("draw_multi: 3 layers x 8 samples (%.2f M scalars, %.1f M draws): %.2f us  -> %.0f GB/s algorithmic (8 B read + 2 B x 8 written per scalar)"
      % (nsc / 1e6, nsc * S / 1e6, us, (8 * nsc + 2 * S * nsc) / us / 1e3))
us2 = t(lambda: ops.draw_layers(layers[1:2], S))
print("draw_multi: layer 2 alone x 8 samples: %.2f us" % us2)
pre = ops.draw_layers(layers, S)
x = torch.randn(B, 784, device=dev).bfloat16()
h = torch.randn(S, B, 1200, device=dev).relu_().bfloat16()
for name, xx, stride, p, K, relu, odt in (("layer 1 (512x784x1200 x8, shared x)", x, 0, pre[0], 784, True, torch.bfloat16),
                                          ("layer 2 (512x1200x1200 x8)", h, B * 1200, pre[1], 1200, True, torch.bfloat16),
                                          ("head    (512x1200x10 x8)", h, B * 1200, pre[2], 1200, False, torch.float32)):
    us = t(lambda: ops._dense_raw(xx, stride, B, p, K, relu, odt))
    fl = 2.0 * S * B * K * p.w.shape[1]
    print("dense %s: %.2f us = %.1f TFLOP/s" % (name, us, fl / us / 1e6))
# layer 2 with 128-B aligned activation rows (ldx = 1216)
hp = torch.zeros(S, B, 1216, device=dev, dtype=torch.bfloat16)
hp[:, :, :1200] = h
yb = torch.empty(S, B, 1200, device=dev, dtype=torch.bfloat16)
def l2_padded():
    _lib.check(lib.bnn_dense_forward(_lib.ptr(hp), B * 1216, 1216, _lib.ptr(pre[1].w), 1200 * 1216, 1216, _lib.ptr(pre[1].b), 1200,
                                     _lib.ptr(yb), B * 1200, 1200, B, 1200, 1200, S, _lib.FLAG_RELU | _lib.FLAG_Y_BF16, _lib.stream_ptr(dev)), "dense")
us = t(l2_padded)
print("dense layer 2, activation rows padded to 1216 (128-B aligned): %.2f us = %.1f TFLOP/s" % (us, 2.0 * S * B * 1200 * 1200 / us / 1e6))
# fixed cost of a dense launch at the layer-2 grid: ONE 64-k step (prologue + epilogue + launch ramp)
w1 = torch.zeros(S, 1200, 64, device=dev, dtype=torch.bfloat16)
p1 = ops.Predrawn(w1, pre[1].b, None, None)
h64 = hp[:, :, :64]
us = t(lambda: ops._dense_raw(h64, B * 1216, B, p1, 64, True, torch.bfloat16, ldx=1216, pad_rows=True))
print("dense layer-2 grid with K = 64 overhead probe (1 step): %.2f us" % us)
# 4096^3, one sample
M = N = K = 4096
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(1, N, K, device=dev) * 0.02).bfloat16()
pw = ops.Predrawn(w, None, None, None)
us = t(lambda: ops._dense_raw(a, 0, M, pw, K, False, torch.bfloat16), 10)
print("dense 4096^3 bf16 (random operands): %.1f us = %.1f TFLOP/s" % (us, 2.0 * M * N * K / us / 1e6))
