"""Do draw workgroups run BESIDE dense workgroups on a CU (VALU under the MFMAs), or only between them?  A dense workgroup takes
144 KiB of LDS and 2 waves per SIMD; what is left of a SIMD's 512 registers decides whether a draw wave (108 -> 112 registers) fits
next to them: 2 x 200 + 112 = 512 for the layer-2 + head instantiation (YM = 3), 2 x 240 + 112 > 512 for the plain bf16-output one
(YM = 1).  Two streams, one replaying a graph of N dense launches, the other a graph of N draw launches.
usage: python tools/coissue_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
S, B, N = 8, 512, 16
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=1), DrawKey(1, 2 * i + 2, 0, S, 0, gen=1)) for i, (mw, rw, mb, rb) in enumerate(post)]
pre = ops.draw_layers(layers, S)
hb = torch.zeros(S, B, 1216, dtype=torch.bfloat16, device=dev); hb[:, :, :1200] = torch.randn(S, B, 1200, device=dev).relu_()
h = hb[:, :, :1200]


def graph(fn, n):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): fn()
    return g, s


def dense_plain(): ops._dense_raw(h, h.stride(0), B, pre[1], 1200, True, torch.bfloat16, ldx=1216, pad_rows=True)
def dense_head(): ops._dense_head_raw(h, h.stride(0), B, pre[1], 1200, True, pre[2], ldx=1216)
def draw(): ops.draw_layers(layers, S)


def run(pairs, reps=20):
    for _ in range(3):
        for g, s in pairs:
            with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in pairs:
            with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / N * 1e6


gd = graph(draw, N)
ta_d = run([gd])
print("draw alone: %.2f us per launch" % ta_d)
for name, fn in (("dense YM=1 (240 regs)", dense_plain), ("dense + head YM=3 (200 regs)", dense_head)):
    ga = graph(fn, N)
    ta = run([ga])
    tb = run([ga, gd])
    print("%-30s alone %.2f us; beside the draw %.2f us per (dense, draw) pair = %.2f of the sum" % (name, ta, tb, tb / (ta + ta_d)))
