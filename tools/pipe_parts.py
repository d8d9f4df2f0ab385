"""What the pipelined step would cost WITHOUT one of its kernels: four graphs on four streams (as bench.PipelinedSteps), each the
step's launches on fixed drawn weights -- all four kernels, no draw, draw only, dense only.   usage: python tools/pipe_parts.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
S, B = 8, 512
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
x = torch.zeros(B, 832, dtype=torch.bfloat16, device=dev); x[:, :784] = torch.randn(B, 784, device=dev); xv = x[:, :784]


def make(parts):
    layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=1), DrawKey(1, 2 * i + 2, 0, S, 0, gen=1)) for i, (mw, rw, mb, rb) in enumerate(post)]
    pre = ops.draw_layers(layers, S)
    out = torch.zeros(B, 10, device=dev)
    mus = [t for p in post for t in (p[0].reshape(-1), p[2])]; rhos = [t for p in post for t in (p[1].reshape(-1), p[3])]
    klo = torch.zeros(7, device=dev)

    def body():
        p_ = pre
        h = None
        if "draw" in parts:
            kh = ops.kl_normal_begin(mus, rhos, [(0.0, 0.1)] * 6, 1.0, out=klo, carry=True)
            p_ = ops.draw_layers(layers, S, kl=kh)
            ops._tls.kl_carry = None
        if "dense" in parts:
            h = ops._dense_raw(xv, 0, B, p_[0], 784, True, torch.bfloat16, ldx=832, pad_rows=True)
            hp = ops._dense_head_raw(h, h.stride(0), B, p_[1], 1200, True, p_[2], ldx=h.stride(-2))
            if "mc" in parts:
                ops.mc_mean(hp, out=out)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        body(); body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    return g, s


for parts in (("draw", "dense", "mc"), ("dense", "mc"), ("dense",), ("draw",)):
    gs = [make(parts) for _ in range(4)]
    for depth in (1, 4):
        for _ in range(100):
            for g, s in gs[:depth]:
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        n = 400
        t0 = time.perf_counter()
        for i in range(n):
            g, s = gs[i % depth]
            with torch.cuda.stream(s):
                g.replay()
        torch.cuda.synchronize()
        print("%-22s %d in flight: %.2f us per step" % (" + ".join(parts), depth, (time.perf_counter() - t0) / n * 1e6))
