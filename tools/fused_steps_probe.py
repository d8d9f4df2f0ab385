"""What T steps fused into the same launches would cost per step: draw for 8 T samples, layer 1 / layer 2 + head on 8 T samples,
one reduction -- graph timing, one stream.   usage: python tools/fused_steps_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
B = 512
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
x = torch.zeros(B, 832, dtype=torch.bfloat16, device=dev); x[:, :784] = torch.randn(B, 784, device=dev); xv = x[:, :784]
mus = [t for p in post for t in (p[0].reshape(-1), p[2])]; rhos = [t for p in post for t in (p[1].reshape(-1), p[3])]
klo = torch.zeros(7, device=dev)
for T in (1, 2, 4):
    S = 8 * T
    layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=1), DrawKey(1, 2 * i + 2, 0, S, 0, gen=1)) for i, (mw, rw, mb, rb) in enumerate(post)]
    pre = ops.draw_layers(layers, S)
    out = torch.zeros(B, 10, device=dev)
    def draw():
        kh = ops.kl_normal_begin(mus, rhos, [(0.0, 0.1)] * 6, 1.0, out=klo, carry=True)
        ops.draw_layers(layers, S, kl=kh)
        ops._tls.kl_carry = None
    def l1():
        return ops._dense_raw(xv, 0, B, pre[0], 784, True, torch.bfloat16, ldx=832, pad_rows=True)
    h = l1()
    def l2():
        return ops._dense_head_raw(h, h.stride(0), B, pre[1], 1200, True, pre[2], ldx=h.stride(-2))
    hp = l2()
    hp8 = ops.HeadPartials(hp.p[:, :8].contiguous())
    def whole():
        draw(); hh = l1(); q = ops._dense_head_raw(hh, hh.stride(0), B, pre[1], 1200, True, pre[2], ldx=hh.stride(-2)); ops.mc_mean(ops.HeadPartials(q.p[:, :8]) if False else hp8, out=out)
    ud, u1, u2 = bench._graph_time(draw, dev), bench._graph_time(l1, dev), bench._graph_time(l2, dev)
    um = bench._graph_time(lambda: ops.mc_mean(hp8, out=out), dev)
    uw = bench._graph_time(whole, dev, reps=4)
    print("T = %d (S = %2d): draw %.2f, layer 1 %.2f, layer 2 + head %.2f, reduction (one step's) %.2f us per launch -> per step %.2f (kernels one by one), "
          "chain of the four in one graph %.2f per step" % (T, S, ud, u1, u2, um, (ud + u1 + u2) / T + um, uw / T))
