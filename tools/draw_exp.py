"""Draw launch (BASELINE net, S = 8) under the RNG timing experiments of k_draw_multi (BNN_DRAW_EXP, set per process) with and
without the KL first pass riding along.   usage: BNN_DRAW_EXP=n python tools/draw_exp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
S = int(os.environ.get("S", "8"))
G = int(os.environ.get('GEN', '0'))
layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=G), DrawKey(1, 2 * i + 2, 0, S, 0, gen=G)) for i, (mw, rw, mb, rb) in enumerate(post)]
us = bench._graph_time(lambda: ops.draw_layers(layers, S), dev)
mus = [t for p in post for t in (p[0].reshape(-1), p[2])]
rhos = [t for p in post for t in (p[1].reshape(-1), p[3])]
out = torch.zeros(7, device=dev)
def with_kl():
    h = ops.kl_normal_begin(mus, rhos, [(0.0, 0.1)] * 6, 1.0, out=out, carry=True)
    ops.draw_layers(layers, S, kl=h)
    ops._tls.kl_carry = None
us_kl = bench._graph_time(with_kl, dev)
print("GEN=%d BNN_DRAW_EXP=%s S=%d: draw %.2f us, draw + KL first pass %.2f us" % (G, os.environ.get("BNN_DRAW_EXP", "0"), S, us, us_kl))
