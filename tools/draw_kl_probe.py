"""Draw launch of the BASELINE net for S samples: alone, with the whole KL riding in its items, with the KL of a 1 / G slice of every tensor
(what a rank of a G-GPU job carries) as piggy-back workgroups.   usage: python tools/draw_kl_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops, distributed as bd
from bayesianneuralnetworks_amd._rng import DrawKey
import bench
lib = _lib.load(); dev = torch.device("cuda:0")
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
for S, G in ((8, 1), (4, 2), (2, 4), (1, 8)):
    layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=1), DrawKey(1, 2 * i + 2, 0, S, 0, gen=1)) for i, (mw, rw, mb, rb) in enumerate(post)]
    us = bench._graph_time(lambda: ops.draw_layers(layers, S), dev)
    mus = [t for p in post for t in (p[0].reshape(-1), p[2])]
    rhos = [t for p in post for t in (p[1].reshape(-1), p[3])]
    out = torch.zeros(7, device=dev)

    def with_kl(ms, rs):
        h = ops.kl_normal_begin(ms, rs, [(0.0, 0.1)] * 6, 1.0, out=out, carry=True)
        ops.draw_layers(layers, S, kl=h)
        ops._tls.kl_carry = None
    us_full = bench._graph_time(lambda: with_kl(mus, rhos), dev)
    sl = [bd.shard_range(m.numel(), 0, G) for m in mus]
    ms = [m[lo:hi] for m, (lo, hi) in zip(mus, sl)]
    rs = [r[lo:hi] for r, (lo, hi) in zip(rhos, sl)]
    us_slice = bench._graph_time(lambda: with_kl(ms, rs), dev)
    print("S = %d: draw %.2f us; + whole KL %.2f; + KL of a 1/%d slice %.2f" % (S, us, us_full, G, us_slice))
