"""Phase breakdown of k_conv_bf16 (BNN_CONV_DIAG bit 1 skips the image fill, 2 the k loop, 4 the output copy; timing-only, wrong
outputs): graph timing of the layer call (draw launch + contraction) per configuration.   usage: BNN_CONV_DIAG=n python tools/conv_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd.nn import NormalConv2d
from bayesianneuralnetworks_amd import _mc
import bench
dev = torch.device("cuda:0"); S = 8
out = []
for name in ("lenet", "cifar"):
    B, C, O, HW, k, s, p = {"lenet": (1024, 64, 64, 6, 3, 2, 1), "cifar": (256, 128, 128, 4, 3, 1, 1)}[name]
    layer = NormalConv2d(C, O, k, stride=s, padding=p).to(dev)
    x = torch.randn(S * B, C, HW, HW, device=dev)
    bnn.set_compute("bf16")
    with torch.no_grad(), _mc.McContext(S, B, 0):
        us = bench._graph_time(lambda: layer(x), dev)
    out.append("%s %.2f us" % (name, us))
print("BNN_CONV_DIAG=%s: %s (layer call = draw + contraction)" % (os.environ.get("BNN_CONV_DIAG", "0"), ", ".join(out)))
