"""Timing of the sampled conv2d (implicit GEMM) at the BASELINE conv shapes (configs[2], configs[3])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd.nn import NormalConv2d
from bayesianneuralnetworks_amd import _mc
dev = torch.device("cuda:0")
S = 8
for name, (B, C, O, HW, k, s, p) in {"configs[2] LeNet conv 64->64 k3 s2 p1 on 6x6, B=1024": (1024, 64, 64, 6, 3, 2, 1),
                                     "configs[3] CIFAR conv 128->128 k3 p1 on 4x4, B=256": (256, 128, 128, 4, 3, 1, 1)}.items():
    layer = NormalConv2d(C, O, k, stride=s, padding=p).to(dev)
    x = torch.randn(S * B, C, HW, HW, device=dev)
    OH = (HW + 2 * p - k) // s + 1
    flop = 2.0 * S * B * OH * OH * O * C * k * k
    for mode in ("bf16", "f32"):
        bnn.set_compute(mode)
        import bench
        with torch.no_grad(), _mc.McContext(S, B, 0):
            us = bench._graph_time(lambda: layer(x), dev)      # replayed from a HIP graph: eager calls are host-bound here
        print("%s %s: %.1f us per 8-sample launch, %.1f TFLOP/s, %.0f MC-samples/s" % (name, mode, us, flop / us / 1e6, S / us * 1e6))
