"""Isolated probe of the split-K head kernel (one configuration per process)."""
import sys, os
os.environ["BNN_SPLITK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd.nn import NormalLinear
from bayesianneuralnetworks_amd import _mc
mode, S, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = NormalLinear(1200, 10).to(dev)
bnn.set_compute("bf16")
x = torch.randn(S * B, 1200, device=dev)
if mode == "bf16":
    x = x.bfloat16()
with torch.no_grad(), _mc.McContext(S, B, 0):
    y = layer(x)
    torch.cuda.synchronize()
print(mode, S, B, "ok", float(y.abs().mean()), "nan" if torch.isnan(y).any() else "")
