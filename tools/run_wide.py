"""configs[4] layer launches alone (rocprofv3): NormalLinear(4096, 4096), batch 4096, fp32 mode.  python tools/run_wide.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd.nn import NormalLinear
from bayesianneuralnetworks_amd import _mc
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(12)
layer = NormalLinear(4096, 4096).to(dev)
layer.compute = "f32"
x = torch.randn(4096, 4096, device=dev)
with torch.no_grad(), _mc.McContext(1, 4096, 0):
    for _ in range(n):
        layer(x)
torch.cuda.synchronize()
