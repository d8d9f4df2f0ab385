"""Which torch elementwise / copy kernels a bench step launches (torch.profiler over one step): a diagnostic for stray host-side ops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import bayesianneuralnetworks_amd as bnn
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
bnn.manual_seed(2); bnn.set_compute("bf16")
net = bench.build_net(dev, bench.posteriors(0))
x = bench.resident_input(torch.randn(bench.BATCH, bench.DIMS[0]).to(dev), "bf16")
st = bench.Step(net, x, 0, 1, False)
st._body(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    st._body(); torch.cuda.synchronize()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and not e.name.startswith(("aten::empty", "aten::view", "aten::as_strided", "aten::reshape", "aten::slice", "aten::select", "aten::detach", "aten::_unsafe_view", "aten::unbind", "aten::alias")):
        print(e.name, e.input_shapes, [s for s in (e.stack or [])[:6]])
