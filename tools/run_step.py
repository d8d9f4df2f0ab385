"""The bench step's kernels alone, eager (rocprofv3 --pmc serialises kernels anyway): python tools/run_step.py [n] [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2)
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
bnn.set_compute(mode)
step = bench.Step(net, bench.resident_input(x, mode), 0, 1, False)
for _ in range(n):
    step.run()
torch.cuda.synchronize()
