"""The bench step as ONE captured graph, replayed back to back on one stream (for rocprofv3 --kernel-trace timelines).
usage: python tools/run_step_graph.py [n] [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2)
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
bnn.set_compute(mode)
step = bench.Step(net, bench.resident_input(x, mode), 0, 1, True)
for _ in range(n):
    step.run()
torch.cuda.synchronize()
