"""bench.PipelinedSteps (4 captured steps in flight on 4 streams) for a kernel-trace timeline.  usage: python tools/run_pipe.py [steps] [inflight]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2)
bnn.set_compute("bf16")
os.environ["BNN_BENCH_PREROLL"] = "64"
pipe = bench.PipelinedSteps(net, bench.resident_input(x, "bf16"), depth)
for _ in range(n):
    pipe.run()
pipe.finish()
torch.cuda.synchronize()
