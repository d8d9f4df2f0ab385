"""The step of one rank of a G-GPU strong-scaling job (8 / G samples), eager launches, for a kernel trace.   usage: python tools/run_shard.py G [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
G = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bnn.manual_seed(2); bnn.set_compute("bf16")
net = bench.build_net(dev, bench.posteriors(0))
x = bench.resident_input(torch.randn(bench.BATCH, bench.DIMS[0]).to(dev), "bf16")
st = bench.Step(net, x, 0, 1, False, samples=bench.SAMPLES // G, sample0=0, total_samples=bench.SAMPLES)
for _ in range(n):
    st.run()
torch.cuda.synchronize()
