"""Host cost of one graph replay against the GPU time of the step it launches (is the pipelined bench host-bound?).
usage: python tools/host_launch_probe.py [steps_in_flight]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
post = bench.posteriors(0)
import bayesianneuralnetworks_amd as bnn
bnn.manual_seed(2); bnn.set_compute("bf16")
net = bench.build_net(dev, post)
x = bench.resident_input(torch.randn(bench.BATCH, bench.DIMS[0]).to(dev), "bf16")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pipe = bench.PipelinedSteps(net, x, depth)
for _ in range(30):
    pipe.run()
pipe.finish(); torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K):
    pipe.run()
t1 = time.perf_counter()
pipe.finish(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("depth %d: host loop %.1f us per replay; until the GPU is done %.1f us per step" % (depth, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
g = pipe.steps[0].graph
t0 = time.perf_counter()
for _ in range(K):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("one graph, one stream: host %.1f us per replay; GPU done %.1f us per step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
