"""Is the pipelined bench host-bound?  Time the submission loop (before the final sync) against the whole timed region, and the
bare graph launch rate of a trivial graph.   usage: python tools/host_bound.py [inflight]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2); bnn.set_compute("bf16")
pipe = bench.PipelinedSteps(net, bench.resident_input(x, "bf16"), depth)
for n in (200, 1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.run()
    t1 = time.perf_counter()
    pipe.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d in flight, %4d steps: submission loop %.2f us per step, whole region %.2f us per step" % (depth, n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# bare replay cost of the step graphs without the Python around them
gs = [s.graph for s in pipe.steps]; sts = pipe.streams
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1000):
    with torch.cuda.stream(sts[i % depth]):
        gs[i % depth].replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("bare graph.replay() round-robin: submission %.2f us per step, whole %.2f us per step" % ((t1 - t0) / 1e3 * 1e6 / 1e3, (t2 - t0) / 1e3 * 1e6 / 1e3))
# a trivial one-kernel graph: the host's graph launch rate
a = torch.zeros(64, device=dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    a.add_(1); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        a.add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("trivial 1-kernel graph: submission %.2f us per replay, whole %.2f us per replay" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
