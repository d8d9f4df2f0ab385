"""The 8-9 us between two graph launches on one stream: same exec relaunched vs two / four execs alternating, vs k steps captured in
one graph.   usage: python tools/graph_gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2); bnn.set_compute("bf16")
xin = bench.resident_input(x, "bf16")
steps = [bench.Step(net, xin, 0, 1, True, private=True) for _ in range(4)]
def run(objs, n=2000):
    for i in range(200): objs[i % len(objs)].run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): objs[i % len(objs)].run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("one exec relaunched on one stream : %.2f us per step" % run(steps[:1]))
print("two execs alternating, one stream : %.2f us per step" % run(steps[:2]))
print("four execs alternating, one stream: %.2f us per step" % run(steps[:4]))
# k steps per graph
for k in (2, 4):
    st = bench.Step(net, xin, 0, 1, False, private=False)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        st._body(); st._body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(k): st._body()
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): g.replay()
    torch.cuda.synchronize()
    print("%d steps per graph, one stream      : %.2f us per step" % (k, (time.perf_counter() - t0) / 1000 / k * 1e6))
