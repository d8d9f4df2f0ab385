"""Per-step start / end events in the pipelined bench (no profiler): how long does a step take inside the pipeline and how long
does its stream sit idle before the next one?   usage: python tools/pipe_events.py [inflight]"""
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
n = 400
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
base.record()
t0 = time.perf_counter()
for i in range(n):
    k = i % depth
    with torch.cuda.stream(pipe.streams[k]):
        ev[i][0].record()
        pipe.steps[k].run()
        ev[i][1].record()
torch.cuda.synchronize()
t1 = time.perf_counter()
st = [base.elapsed_time(a) * 1e3 for a, b in ev]
en = [base.elapsed_time(b) * 1e3 for a, b in ev]
lat = [e - s for s, e in zip(st, en)]
gap = [st[i] - en[i - depth] for i in range(depth, n)]
half = n // 2
print("%d in flight: %.2f us per step overall; step latency inside the pipeline mean %.1f us (min %.1f max %.1f); idle gap on a stream before its next step mean %.1f us"
      % (depth, (t1 - t0) / n * 1e6, sum(lat[half:]) / (n - half), min(lat[half:]), max(lat[half:]), sum(gap[half:]) / len(gap[half:])))
for i in range(half, half + 12):
    print("  step %3d stream %d: start %8.1f end %8.1f (%.1f us)" % (i, i % depth, st[i] - st[half], en[i] - st[half], lat[i]))
