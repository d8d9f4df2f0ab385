"""Pipelined steps (bench.PipelinedSteps) for a range of steps in flight; GPU_MAX_HW_QUEUES comes from the environment.
usage: [GPU_MAX_HW_QUEUES=8] python tools/inflight_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench
dev = torch.device("cuda:0")
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2)
bnn.set_compute("bf16")
os.environ["BNN_BENCH_PREROLL"] = "128"
xin = bench.resident_input(x, "bf16")
out = []
for depth in (1, 2, 3, 4, 5, 6, 8):
    pipe = bench.PipelinedSteps(net, xin, depth)
    best = []
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pipe.run()
        pipe.finish()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    best.sort()
    out.append("%d in flight: %.2f us per step (median of 7 x 200 steps; min %.2f)" % (depth, best[3], best[0]))
    del pipe
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)"))
print("\n".join(out))
