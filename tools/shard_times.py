"""Per-rank step time of the strong-scaling shards (S = 8 / G samples per rank), measured on ONE GPU without the collective:
what a rank of a G-GPU job computes per step.   usage: python tools/shard_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
import bench

dev = torch.device("cuda:0")
bnn.manual_seed(2); bnn.set_compute("bf16")
net = bench.build_net(dev, bench.posteriors(0))
x = bench.resident_input(torch.randn(bench.BATCH, bench.DIMS[0]).to(dev), "bf16")
base = None
for G in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    cnt = bench.SAMPLES // G
    kw = dict(samples=cnt, sample0=0, total_samples=bench.SAMPLES)
    st = bench.Step(net, x, 0, 1, True, **kw)
    t1 = bench.time_steps(st, 1000, 200, 1, dev) / 1000
    pipe = bench.PipelinedSteps(net, x, 4, **kw)
    t3 = bench.time_steps(pipe, 1000, 200, 1, dev) / 1000
    rate = bench.SAMPLES / t3
    base = rate if base is None else base
    print("G = %d (%d samples per rank): one stream %.1f us per step, 4 in flight %.1f us -> a %d-GPU job at best %.0f k MC-samples/s = %.2f of %d x the 1-GPU rate"
          % (G, cnt, t1 * 1e6, t3 * 1e6, G, rate / 1e3, rate / (G * base), G))
    del st, pipe
