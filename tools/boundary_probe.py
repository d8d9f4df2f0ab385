"""Cost of a dependent kernel boundary as this process sees it: graphs of N trivial kernels / N head launches, eager streams, and
the bench step itself.   usage: [env ...] python tools/boundary_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd import _lib, ops
import bench
dev = torch.device("cuda:0")
lib = _lib.load()
a = torch.zeros(64, device=dev)
cell = torch.zeros(4, dtype=torch.int32, device=dev)
def trivial():
    lib.bnn_rng_advance(_lib.ptr(cell), 1, _lib.stream_ptr(dev))
def tiny_torch():
    a.add_(1.0)
print("env:", {k: v for k, v in os.environ.items() if k.startswith(("HIP_", "DEBUG_CLR", "ROC_", "GPU_", "HSA_")) and k != "HSA_ENABLE_IPC_MODE_LEGACY"})
print("graph of 10 x bnn_rng_advance (1 thread): %.2f us per kernel" % bench._graph_time(trivial, dev))
print("graph of 10 x torch add_ on 64 floats   : %.2f us per kernel" % bench._graph_time(tiny_torch, dev))
# eager back-to-back on one stream
for name, fn in (("bnn_rng_advance", trivial), ("torch add_", tiny_torch)):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    torch.cuda.synchronize()
    print("eager %-16s: %.2f us per kernel" % (name, (time.perf_counter() - t0) / 2000 * 1e6))
post = bench.posteriors(0)
net = bench.build_net(dev, post)
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
bnn.manual_seed(2); bnn.set_compute("bf16")
step = bench.Step(net, bench.resident_input(x, "bf16"), 0, 1, True)
dt = bench.time_steps(step, 500, 50, 1, dev)
print("bench step, one stream, graph: %.2f us per step" % (dt / 500 * 1e6))
