"""Do the draw launch (VALU-bound) and the dense contraction (MFMA / LDS-DMA-bound) overlap when they run on two streams?
usage: python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import bench

lib = _lib.load(); dev = torch.device("cuda:0")
S, B = 8, 512
post = [[t.to(dev) for t in p] for p in bench.posteriors(0)]
layers = [(mw, rw, mb, rb, DrawKey(1, 2 * i + 1, 0, S, 0, gen=1), DrawKey(1, 2 * i + 2, 0, S, 0, gen=1)) for i, (mw, rw, mb, rb) in enumerate(post)]
pre = ops.draw_layers(layers, S)
h = torch.zeros(S, B, 1216, device=dev, dtype=torch.bfloat16)
h[:, :, :1200] = torch.randn(S, B, 1200, device=dev).relu_().bfloat16()
y = torch.empty(S, B, 1216, device=dev, dtype=torch.bfloat16)


def gemm():
    _lib.check(lib.bnn_dense_forward(_lib.ptr(h), B * 1216, 1216, _lib.ptr(pre[1].w), 1200 * 1216, 1216, _lib.ptr(pre[1].b), 1200,
                                     _lib.ptr(y), B * 1216, 1216, B, 1200, 1200, S, _lib.FLAG_RELU | _lib.FLAG_Y_BF16, _lib.stream_ptr(dev)), "dense")


def draw():
    ops.draw_layers(layers, S)


def graph_of(fn, n, stream):
    with torch.cuda.stream(stream):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(n):
            fn()
    return g


sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
N = 20
ga = graph_of(draw, N, sa)
gb = graph_of(gemm, N, sb)


def run(pairs, reps=20):
    for g, s in pairs:
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in pairs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / N * 1e6


ta = run([(ga, sa)]); tb = run([(gb, sb)]); tab = run([(ga, sa), (gb, sb)])
print("draw alone %.2f us, gemm alone %.2f us, both streams together %.2f us per (draw + gemm) pair: serial sum %.2f, perfect overlap %.2f"
      % (ta, tb, tab, ta + tb, max(ta, tb)))
gb2 = graph_of(gemm, N, sa)
tbb = run([(gb, sb), (gb2, sa)])
print("two gemm streams: %.2f us per pair (alone %.2f each)" % (tbb, tb))
ga2 = graph_of(draw, N, sb)
taa = run([(ga, sa), (ga2, sb)])
print("two draw streams: %.2f us per pair (alone %.2f each)" % (taa, ta))

# one graph with two parallel branches (fork / join inside the capture): does the graph executor run them concurrently?
main = torch.cuda.Stream(dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g2, stream=main):
        sa.wait_stream(main)
        with torch.cuda.stream(sa):
            for _ in range(N):
                draw()
        for _ in range(N):
            gemm()
        main.wait_stream(sa)
tg = run([(g2, main)])
print("ONE graph, draw branch || gemm branch: %.2f us per pair" % tg)

# stream priorities: GEMM chain on a high-priority stream, draws on a low-priority one -- does the dispatcher prefer the GEMM?
lo_, hi_ = torch.cuda.Stream(dev, priority=0), torch.cuda.Stream(dev, priority=-1)
gd = graph_of(draw, N, lo_)
gg = graph_of(gemm, N, hi_)


def timed_pair(reps=10):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    with torch.cuda.stream(lo_):
        e[0].record()
    with torch.cuda.stream(hi_):
        e[2].record()
    for _ in range(reps):
        with torch.cuda.stream(lo_):
            gd.replay()
        with torch.cuda.stream(hi_):
            gg.replay()
    with torch.cuda.stream(lo_):
        e[1].record()
    with torch.cuda.stream(hi_):
        e[3].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps / N * 1e3, e[2].elapsed_time(e[3]) / reps / N * 1e3


timed_pair(2)
td, tg = timed_pair()
print("priorities (draw low, gemm high), both running: draw stream %.2f us per launch, gemm stream %.2f us per launch" % (td, tg))
