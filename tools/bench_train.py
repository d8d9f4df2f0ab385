#!/usr/bin/env python3
"""Training step of the BASELINE MLP: the loop body of the reference's
examples/MNIST/train.py:53-65 -- zero_grad, S-sample forward, KL, mean cross-entropy over the
samples, backward, Adam -- with every contraction, draw and draw-backward in HIP.

    python tools/bench_train.py [--steps 50] [--dtype bf16|f32] [--graph]

Prints one JSON line: training-step MC-samples/s and ms per step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib
    from bayesianneuralnetworks_amd.nn import KLDivergence
    from bayesianneuralnetworks_amd._rng import default_generator
    lib = _lib.load()
    net = bench.build_net(dev, bench.posteriors(0))
    bnn.set_compute(args.dtype)
    bnn.manual_seed(2)
    S, B = bench.SAMPLES, bench.BATCH
    x = torch.randn(B, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
    if args.dtype == "bf16":
        x = x.bfloat16()
    target = torch.randint(0, 10, (B,), generator=torch.Generator().manual_seed(3)).to(dev).repeat(S)
    kld = KLDivergence(number_of_batches=100)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=args.graph, foreach=True)
    cell = default_generator.epoch_dev(dev)
    loss_out = torch.zeros((), device=dev)

    def body():
        ys = net.forward_stacked(x, S)                                     # (S, B, 10)
        loss = torch.nn.functional.cross_entropy(ys.reshape(S * B, -1).float(), target) + kld(net)
        loss.backward()
        loss_out.copy_(loss.detach())

    def step_eager():
        opt.zero_grad(set_to_none=True)
        body()
        opt.step()

    graph = None
    if args.graph:
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                step_eager()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            body()
            opt.step()
            # fresh noise on every replay, bumped AFTER the backward re-created this step's draws
            _lib.check(lib.bnn_rng_advance(_lib.ptr(cell), 1, _lib.stream_ptr(dev)), "bnn_rng_advance")

    def run():
        if graph is not None:
            graph.replay()
        else:
            step_eager()

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    first = float(loss_out)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "MC-samples/sec, TRAINING step (fwd + KL + CE + bwd + Adam), 784-1200-1200-10, batch 512",
                      "value": round(S * args.steps / dt, 1), "unit": "MC-samples/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                      "steps": args.steps, "dtype": args.dtype, "hip_graph": bool(args.graph),
                      "loss_first": round(first, 5), "loss_last": round(float(loss_out), 5)}), flush=True)


if __name__ == "__main__":
    main()
