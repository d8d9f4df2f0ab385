#!/usr/bin/env python3
"""Training-step profiling target: `python bench.py --mode train` without the forward / CPU legs.
usage: tools/bench_train.py [--steps 50] [--dtype bf16|f32] [--no-graph]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
ap.add_argument("--no-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
import bayesianneuralnetworks_amd as bnn  # noqa: E402
bnn.set_compute(args.dtype)
bnn.manual_seed(2)
net = bench.build_net(dev, bench.posteriors(0))
x = torch.randn(bench.BATCH, bench.DIMS[0], generator=torch.Generator().manual_seed(1)).to(dev)
step = bench.TrainStep(net, x.bfloat16() if args.dtype == "bf16" else x, 0, 1, not args.no_graph)
dt = bench.time_steps(step, args.steps, args.warmup, 1, dev)
print(json.dumps({"train_mc_samples_per_s": round(bench.SAMPLES * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4),
                  "dtype": args.dtype, "hip_graph": not args.no_graph, "loss": round(float(step.loss), 4)}))
