"""The whole-network bench legs alone (for rocprofv3 / quick timing): python tools/run_legs.py [lenet|wide|both]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("lenet", "both"):
    print(json.dumps(bench.lenet_net_roofline("bf16", dev)))
if which in ("wide", "both"):
    print(json.dumps(bench.wide_stack_roofline(dev)))
