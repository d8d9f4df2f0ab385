"""K1 / K3 streaming legs of bench.py alone (diagnostic): python tools/bench_k1.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianneuralnetworks_amd import _lib  # noqa: E402

if os.environ.get("BNN_AB_LIB"):          # A/B against another build of the library
    _lib.LIB_PATH = os.environ["BNN_AB_LIB"]
import bench  # noqa: E402

dev = torch.device("cuda:0")
print(json.dumps({"env": os.environ.get("BNN_K1_KEYS"), "lib": _lib.LIB_PATH, "sampler": bench.sampler_roofline(dev), "kl": bench.kl_roofline(dev)}))
