"""VALU cost of the eps draw per stage and occupancy (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
out = torch.empty(256 * 32 * 256, device=dev)
iters = 2000
print("stage blocks waves/SIMD  ms   Gdraw/s  SIMD-cycles/draw(@2.4GHz)")
for stage in range(4):
    for blocks in (256, 512, 1024, 2048, 4096):
        for _ in range(2):
            lib.bnn_diag_sampler(_lib.ptr(out), blocks, iters, stage, _lib.stream_ptr(dev))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.bnn_diag_sampler(_lib.ptr(out), blocks, iters, stage, _lib.stream_ptr(dev))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        draws = blocks * 256 * iters * 4
        gps = draws / (ms * 1e-3) / 1e9
        wps = blocks * 4 / 1024.0
        # SIMD-cycles per draw = 1024 SIMDs * 2.4e9 / draws-per-second (valid when >= 1 wave/SIMD everywhere)
        print("%d %5d %5.1f %8.3f %8.1f %6.2f" % (stage, blocks, wps, ms, gps, 1024 * 2.4 / gps))
