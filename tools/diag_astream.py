"""L2 -> CU activation-stream rate per access shape / occupancy (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
S, M, K = 8, 512, 1216   # K multiple of 64 close to 1200
x = torch.randn(S, M, K, device=dev)
out = torch.empty(1 << 22, device=dev)
print("pattern waves rows/WG ntn  WGs   us    TB/s  B/clk/CU(@2.4GHz, CUs busy)")
for pattern in (0, 1, 2):
    for nw, rows in ((8, 512), (8, 256), (4, 256), (4, 128), (16, 512)):
        for ntn in (15, 25):
            grid = S * (M // rows) * ntn
            def run():
                rc = lib.bnn_diag_astream(_lib.ptr(x), S, M, K, rows, ntn, pattern, nw, _lib.ptr(out), _lib.stream_ptr(dev))
                assert rc == 0, lib.bnn_last_error()
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            byts = grid * rows * K * 4
            cus = min(grid, 256)
            print("%d %5d %6d %4d %5d %7.1f %6.2f %7.1f" % (pattern, nw, rows, ntn, grid, us, byts / us / 1e6,
                                                          byts / cus / (us * 2400)))
