"""Where a dense workgroup's time goes: s_memrealtime stamps of consumer wave 0 of every workgroup of the layer-2 launch
(BNN_DENSE_DIAG=6 build path).   usage: python tools/dense_stamps.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(4096 * 5, dtype=torch.int64, device=dev)
os.environ["BNN_DENSE_DIAG"] = os.environ.get("STAMP_DIAG", "6")
os.environ["BNN_DENSE_STAMPS"] = hex(buf.data_ptr())
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
import numpy as np
import bench
S, B = 8, int(os.environ.get('STAMP_B', '512'))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
NN = int(os.environ.get('STAMP_N', '1200'))
mw = torch.randn(NN, K, device=dev) * 0.03; rw = torch.full((NN, K), -2.0, device=dev)
mb = torch.zeros(NN, device=dev); rb = torch.full((NN,), -2.0, device=dev)
pre = ops.draw_layers([(mw, rw, mb, rb, DrawKey(1, 1, 0, S, 0, gen=1), DrawKey(1, 2, 0, S, 0, gen=1))], S)[0]
ld = (K + 63) // 64 * 64
hb = torch.zeros(S, B, ld, dtype=torch.bfloat16, device=dev); hb[:, :, :K] = torch.randn(S, B, K, device=dev).relu_()
h = hb[:, :, :K]
for it in range(5):
    buf.zero_()
    torch.cuda.synchronize()
    ops._dense_raw(h, B * ld, B, pre, K, True, torch.bfloat16, ldx=ld, pad_rows=True)
    torch.cuda.synchronize()
tile = int(os.environ.get('BNN_DENSE_TILE', '1'))
bm, bn = {0: (256, 80), 1: (128, 160), 2: (256, 128), 3: (64, 160), 4: (32, 160)}[tile]
NWG = S * ((B + bm - 1) // bm) * ((NN + bn - 1) // bn)
st = buf.cpu().numpy().reshape(-1, 5)[:NWG].astype(np.float64) * 0.01    # us
t0 = st[:, 0].min()
def q(v): return "p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(v, [10, 50, 90, 100]))
print("DIAG %s, tile %d, N = %d, K = %d (%d k-steps), %d workgroups, times in us" % (os.environ["BNN_DENSE_DIAG"], tile, NN, K, (K + 63) // 64, NWG))
print("workgroup start after the first      : " + q(st[:, 0] - t0))
print("entry -> first stage landed          : " + q(st[:, 1] - st[:, 0]))
print("main loop                            : " + q(st[:, 2] - st[:, 1]) + "   (per k-step p50 %.3f)" % (np.percentile(st[:, 2] - st[:, 1], 50) / ((K + 63) // 64)))
print("epilogue until stores issued         : " + q(st[:, 3] - st[:, 2]))
print("stores issued -> retired             : " + q(st[:, 4] - st[:, 3]))
print("workgroup lifetime                   : " + q(st[:, 4] - st[:, 0]))
print("launch span (first entry -> last end): %.2f" % (st[:, 4].max() - t0))
