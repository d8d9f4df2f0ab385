"""Run the fused weight-gradient kernel repeatedly (profiling target).
usage: run_wgrad.py K N {bf16|f32} [iters]   (x, gy bf16 in bf16 mode; 8 samples x 512 rows)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
K, N = int(sys.argv[1]), int(sys.argv[2])
comp = 1 if sys.argv[3] == "bf16" else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _lib.load(); dev = torch.device("cuda:0"); S, M = 8, 512
_lib.ensure_workspace(dev)
x = torch.randn(S, M, K, device=dev); gy = torch.randn(S, M, N, device=dev)
flags = 0
if comp:
    x = x.bfloat16(); gy = gy.bfloat16(); flags = _lib.FLAG_X_BF16 | _lib.FLAG_Y_BF16
rho = torch.full((N, K), -2.0, device=dev)
gm = torch.empty(N, K, device=dev); gr = torch.empty(N, K, device=dev)
kw = ops._rng_struct(DrawKey(1, 1, 0, S, 0, gen=int(os.environ.get("GEN", "0"))), dev)
st = _lib.stream_ptr(dev)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for it in range(iters + 3):
    if it == 3:
        e0.record()
    _lib.check(lib.bnn_linear_backward_weight_sampled(_lib.ptr(x), M * K, K, _lib.ptr(gy), M * N, N, _lib.ptr(rho), _lib.ptr(gm),
                                                      _lib.ptr(gr), None, None, None, M, N, K, S, ctypes.byref(kw), None, None, comp, flags, 0, st), "wgrad")
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print("wgrad K=%d N=%d %s: %.1f us/launch, %.1f TFLOP/s" % (K, N, sys.argv[3], us, 2.0 * S * M * N * K / us / 1e6))
