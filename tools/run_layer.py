"""Run one fused-linear configuration repeatedly (profiling target).
usage: run_layer.py K N {bf16|f32} {sampled|plain} [iters]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
K, N = int(sys.argv[1]), int(sys.argv[2])
comp = 1 if sys.argv[3] == "bf16" else 0
sampled = sys.argv[4] == "sampled"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
lib = _lib.load(); dev = torch.device("cuda:0"); S, M = 8, 512
x = torch.randn(S, M, K, device=dev)
if os.environ.get("ABF"): x = x.bfloat16()
mu = torch.randn(N, K, device=dev) * 0.05; rho = torch.full((N, K), -2.0, device=dev)
mub = torch.zeros(N, device=dev); rhob = torch.full((N,), -2.0, device=dev)
w = torch.randn(S, N, K, device=dev) * 0.05; y = torch.empty(S, M, N, device=dev)
kw = ops._rng_struct(DrawKey(1, 1, 0, S, 0), dev); kb = ops._rng_struct(DrawKey(1, 2, 0, S, 0), dev)
st = _lib.stream_ptr(dev)
for _ in range(iters):
    if sampled:
        lib.bnn_linear_forward_sampled(_lib.ptr(x), M * K, K, _lib.ptr(mu), _lib.ptr(rho), _lib.ptr(mub), _lib.ptr(rhob),
                                       _lib.ptr(y), M * N, N, M, N, K, S, ctypes.byref(kw), ctypes.byref(kb), comp, 2 if x.dtype == torch.bfloat16 else 0, st)
    else:
        lib.bnn_linear_forward(_lib.ptr(x), M * K, K, _lib.ptr(w), N * K, None, 0, _lib.ptr(y), M * N, N, M, N, K, S, comp, 0, st)
torch.cuda.synchronize()
print("done")
