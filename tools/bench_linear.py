"""Per-kernel timing of the fused linear kernel variants at the BASELINE layer shapes (GPU box)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
lib = _lib.load()
dev = torch.device("cuda:0")
S, M = 8, 512

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (K, N) in [(1200, 1200), (784, 1200), (1200, 10)]:
    x = torch.randn(S, M, K, device=dev)
    mu = torch.randn(N, K, device=dev) * 0.05
    rho = torch.full((N, K), -2.0, device=dev)
    mub = torch.zeros(N, device=dev); rhob = torch.full((N,), -2.0, device=dev)
    w = torch.randn(S, N, K, device=dev) * 0.05
    y = torch.empty(S, M, N, device=dev)
    kw = ops._rng_struct(DrawKey(1, 1, 0, S, 0), dev); kb = ops._rng_struct(DrawKey(1, 2, 0, S, 0), dev)
    st = _lib.stream_ptr(dev)
    for comp, name in ((1, "bf16"), (0, "f32")):
        def sampled():
            lib.bnn_linear_forward_sampled(_lib.ptr(x), M * K, K, _lib.ptr(mu), _lib.ptr(rho), _lib.ptr(mub), _lib.ptr(rhob),
                                           _lib.ptr(y), M * N, N, M, N, K, S, ctypes.byref(kw), ctypes.byref(kb), comp, 0, st)
        def plain():
            lib.bnn_linear_forward(_lib.ptr(x), M * K, K, _lib.ptr(w), N * K, None, 0, _lib.ptr(y), M * N, N, M, N, K, S, comp, 0, st)
        ts, tp = timeit(sampled), timeit(plain)
        fl = 2.0 * S * M * N * K
        print("K=%4d N=%4d %-4s sampled %7.1f us (%6.1f TF/s)   plain-W %7.1f us (%6.1f TF/s)   draws/us %.0f"
              % (K, N, name, ts, fl / ts / 1e6, tp, fl / tp / 1e6, S * N * K / ts))
