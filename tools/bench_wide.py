"""BASELINE configs[4] layer (4096 x 4096 NormalLinear, batch 4096, fp32 mode): one sampled forward launch, S MC samples.
usage: bench_wide.py [S]   (BNN_F32_MFMA=native for the v_mfma_f32_16x16x4_f32 path)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
lib = _lib.load(); dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = N = K = 4096
x = torch.randn(M, K, device=dev)
mu = torch.randn(N, K, device=dev) * 0.02; rho = torch.full((N, K), -2.0, device=dev)
mub = torch.zeros(N, device=dev); rhob = torch.full((N,), -2.0, device=dev)
y = torch.empty(S, M, N, device=dev)
kw = ops._rng_struct(DrawKey(1, 1, 0, S, 0), dev); kb = ops._rng_struct(DrawKey(1, 2, 0, S, 0), dev)
st = _lib.stream_ptr(dev)
w = (torch.randn(S, N, K, device=dev) * 0.02)
for comp, name in ((0, "f32"), (1, "bf16"), (10, "f32 draw-once"), (11, "bf16 draw-once")):
    def run_once(c=comp - 10):
        # K1 materialises W_s (fp32), then the same kernel with explicit weights
        ww = ops._sample_affine_philox_raw(mu.view(-1), rho.view(-1), DrawKey(1, 1, 0, S, 0)).view(S, N, K)
        lib.bnn_linear_forward(_lib.ptr(x), 0, K, _lib.ptr(ww), N * K, None, 0, _lib.ptr(y), M * N, N, M, N, K, S, c, 0, st)
    def run():
        lib.bnn_linear_forward_sampled(_lib.ptr(x), 0, K, _lib.ptr(mu), _lib.ptr(rho), _lib.ptr(mub), _lib.ptr(rhob),
                                       _lib.ptr(y), M * N, N, M, N, K, S, ctypes.byref(kw), ctypes.byref(kb), comp, 0, st)
    if comp >= 10: run = run_once
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * S * M * N * K
    print("4096x4096 layer, batch 4096, S=%d, %s (%s): %.3f ms = %.1f TFLOP/s" % (S, name, os.environ.get("BNN_F32_MFMA", "x3") if comp == 0 else "-", ms, fl / ms / 1e9))
