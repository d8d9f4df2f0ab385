"""In-kernel s_memtime stamps of the fused linear kernel (diagnostic build).  GPU box only."""
import sys, os, ctypes, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(8000 * 2, dtype=torch.int64, device=dev)
os.environ["BNN_STAMPS"] = hex(dbg.data_ptr())
from bayesianneuralnetworks_amd import _lib, ops
from bayesianneuralnetworks_amd._rng import DrawKey
lib = _lib.load(); S, M, K, N = 8, 512, 1200, 1200
x = torch.randn(S, M, K, device=dev)
if os.environ.get("STAMP_ABF"): x = x.bfloat16()
mu = torch.randn(N, K, device=dev) * 0.05; rho = torch.full((N, K), -2.0, device=dev)
mub = torch.zeros(N, device=dev); rhob = torch.full((N,), -2.0, device=dev); y = torch.empty(S, M, N, device=dev)
kw = ops._rng_struct(DrawKey(1, 1, 0, S, 0), dev); kb = ops._rng_struct(DrawKey(1, 2, 0, S, 0), dev)
for it in range(3):
    dbg.zero_()
    lib.bnn_linear_forward_sampled(_lib.ptr(x), M * K, K, _lib.ptr(mu), _lib.ptr(rho), _lib.ptr(mub), _lib.ptr(rhob),
                                   _lib.ptr(y), M * N, N, M, N, K, S, ctypes.byref(kw), ctypes.byref(kb), 1,
                                   2 if x.dtype == torch.bfloat16 else 0, _lib.stream_ptr(dev))
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 2)
for name, base in (("wave 0", 0),):
    rows = [(int(t), int(c)) for t, c in d[base:base + 2000] if c != 0]
    if not rows:
        print(name, "no stamps"); continue
    t0 = rows[0][1]
    print("== %s: %d stamps, total %.2f us (100 MHz ticks?) raw span %d" % (name, len(rows), 0, rows[-1][1] - t0))
    # per-tag deltas: time from previous stamp to this one
    import collections
    acc = collections.defaultdict(list)
    for (pt, pc), (t, c) in zip(rows[:-1], rows[1:]):
        acc[(pt, t)].append(c - pc)
    for k in sorted(acc):
        v = acc[k]
        print("   %3d -> %3d : n=%3d  mean %8.0f  min %6d  max %7d  sum %9d" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v), sum(v)))

    marks = {t: c for t, c in rows if t >= 10}
    if 10 in marks and 13 in marks:
        print("   timeline (ticks): entry->prologue %d, prologue->loop end %d, loop end->stores retired %d, total %d" % (
            marks[11] - marks[10], marks[12] - marks[11], marks[13] - marks[12], marks[13] - marks[10]))
