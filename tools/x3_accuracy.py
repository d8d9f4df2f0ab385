import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bayesianneuralnetworks_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(21)
for K in (1200, 4096):
    S, M, Nn = 2, 512, 1200
    x = torch.randn(S, M, K, generator=gen)
    w = torch.randn(S, Nn, K, generator=gen) * 0.05
    want = torch.einsum("smk,snk->smn", x.double(), w.double())
    scale = float(torch.einsum("smk,snk->smn", x.double().abs(), w.double().abs()).mean())
    y = ops.linear_plain(x.to(dev), w.to(dev), None, False, "f32").cpu().double()
    ref32 = torch.einsum("smk,snk->smn", x, w).double()
    print("K=%d mode=%s  max err / mean sum|ab| = %.3e   rms err / rms y = %.3e   (torch CPU fp32 einsum: %.3e)" % (
        K, os.environ.get("BNN_F32_MFMA", "x3"), float((y - want).abs().max()) / scale,
        float((y - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()), float((ref32 - want).abs().max()) / scale))
