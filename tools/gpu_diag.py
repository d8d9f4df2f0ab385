"""Numerics diagnostics on the GPU box (not a test): error of each stage vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianneuralnetworks_amd as bnn
from bayesianneuralnetworks_amd import ops
from bayesianneuralnetworks_amd._rng import DrawKey
from bayesianneuralnetworks_amd.nn import NormalLinear
from oracle import oracle as orc

dev = torch.device("cuda:0")
key = DrawKey(0xDEADBEEF12345, 9, 0, 1, 3)
n = 1 << 22
e = ops.eps_philox((n,), key, dev)[0].cpu().numpy()
w = orc.eps_fill(key.seed, key.stream, 0, key.epoch_host, 0, (n,), key.gen)
d = np.abs(e - w)
print("eps: max abs err %.3e  rms err %.3e  at |eps|=%.3f" % (d.max(), np.sqrt((d**2).mean()), abs(w[d.argmax()])))
for (M, K, Nn) in [(512, 784, 1200), (512, 1200, 1200), (512, 1200, 10), (4096, 4096, 4096)]:
    if M == 4096:
        continue
    torch.manual_seed(0)
    layer = NormalLinear(K, Nn).to(dev)
    bnn.manual_seed(5)
    x = torch.randn(M, K, device=dev)
    y = layer(x).detach().cpu().numpy()
    kw, kb = layer.weight.draw_key, layer.bias.draw_key
    ew = orc.eps_fill(kw.seed, kw.stream, 0, kw.epoch_host, 0, (Nn, K), kw.gen)
    eb = orc.eps_fill(kb.seed, kb.stream, 0, kb.epoch_host, 0, (Nn,), kb.gen)
    wo = orc.sample_affine(layer.weight.mean.detach().cpu().numpy(), layer.weight.scale.detach().cpu().numpy(), ew)
    bo = orc.sample_affine(layer.bias.mean.detach().cpu().numpy(), layer.bias.scale.detach().cpu().numpy(), eb)
    ws = layer.sampled[0].detach().cpu().numpy()
    yo = orc.linear(x.cpu().numpy(), wo, bo)
    # torch CPU fp32 (what the reference would compute from the same w): its own deviation from exact
    yt = torch.nn.functional.linear(x.cpu(), torch.from_numpy(wo), torch.from_numpy(bo)).numpy()
    rms = np.sqrt((yo**2).mean())
    print("linear %s: w max err %.2e | y max err %.2e (rms(y) %.2f, ratio to 1e-5*(1+|y|): %.2f) | torch-CPU fp32 vs exact: %.2e"
          % ((M, K, Nn), np.abs(ws - wo).max(), np.abs(y - yo).max(), rms,
             (np.abs(y - yo) / (1e-5 + 1e-5 * np.abs(yo))).max(), np.abs(yt - yo).max()))
