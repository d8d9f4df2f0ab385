"""Round-2 fixtures from the real reference, on the device:
  * MultivariateNormalLinear(128, 10) -- the CIFAR10 example's head: PyTorch-ROCm ops, parity pinned;
  * the shipped MNIST example net end to end (examples/MNIST/model.py:20-33) with its trained checkpoint -- parity mode
    (the reference's own eps) and mc_batched mode (shared deterministic prefix, Philox draws) against the oracle;
  * tolerance bookkeeping for the K >= 784 goldens: |HIP - float64| next to |reference - float64|.
"""
import numpy as np
import pytest
import torch

from conftest import allclose, allclose_scaled, load_golden
import seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib, ops
    from oracle import oracle as orc
    return dict(bnn=bnn, lib=_lib.load(), ops=ops, orc=orc, dev=torch.device("cuda:0"))


def N(t):
    return t.detach().float().cpu().numpy()


def test_mvn_head_on_device_matches_reference_golden(env):
    from bayesianneuralnetworks_amd.nn import MultivariateNormalLinear, KLDivergence, BayesianNetworkModule
    dev = env["dev"]
    g = load_golden("mvn_linear_128x10")
    layer = MultivariateNormalLinear(128, 10)
    with torch.no_grad():
        layer.weight.mean.copy_(torch.from_numpy(g["mu_w"])); layer.weight.scale.copy_(torch.from_numpy(g["scale_w"]))
        layer.bias.mean.copy_(torch.from_numpy(g["mu_b"])); layer.bias.scale.copy_(torch.from_numpy(g["scale_b"]))
    layer = layer.to(dev)
    layer.weight.sample_with_noise(torch.from_numpy(g["u_w"]).to(dev))
    layer.bias.sample_with_noise(torch.from_numpy(g["u_b"]).to(dev))
    layer.sampled = (layer.weight.sampled, layer.bias.sampled)
    assert allclose(N(layer.sampled[0]), g["w"]) and allclose(N(layer.sampled[1]), g["b"])
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    y = layer(x, sample=False)
    assert y.is_cuda and allclose(N(y), g["y"])

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(1, 1, 1)
            self.layers = torch.nn.Sequential(layer)

    kl = KLDivergence(number_of_batches=float(g["n_batches"]))(Net())
    assert kl.is_cuda and abs(kl.item() - float(g["kl"])) <= 1e-5 * abs(float(g["kl"]))
    ((y * torch.from_numpy(g["gy"]).to(dev)).sum() + kl).backward()
    for got, want in ((layer.weight.mean.grad, "g_mu_w"), (layer.weight.scale.grad, "g_scale_w"),
                      (layer.bias.mean.grad, "g_mu_b"), (layer.bias.scale.grad, "g_scale_b"), (x.grad, "g_x")):
        assert allclose(N(got), g[want], 2e-5), want
    # a draw of its own on the device: finite, and different from the fixture's noise
    y3 = layer(x.detach())
    assert torch.isfinite(y3).all() and not torch.equal(y3, y.detach())


def _bcnn(sd, samples):
    """examples/MNIST/model.py:20-33 restated with this package's layers."""
    from torch.nn import Conv2d, BatchNorm2d, ELU, Softmax, Flatten
    from bayesianneuralnetworks_amd.nn import BayesianNetworkModule, NormalConv2d, NormalLinear

    class BCNN(BayesianNetworkModule):
        def __init__(self):
            super().__init__(1, 10, samples)
            self.layers = torch.nn.Sequential(
                Conv2d(1, 32, 5, padding=2, stride=2), BatchNorm2d(32), ELU(),
                Conv2d(32, 32, 3, padding=1, stride=1), ELU(),
                Conv2d(32, 64, 3, padding=0, stride=2), ELU(),
                NormalConv2d(64, 64, 3, padding=1, stride=2), ELU(), Flatten(),
                NormalLinear(576, 10), Softmax(dim=-1))

        def _forward(self, x):
            return self.layers(x)

    net = BCNN()
    missing = net.load_state_dict(sd, strict=True)
    net.eval()
    return net


def test_shipped_mnist_net_end_to_end(env):
    from bayesianneuralnetworks_amd.nn import KLDivergence
    dev, orc = env["dev"], env["orc"]
    g = load_golden("mnist_bcnn_pretrained")
    sd = {k[4:].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd__")}
    net = _bcnn(sd, 2).to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    conv, lin = net.layers[7], net.layers[10]
    kl = KLDivergence()(net)
    assert abs(kl.item() - float(g["kl"])) <= 1e-5 * float(g["kl"])
    # ---- parity mode: the reference's own eps (per sample: conv w, conv b, linear w, linear b)
    shapes = [(tuple(conv.weight.shape), tuple(conv.bias.shape)), (tuple(lin.weight.shape), tuple(lin.bias.shape))]
    eps = seeded.eps_like_reference(int(g["eps_seed"]), shapes, samples=2)
    n0 = env["lib"].bnn_launch_count()
    with torch.no_grad():
        pre = net.layers[:7](x)                                    # deterministic prefix (stock torch layers, eval mode)
        for s in range(2):
            conv.weight.sample_with_eps(eps[s][0][0].to(dev)); conv.bias.sample_with_eps(eps[s][0][1].to(dev))
            lin.weight.sample_with_eps(eps[s][1][0].to(dev)); lin.bias.sample_with_eps(eps[s][1][1].to(dev))
            h = torch.nn.functional.elu(conv(pre, sample=False)).flatten(1)
            y = torch.softmax(lin(h, sample=False), -1)
            assert allclose(N(y), g["y%d" % s]), np.abs(N(y) - g["y%d" % s]).max()
    assert env["lib"].bnn_launch_count() > n0
    # ---- mc_batched: the prefix runs once, every Bayesian layer launches both samples; checked against the oracle on the
    # layers' Philox draw keys (fp32 parity mode)
    net.mc_batched = True
    env["bnn"].manual_seed(5)
    with torch.no_grad():
        ys = net(x)
    assert isinstance(ys, list) and len(ys) == 2
    pre_np = N(pre)
    for s in range(2):
        kw, kb = conv.weight.draw_key, conv.bias.draw_key
        w = orc.sample_affine(N(conv.weight.mean), N(conv.weight.scale), orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, 0, tuple(conv.weight.shape), kw.gen))
        b = orc.sample_affine(N(conv.bias.mean), N(conv.bias.scale), orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, 0, tuple(conv.bias.shape), kb.gen))
        h = orc.conv2d(pre_np, w, b, stride=(2, 2), padding=(1, 1))
        h = np.where(h > 0, h, np.expm1(h)).reshape(h.shape[0], -1).astype(np.float32)          # ELU
        kw, kb = lin.weight.draw_key, lin.bias.draw_key
        w = orc.sample_affine(N(lin.weight.mean), N(lin.weight.scale), orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, 0, tuple(lin.weight.shape), kw.gen))
        b = orc.sample_affine(N(lin.bias.mean), N(lin.bias.scale), orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, 0, tuple(lin.bias.shape), kb.gen))
        z = orc.linear(h, w, b).astype(np.float64)
        z = np.exp(z - z.max(-1, keepdims=True))
        want = (z / z.sum(-1, keepdims=True)).astype(np.float32)
        assert allclose(N(ys[s]), want), np.abs(N(ys[s]) - want).max()
    assert not torch.equal(ys[0], ys[1])


def test_tolerance_bookkeeping_k784_goldens(env):
    """the K >= 784 goldens are judged at 1e-5 OF THE OUTPUT SCALE (conftest.allclose_scaled) because
    the reference's own fp32 sgemm is further from exact arithmetic than 1e-5 absolute there.  With numbers: against the
    float64 evaluation of the same network on the same fp32 weights,
        (i) max |HIP fp32 mode - f64|, (ii) max |reference fp32 - f64| (stored by make_golden.py).
    Measured on MI355X (round 3): (i) = 5.4e-5 / 6.3e-5, (ii) = 6.6e-5 / 7.2e-5 at output rms 36 -- both ~2e-6 OF THE OUTPUT
    SCALE: three layers of bf16x3 contractions (six of nine partial products, the five small ones summed apart from the
    large one) against MKL's sgemm.  The 1e-5 absolute bar would reject both; the scaled bar (3.6e-4 here) holds both with
    margin.  Asserted: (i) inside the scaled tolerance with 2x margin, (i) <= (ii), (i) <= 1e-4."""
    g = load_golden("mlp_784_1200_1200_10")
    g64 = load_golden("mlp_784_1200_1200_10_f64")
    dev = env["dev"]
    from bayesianneuralnetworks_amd.nn import NormalLinear
    post = seeded.mlp_posteriors((784, 1200, 1200, 10), seed=int(g["param_seed"]))
    layers = []
    for mw, rw, mb, rb in post:
        L = NormalLinear(mw.shape[1], mw.shape[0])
        with torch.no_grad():
            L.weight.mean.copy_(mw); L.weight.scale.copy_(rw); L.bias.mean.copy_(mb); L.bias.scale.copy_(rb)
        layers.append(L.to(dev))
    x = seeded.mlp_input(512, 784, seed=int(g["x_seed"])).to(dev)
    shapes = [(tuple(p[0].shape), tuple(p[2].shape)) for p in post]
    eps = seeded.eps_like_reference(int(g["eps_seed"]), shapes, samples=2)
    report = []
    for s in range(2):
        h = x
        with torch.no_grad():
            for li, L in enumerate(layers):
                L.weight.sample_with_eps(eps[s][li][0].to(dev)); L.bias.sample_with_eps(eps[s][li][1].to(dev))
                h = L(h, sample=False)
                if li < 2:
                    h = torch.relu(h)
        ours = float(np.abs(N(h).astype(np.float64) - g64["y%d_f64" % s]).max())
        ref = float(g64["y%d_ref_err" % s])
        ref_vs_golden = float(np.abs(g["y%d" % s].astype(np.float64) - g64["y%d_f64" % s]).max())
        report.append((ours, ref, ref_vs_golden))
        print("sample %d: max|HIP - f64| = %.3e   max|reference fp32 - f64| = %.3e (golden y: %.3e)   output rms %.1f"
              % (s, ours, ref, ref_vs_golden, float(np.sqrt((g64["y%d_f64" % s] ** 2).mean()))))
        rms = float(np.sqrt((g64["y%d_f64" % s] ** 2).mean()))
        assert ours <= 0.5 * 1e-5 * rms, (ours, rms)    # half the scaled tolerance the goldens are judged by
        # round 3: NO FARTHER from exact arithmetic than the reference's own fp32 sgemm (round 2 asserted 4 x: the small
        # partial products of the bf16x3 contraction were accumulated into the large sum block by block; they now have an
        # accumulator of their own / a sweep over K of their own -- 1.36e-4 / 1.66e-4 became 5.4e-5 / 6.3e-5 against 6.6e-5 / 7.2e-5)
        assert ours <= ref, (ours, ref)
        assert ours <= 1.0e-4, ours
        assert abs(ref - ref_vs_golden) <= 1e-6         # the stored reference error is the golden's own distance
    import json, os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "tolerance_bookkeeping.json"), "w") as f:
            json.dump([{"hip_vs_f64": a, "reference_vs_f64": b} for a, b, _ in report], f)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_fashion_mnist_flipout_net_end_to_end(env, mode):
    """The FashionMNIST example's network (examples/FashionMNIST/model.py:20-33: stock conv prefix, FlipOutNormalConv2d(64, 64, 3, p1,
    s2), FlipoutNormalLinear(576, 10), Softmax) on the device: the Bayesian tail -- one fused Flipout conv launch in either mode, one
    contraction for the Flipout linear -- against float64 of conv.py:207-221 / dense.py:70-83 on the layers' own sign tensors,
    fed the device's own prefix output.  (No reference fixture holds this net's outputs: parity is pinned through the layer-level
    Flipout goldens, tests/golden/flipout_*.npz, and this float64 restatement.)"""
    from torch.nn import Conv2d, BatchNorm2d, ELU, Softmax, Flatten, Sequential
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import BayesianNetworkModule, FlipOutNormalConv2d, FlipoutNormalLinear
    dev = env["dev"]

    class BCNN(BayesianNetworkModule):
        def __init__(self):
            super().__init__(1, 10, 3)
            self.layers = Sequential(Conv2d(1, 32, 5, padding=2, stride=2), BatchNorm2d(32), ELU(), Conv2d(32, 32, 3, padding=1, stride=1), ELU(),
                                     Conv2d(32, 64, 3, padding=0, stride=2), ELU(), FlipOutNormalConv2d(64, 64, 3, padding=1, stride=2), ELU(),
                                     Flatten(), FlipoutNormalLinear(576, 10), Softmax(dim=-1))

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(17)
    net = BCNN().to(dev).eval()
    x = torch.randn(48, 1, 28, 28, generator=torch.Generator().manual_seed(2)).to(dev)
    conv, lin = net.layers[7], net.layers[10]
    bnn.set_compute(mode)
    try:
        n0 = env["lib"].bnn_launch_count()
        with torch.no_grad():
            ys = net(x)                                             # 3 stochastic forwards: fresh signs each
            pre = net.layers[:7](x)
            y = net._forward(x)                                     # one more, whose signs the layers still hold
        assert isinstance(ys, list) and len(ys) == 3 and not torch.equal(ys[0], ys[1])
        assert env["lib"].bnn_launch_count() > n0
        F = torch.nn.functional
        p64 = pre.double().cpu()
        R, S = (t.double().cpu() for t in conv.sampled)
        m, sd = conv.weight.mean.detach().double().cpu(), conv.weight.stddev.detach().double().cpu()
        h = F.conv2d(p64, m, None, 2, 1) + F.conv2d(p64 * S.expand_as(p64), sd, None, 2, 1) * R
        h = F.elu(h).flatten(1)
        Rl, Sl = (t.double().cpu() for t in lin.sampled)
        ml, sl = lin.weight.mean.detach().double().cpu(), lin.weight.stddev.detach().double().cpu()
        z = h @ ml.t() + ((h * Sl) @ sl.t()) * Rl
        want = torch.softmax(z, -1).numpy()
        tol = 2e-5 if mode == "f32" else 2e-2
        assert np.abs(N(y) - want).max() <= tol, (mode, np.abs(N(y) - want).max())
    finally:
        bnn.set_compute("f32")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_cifar10_example_net_end_to_end(env, mode):
    """The CIFAR10 example's network (examples/CIFAR10/model.py:20-38: five stock convolutions, NormalConv2d(128, 128, 3, p1), Linear,
    MultivariateNormalLinear(128, 10), Softmax) in one MC-batched pass on the device: the prefix runs once, the Bayesian conv launches
    all samples (draw + implicit GEMM) and is checked against float64 conv2d on the K1 draw of its recorded keys; the tail is torch's
    (the MVN head's own parity: tests/golden/mvn_linear_128x10.npz)."""
    from torch.nn import Linear, Conv2d, BatchNorm2d, ELU, Softmax, Flatten, Sequential
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import BayesianNetworkModule, NormalConv2d, MultivariateNormalLinear
    dev, ops = env["dev"], env["ops"]
    S, B = 4, 16

    class BCNN(BayesianNetworkModule):
        def __init__(self):
            super().__init__(3, 10, S)
            self.layers = Sequential(Conv2d(3, 64, 5, padding=2, stride=2), BatchNorm2d(64), ELU(), Conv2d(64, 128, 5, padding=2, stride=2), ELU(),
                                     Conv2d(128, 128, 5, padding=2, stride=2), ELU(), Conv2d(128, 128, 3, padding=1), ELU(),
                                     Conv2d(128, 128, 3, padding=1), ELU(), NormalConv2d(128, 128, 3, padding=1), ELU(), Flatten(),
                                     Linear(2048, 128), ELU(), MultivariateNormalLinear(128, 10), Softmax(dim=-1))

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(23)
    net = BCNN().to(dev).eval()
    net.mc_batched = True
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(dev)
    conv = net.layers[11]
    seen = {}
    hk = conv.register_forward_hook(lambda m, i, o: seen.update(x=i[0].detach(), y=o.detach()))
    bnn.set_compute(mode)
    try:
        bnn.manual_seed(11)
        n0 = env["lib"].bnn_launch_count()
        with torch.no_grad():
            ys = net(x)
        assert env["lib"].bnn_launch_count() == n0 + 2              # draw + implicit GEMM for all S samples; everything else is torch's
    finally:
        bnn.set_compute("f32")
        hk.remove()
    assert isinstance(ys, list) and len(ys) == S and all(t.shape == (B, 10) for t in ys)
    assert all(torch.allclose(t.sum(-1), torch.ones(B, device=dev), atol=1e-5) for t in ys) and not torch.equal(ys[0], ys[1])
    xin, yout = seen["x"], seen["y"]
    assert xin.shape == (B, 128, 4, 4) and yout.shape == (S * B, 128, 4, 4)      # the prefix ran ONCE, the Bayesian layer on all samples
    kw, kb = conv.weight.draw_key, conv.bias.draw_key
    w = ops._sample_affine_philox_raw(conv.weight.mean.detach(), conv.weight.scale.detach(), kw).double().cpu()
    b = ops._sample_affine_philox_raw(conv.bias.mean.detach(), conv.bias.scale.detach(), kb).double().cpu()
    x64 = xin.double().cpu()
    tol = 1e-5 if mode == "f32" else 2e-2
    for s in range(S):
        want = torch.nn.functional.conv2d(x64, w[s], b[s], 1, 1).numpy()
        got = N(yout[s * B:(s + 1) * B])
        scale = max(1.0, float(np.sqrt((want ** 2).mean())))
        assert np.abs(got - want).max() <= tol * scale + tol * np.abs(want).max(), (mode, s, np.abs(got - want).max())
