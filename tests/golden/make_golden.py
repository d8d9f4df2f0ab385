#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports pytorch_bayesian 0.0.4 from /root/reference (read-only), drives its
own layers / loss on seeded inputs and writes inputs + expected outputs as small
.npz files (numpy, no pickle).  Nothing of the reference's source is stored --
only numbers.  Shipped checkpoints are read with torch.load(weights_only=True).

eps capture: the reference draws eps with torch.randn_like from the global CPU
generator, weight first then bias, layer by layer (core.py:45, dense.py:46-54).
After torch.manual_seed(k) the same draws are regenerated here with
torch.randn(shape) in that order; the script asserts bit-equality of the
reference's `.sampled` with mean + stddev * eps before writing anything.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("BNN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import pytorch_bayesian  # noqa: E402
from pytorch_bayesian.nn import (NormalLinear, NormalConv2d, KLDivergence,  # noqa: E402
                                 BayesianNetworkModule, WeightNormal, FlipoutNormalLinear,
                                 FlipOutNormalConv2d)
from torch.distributions import Normal  # noqa: E402

assert pytorch_bayesian.__version__ == "0.0.4"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import seeded  # noqa: E402
torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def seeded_params(layer, gen):
    """Overwrite a layer's posterior with tests/golden/seeded.py's draws."""
    mu_w, rho_w, mu_b, rho_b = seeded.posterior(gen, tuple(layer.weight.mean.shape),
                                                layer.bias is not None)
    with torch.no_grad():
        layer.weight.mean.copy_(mu_w)
        layer.weight.scale.copy_(rho_w)
        if layer.bias is not None:
            layer.bias.mean.copy_(mu_b)
            layer.bias.scale.copy_(rho_b)


def draw_eps(layer, seed):
    """Regenerate the eps a forward after torch.manual_seed(seed) consumes."""
    torch.manual_seed(seed)
    ew = torch.randn(layer.weight.mean.shape)
    eb = torch.randn(layer.bias.mean.shape) if layer.bias is not None else None
    return ew, eb


def kl_parts(model_or_layers):
    """Per-tensor KL means in the reference's traversal order (loss.py:16-28)."""
    from torch.distributions.kl import kl_divergence
    parts = []
    for layer in model_or_layers:
        parts.append(kl_divergence(layer.weight.dist, layer.weight_prior).mean())
        if layer.bias is not None:
            parts.append(kl_divergence(layer.bias.dist, layer.bias_prior).mean())
    return parts


class Net(BayesianNetworkModule):
    def __init__(self, layers, samples=1):
        super().__init__(1, 1, samples)
        self.layers = torch.nn.Sequential(*layers)

    def _forward(self, x):
        return self.layers(x)


def linear_case(name, i, o, bias, seed, batch, prior=None, rho_override=None):
    gen = torch.Generator().manual_seed(seed)
    layer = NormalLinear(i, o, bias) if prior is None else NormalLinear(i, o, bias, prior)
    seeded_params(layer, gen)
    if rho_override is not None:
        with torch.no_grad():
            flat = layer.weight.scale.view(-1)
            for j, v in enumerate(rho_override):
                flat[j % flat.numel()] = v
    x = torch.randn(batch, i, generator=gen)
    gy = torch.randn(batch, o, generator=gen)
    ew, eb = draw_eps(layer, seed + 1000)
    torch.manual_seed(seed + 1000)
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    w, b = layer.sampled
    assert torch.equal(w, layer.weight.mean + layer.weight.stddev * ew)
    if bias:
        assert torch.equal(b, layer.bias.mean + layer.bias.stddev * eb)
    kld = KLDivergence(number_of_batches=3)(Net([layer]))
    parts = kl_parts([layer])
    loss = (y * gy).sum() + kld
    loss.backward()
    d = dict(mu_w=npf(layer.weight.mean), rho_w=npf(layer.weight.scale), eps_w=npf(ew),
             x=npf(x), gy=npf(gy), w=npf(w), y=npf(y), sigma_w=npf(layer.weight.stddev),
             kl_parts=np.array([p.item() for p in parts], dtype=np.float32),
             kl=np.float32(kld.item()), n_batches=np.float32(3),
             prior_mu=np.float32(layer.weight_prior.mean), prior_sigma=np.float32(layer.weight_prior.stddev),
             g_mu_w=npf(layer.weight.mean.grad), g_rho_w=npf(layer.weight.scale.grad),
             g_x=npf(xr.grad), eps_seed=np.int64(seed + 1000))
    if bias:
        d.update(mu_b=npf(layer.bias.mean), rho_b=npf(layer.bias.scale), eps_b=npf(eb), b=npf(b),
                 g_mu_b=npf(layer.bias.mean.grad), g_rho_b=npf(layer.bias.scale.grad))
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "y", tuple(y.shape), "kl", kld.item())


def conv_case(name, cin, cout, k, stride, pad, dil, groups, bias, seed, batch, hw, state=None,
              store_params=True):
    gen = torch.Generator().manual_seed(seed)
    layer = NormalConv2d(cin, cout, k, stride, pad, dil, groups, bias)
    if state is None:
        seeded_params(layer, gen)
    else:
        with torch.no_grad():
            layer.weight.mean.copy_(state["weight.mean"])
            layer.weight.scale.copy_(state["weight.scale"])
            layer.bias.mean.copy_(state["bias.mean"])
            layer.bias.scale.copy_(state["bias.scale"])
    x = torch.randn(batch, cin, hw, hw, generator=gen)
    ew, eb = draw_eps(layer, seed + 1000)
    torch.manual_seed(seed + 1000)
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    gy = torch.randn(y.shape, generator=gen)
    w, b = layer.sampled
    assert torch.equal(w, layer.weight.mean + layer.weight.stddev * ew)
    kld = KLDivergence()(Net([layer]))
    parts = kl_parts([layer])
    ((y * gy).sum() + kld).backward()
    d = dict(mu_w=npf(layer.weight.mean), rho_w=npf(layer.weight.scale), eps_w=npf(ew),
             x=npf(x), gy=npf(gy), w=npf(w), y=npf(y),
             conv=np.array([layer.stride[0], layer.stride[1], layer.padding[0], layer.padding[1],
                            layer.dilation[0], layer.dilation[1], groups], dtype=np.int64),
             kl_parts=np.array([p.item() for p in parts], dtype=np.float32),
             kl=np.float32(kld.item()), n_batches=np.float32(1),
             prior_mu=np.float32(0.0), prior_sigma=np.float32(0.1),
             g_mu_w=npf(layer.weight.mean.grad), g_rho_w=npf(layer.weight.scale.grad),
             g_x=npf(xr.grad), eps_seed=np.int64(seed + 1000))
    if bias:
        d.update(mu_b=npf(layer.bias.mean), rho_b=npf(layer.bias.scale), eps_b=npf(eb), b=npf(b),
                 g_mu_b=npf(layer.bias.mean.grad), g_rho_b=npf(layer.bias.scale.grad))
    if not store_params:
        # Large case: params / x / gy / eps are regenerated by the tests from the
        # seeds (seeded.posterior with Generator(seed), then x, then -- after the
        # forward -- gy from the same generator; eps = manual_seed(seed+1000)).
        for k_ in ("mu_w", "rho_w", "eps_w", "w", "mu_b", "rho_b", "eps_b", "b", "x", "gy",
                   "g_mu_w"):
            d.pop(k_, None)
        d.update(seed=np.int64(seed), shape=np.array([cin, cout, k, batch, hw], dtype=np.int64))
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "y", tuple(y.shape), "kl", kld.item())


def mlp_params(gen, dims):
    layers = []
    for i, o in zip(dims[:-1], dims[1:]):
        L = NormalLinear(i, o)
        seeded_params(L, gen)
        layers.append(L)
    return layers


def north_star_mlp():
    """784-1200-1200-10 NormalLinear MLP, batch 512, 2 MC samples.

    Parameters and x are NOT stored: tests regenerate them from the documented
    seeds with torch's CPU generator (which ships with PyTorch-ROCm on the GPU
    box).  Params: torch.Generator seed 0, per layer in order
    mu_w = (rand*2-1)/sqrt(in), rho_w = randn*0.15-2, mu_b likewise, rho_b;
    x: Generator seed 1, randn(512, 784); eps: torch.manual_seed(2), then per
    MC sample, per layer, randn(w.shape), randn(b.shape)."""
    layers = []
    for (mu_w, rho_w, mu_b, rho_b) in seeded.mlp_posteriors((784, 1200, 1200, 10), seed=0):
        L = NormalLinear(mu_w.shape[1], mu_w.shape[0])
        with torch.no_grad():
            L.weight.mean.copy_(mu_w)
            L.weight.scale.copy_(rho_w)
            L.bias.mean.copy_(mu_b)
            L.bias.scale.copy_(rho_b)
        layers.append(L)
    seq = []
    for j, L in enumerate(layers):
        seq.append(L)
        if j < len(layers) - 1:
            seq.append(torch.nn.ReLU())
    net = Net(seq, samples=2)
    x = seeded.mlp_input(512, 784, seed=1)
    torch.manual_seed(2)
    with torch.no_grad():
        ys = net(x)
    kld = KLDivergence()(net)
    parts = kl_parts(layers)
    np.savez(os.path.join(OUT, "mlp_784_1200_1200_10.npz"),
             y0=npf(ys[0]), y1=npf(ys[1]), kl=np.float32(kld.item()),
             kl_parts=np.array([p.item() for p in parts], dtype=np.float32),
             pred_mean=npf(torch.stack(ys).mean(0)),
             param_seed=np.int64(0), x_seed=np.int64(1), eps_seed=np.int64(2))
    print("mlp kl", kld.item(), [p.item() for p in parts])


def pretrained_mnist():
    path = os.path.join(REF, "examples/MNIST/mnist_pretrained.pth")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    conv = {k.split("layers.7.")[1]: v.float() for k, v in sd.items() if k.startswith("layers.7.")}
    lin = {k.split("layers.10.")[1]: v.float() for k, v in sd.items() if k.startswith("layers.10.")}
    # (64,64,3,3) s2 p1 conv on (B,64,6,6): the MNIST/FMNIST implicit-GEMM shape.
    conv_case("conv_mnist_pretrained", 64, 64, 3, 2, 1, 1, 1, True, 77, 4, 6, state=conv)
    # (10,576) head with trained tensors.
    layer = NormalLinear(576, 10)
    with torch.no_grad():
        layer.weight.mean.copy_(lin["weight.mean"])
        layer.weight.scale.copy_(lin["weight.scale"])
        layer.bias.mean.copy_(lin["bias.mean"])
        layer.bias.scale.copy_(lin["bias.scale"])
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(8, 576, generator=gen)
    ew, eb = draw_eps(layer, 1078)
    torch.manual_seed(1078)
    y = layer(x)
    w, b = layer.sampled
    assert torch.equal(w, layer.weight.mean + layer.weight.stddev * ew)
    # KLDivergence of the whole shipped MNIST net (2 Bayesian layers, 4 tensors).
    sys.path.insert(0, os.path.join(REF, "examples/MNIST"))
    import model as mnist_model
    net = mnist_model.BCNN(1, 10)
    net.load_state_dict(sd)
    kld = KLDivergence()(net)
    parts = kl_parts([net.layers[7], net.layers[10]])
    np.savez(os.path.join(OUT, "linear_mnist_pretrained.npz"),
             mu_w=npf(layer.weight.mean), rho_w=npf(layer.weight.scale), eps_w=npf(ew),
             mu_b=npf(layer.bias.mean), rho_b=npf(layer.bias.scale), eps_b=npf(eb),
             x=npf(x), w=npf(w), b=npf(b), y=npf(y),
             kl_net=np.float32(kld.item()),
             kl_parts_net=np.array([p.item() for p in parts], dtype=np.float32),
             prior_mu=np.float32(0.0), prior_sigma=np.float32(0.1))
    print("mnist pretrained kl", kld.item())


def weightnormal_case():
    """WeightNormal alone (core.py:7-45) incl. softplus corner values: the
    threshold (20), pruned entries (-30, prune.py:17), collapsed (-100)."""
    wn = WeightNormal(5, 6, 7)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        wn.mean.copy_(torch.randn(5, 6, 7, generator=gen))
        wn.scale.copy_(torch.randn(5, 6, 7, generator=gen) * 3 - 2)
        corner = torch.tensor([-100., -30., -20., -16., -12., -8., -2., 0., 5., 19.99, 20., 20.01, 25., 50.])
        wn.scale.view(-1)[:corner.numel()] = corner
    torch.manual_seed(6)
    eps = torch.randn(5, 6, 7)
    torch.manual_seed(6)
    wn.sample()
    assert torch.equal(wn.sampled, wn.mean + wn.stddev * eps)
    from torch.distributions.kl import kl_divergence
    kl_elem = kl_divergence(wn.dist, Normal(0.3, 0.25))
    np.savez(os.path.join(OUT, "weightnormal_5x6x7.npz"),
             mu=npf(wn.mean), rho=npf(wn.scale), eps=npf(eps), w=npf(wn.sampled),
             sigma=npf(wn.stddev), kl_elem=npf(kl_elem), kl_mean=np.float32(kl_elem.mean().item()),
             prior_mu=np.float32(0.3), prior_sigma=np.float32(0.25))
    print("weightnormal kl", kl_elem.mean().item())


def flipout_cases():
    """dense.py:63-83 and conv.py:145-227: random sign tensors R, S captured from the layer after its own
    forward (torch.rand on the CPU generator), outputs and autograd gradients."""
    for name, make, xshape in (
            ("flipout_linear_12x7", lambda: FlipoutNormalLinear(12, 7), (5, 12)),
            ("flipout_linear_64x48", lambda: FlipoutNormalLinear(64, 48), (33, 64)),
            ("flipout_conv_3_4_k3_p1", lambda: FlipOutNormalConv2d(3, 4, 3, padding=1), (2, 3, 10, 10)),
            # (groups > 1 raises inside the reference itself: S is sized C / groups, conv.py:157-161)
            ("flipout_conv_4_6_k3_s2", lambda: FlipOutNormalConv2d(4, 6, 3, stride=2, padding=1), (3, 4, 9, 9))):
        gen = torch.Generator().manual_seed(31)
        layer = make()
        seeded_params(layer, gen)
        x = torch.randn(*xshape, generator=gen)
        torch.manual_seed(2031)
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        R, S = layer.sampled
        gy = torch.randn(*y.shape, generator=gen)
        (y * gy).sum().backward()
        np.savez(os.path.join(OUT, name + ".npz"), mu_w=npf(layer.weight.mean), rho_w=npf(layer.weight.scale),
                 x=npf(x), gy=npf(gy), R=npf(R), S=npf(S), y=npf(y), g_mu_w=npf(layer.weight.mean.grad),
                 g_rho_w=npf(layer.weight.scale.grad), g_x=npf(xr.grad))
        print(name, "y", tuple(y.shape))


def mvn_case():
    """MultivariateNormalLinear(128, 10), the CIFAR10 example's head (examples/CIFAR10/model.py:37):
    forward with the reference's own noise (torch.rand_like -- UNIFORM, core.py:89-92 -- weight then bias), the
    multivariate KL through KLDivergence (loss.py:24-28), and autograd gradients of both.  The noise tensors are stored
    (they are the layer's input as much as x is); the element-wise sqrt of the lower-triangular factor (core.py:69) is
    part of what the fixture pins."""
    from pytorch_bayesian.nn import MultivariateNormalLinear
    torch.manual_seed(41)
    layer = MultivariateNormalLinear(128, 10)
    gen = torch.Generator().manual_seed(42)
    x = torch.randn(16, 128, generator=gen)
    gy = torch.randn(16, 10, generator=gen)
    torch.manual_seed(43)
    uw = torch.rand(10, 128)
    ub = torch.rand(10)
    torch.manual_seed(43)
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    w, b = layer.sampled
    assert torch.equal(w, layer.weight.mean + torch.matmul(layer.weight.stddev, uw.unsqueeze(-1)).squeeze(-1))
    assert torch.equal(b, layer.bias.mean + torch.matmul(layer.bias.stddev, ub.unsqueeze(-1)).squeeze(-1))
    kld = KLDivergence(number_of_batches=5)(Net([layer]))
    ((y * gy).sum() + kld).backward()
    np.savez_compressed(os.path.join(OUT, "mvn_linear_128x10.npz"),
             mu_w=npf(layer.weight.mean), scale_w=npf(layer.weight.scale), mu_b=npf(layer.bias.mean), scale_b=npf(layer.bias.scale),
             x=npf(x), gy=npf(gy), u_w=npf(uw), u_b=npf(ub), w=npf(w), b=npf(b), y=npf(y),
             kl=np.float32(kld.item()), n_batches=np.float32(5),
             g_mu_w=npf(layer.weight.mean.grad), g_scale_w=npf(layer.weight.scale.grad),
             g_mu_b=npf(layer.bias.mean.grad), g_scale_b=npf(layer.bias.scale.grad), g_x=npf(xr.grad))
    print("mvn y", tuple(y.shape), "kl", kld.item())


def mnist_whole_net():
    """The shipped example net end to end (examples/MNIST/model.py:20-33: Conv/BatchNorm/ELU prefix, NormalConv2d,
    NormalLinear, Softmax) with its trained checkpoint, eval mode, 2 MC samples: the state_dict tensors (numbers only,
    read with weights_only=True), the eps the two forwards consume (torch.manual_seed, randn in the reference's order:
    per sample, conv weight, conv bias, linear weight, linear bias) and the two softmax outputs + KL."""
    path = os.path.join(REF, "examples/MNIST/mnist_pretrained.pth")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sys.path.insert(0, os.path.join(REF, "examples/MNIST"))
    import model as mnist_model
    net = mnist_model.BCNN(1, 10, samples=2)
    net.load_state_dict(sd)
    net.eval()
    x = torch.randn(6, 1, 28, 28, generator=torch.Generator().manual_seed(51))
    torch.manual_seed(52)
    with torch.no_grad():
        ys = net(x)
    kld = KLDivergence()(net)
    d = {"sd__" + k.replace(".", "__"): v.detach().cpu().numpy() for k, v in sd.items()}
    d.update(x=npf(x), y0=npf(ys[0]), y1=npf(ys[1]), kl=np.float32(kld.item()), eps_seed=np.int64(52), x_seed=np.int64(51))
    np.savez_compressed(os.path.join(OUT, "mnist_bcnn_pretrained.npz"), **d)
    print("mnist whole net", tuple(ys[0].shape), "kl", kld.item())


def f64_references():
    """Tolerance bookkeeping: float64 evaluations of the K >= 784 cases next to the reference's
    own fp32 outputs, so that the tests can state |HIP - f64| <= |reference - f64| with numbers.  Same seeds as
    north_star_mlp()."""
    posts = seeded.mlp_posteriors((784, 1200, 1200, 10), seed=0)
    x = seeded.mlp_input(512, 784, seed=1)
    shapes = [(tuple(p[0].shape), tuple(p[2].shape)) for p in posts]
    eps = seeded.eps_like_reference(2, shapes, samples=2)
    out = {}
    for s in range(2):
        h = x.double()
        h32 = x
        for li, (mw, rw, mb, rb) in enumerate(posts):
            ew, eb = eps[s][li]
            # the fp32 weights the reference forms (core.py:44-45), then the contraction in float64 / in the reference's fp32
            w = mw + (1e-10 + torch.nn.functional.softplus(rw)) * ew
            b = mb + (1e-10 + torch.nn.functional.softplus(rb)) * eb
            h = torch.nn.functional.linear(h, w.double(), b.double())
            h32 = torch.nn.functional.linear(h32, w, b)
            if li < 2:
                h = torch.relu(h)
                h32 = torch.relu(h32)
        out["y%d_f64" % s] = h.numpy()
        out["y%d_ref_err" % s] = np.float64((h32.double() - h).abs().max().item())
    np.savez(os.path.join(OUT, "mlp_784_1200_1200_10_f64.npz"), **out)
    print("f64 refs: reference fp32 max err", out["y0_ref_err"], out["y1_ref_err"])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "flipout":
        flipout_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        mvn_case()
        mnist_whole_net()
        f64_references()
        return
    weightnormal_case()
    linear_case("linear_4x3", 3, 4, True, 11, 5)
    linear_case("linear_7x11", 11, 7, True, 12, 6, prior=Normal(0.05, 0.5))
    linear_case("linear_7x11_nobias", 11, 7, False, 13, 6)
    linear_case("linear_64x48", 48, 64, True, 14, 33, rho_override=[-30., -100., 21., 19.5])
    linear_case("linear_1x1", 1, 1, True, 15, 1)
    # reference test shapes (tests/conftest.py:270-281): (1,1,k1) and (3,4,k3,pad1) on 10x10
    conv_case("conv_1_1_k1", 1, 1, 1, 1, 0, 1, 1, True, 21, 1, 10)
    conv_case("conv_3_4_k3_p1", 3, 4, 3, 1, 1, 1, 1, True, 22, 2, 10)
    conv_case("conv_4_6_k3_s2_d2_g2", 4, 6, 3, 2, 2, 2, 2, False, 23, 3, 9)
    conv_case("conv_128_128_k3_p1", 128, 128, 3, 1, 1, 1, 1, True, 24, 2, 4, store_params=False)
    pretrained_mnist()
    north_star_mlp()
    flipout_cases()
    mvn_case()
    mnist_whole_net()
    f64_references()


if __name__ == "__main__":
    main()
