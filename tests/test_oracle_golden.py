"""Pins the CPU oracle (oracle/bnn_oracle.c) against fixtures produced by the real
reference (tests/golden/make_golden.py).  CPU only.

Tolerance: atol = rtol = 1e-5, the reference suite's own bar
(/root/reference/tests/test_nn/test_dense.py:11-12), fp32.
"""
import numpy as np
import pytest
import torch

from conftest import allclose, allclose_scaled, load_golden
from oracle import oracle as orc
import seeded

LINEAR = ["linear_4x3", "linear_7x11", "linear_7x11_nobias", "linear_64x48", "linear_1x1"]
CONV = ["conv_1_1_k1", "conv_3_4_k3_p1", "conv_4_6_k3_s2_d2_g2", "conv_mnist_pretrained"]


def test_philox_published_kat():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(v) for v in orc.philox4x32_10([0, 0, 0, 0], [0, 0])] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                              [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_eps_stream_is_standard_normal():
    e = orc.eps_fill(0xC0FFEE, 3, 5, 7, 0, (1 << 20,))
    assert abs(e.mean()) < 5e-3 and abs(e.std() - 1) < 5e-3
    assert abs((e ** 3).mean()) < 2e-2 and abs((e ** 4).mean() - 3) < 5e-2
    assert np.isfinite(e).all()
    # streams / samples / epochs are distinct, and prefixes are stable
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 6, 7, 0, (64,)))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 4, 5, 7, 0, (64,)))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 5, 8, 0, (64,)))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 5, 7, 1, (64,)))
    assert np.array_equal(e[:61], orc.eps_fill(0xC0FFEE, 3, 5, 7, 0, (61,)))


def test_philox7_published_kat():
    # Random123 kat_vectors, philox4x32 7 rounds (the generator of BNN_GEN_PHILOX7_U16); 10 rounds through the same entry
    assert [hex(v) for v in orc.philox4x32_r([0, 0, 0, 0], [0, 0], 7)] == ['0x5f6fb709', '0xd893f64', '0x4f121f81', '0x4f730a48']
    assert [hex(v) for v in orc.philox4x32_r([0xffffffff] * 4, [0xffffffff] * 2, 7)] == \
        ['0x5207ddc2', '0x45165e59', '0x4d8ee751', '0x8c52f662']
    assert [hex(v) for v in orc.philox4x32_r([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 7)] == \
        ['0x4dfccaba', '0x190a87f0', '0xc47362ba', '0xb6b5242a']
    assert np.array_equal(orc.philox4x32_r([1, 2, 3, 4], [5, 6], 10), orc.philox4x32_10([1, 2, 3, 4], [5, 6]))


def test_eps_stream_u16_is_standard_normal():
    """BNN_GEN_PHILOX7_U16 (eight eps per Philox4x32-7 block from 16-bit uniforms -- the bf16 mode's stream): moments, a
    Kolmogorov-Smirnov test against the normal CDF, independence of neighbours, the truncation the 16-bit radius implies, and
    the addressing (streams / samples / epochs distinct, prefixes stable, not the default stream)."""
    from scipy import stats
    n = 1 << 21
    e = orc.eps_fill(0xC0FFEE, 3, 5, 7, 0, (n,), 1)
    assert np.isfinite(e).all()
    assert abs(e.mean()) < 3e-3 and abs(e.std() - 1) < 3e-3
    assert abs((e ** 3).mean()) < 1.5e-2 and abs((e ** 4).mean() - 3) < 4e-2
    assert np.abs(e).max() <= 4.86                                   # r(2^-17) = sqrt(34 ln 2) = 4.855
    d, pval = stats.kstest(e.astype(np.float64), "norm")
    assert d < 1.5e-3 and pval > 1e-3, (d, pval)
    # the two outputs of a Box-Muller pair, neighbouring pairs and neighbouring blocks are uncorrelated
    for lag in (1, 2, 8):
        c = float(np.mean(e[:-lag].astype(np.float64) * e[lag:]))
        assert abs(c) < 4e-3, (lag, c)
    c2 = float(np.mean((e[:-1].astype(np.float64) ** 2 - 1) * (e[1:].astype(np.float64) ** 2 - 1)))      # cos^2 / sin^2 of one pair share r
    assert abs(c2) < 2e-2, c2
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 6, 7, 0, (64,), 1))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 4, 5, 7, 0, (64,), 1))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 5, 8, 0, (64,), 1))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 5, 7, 1, (64,), 1))
    assert not np.array_equal(e[:64], orc.eps_fill(0xC0FFEE, 3, 5, 7, 0, (64,), 0))
    assert np.array_equal(e[:61], orc.eps_fill(0xC0FFEE, 3, 5, 7, 0, (61,), 1))


def test_weightnormal_sigma_sample_kl():
    g = load_golden("weightnormal_5x6x7")
    assert allclose(orc.sigma(g["rho"]), g["sigma"])
    # relative check too: sigma spans 1e-10 .. 50
    assert np.allclose(orc.sigma(g["rho"]), g["sigma"], rtol=1e-5, atol=0)
    assert allclose(orc.sample_affine(g["mu"], g["rho"], g["eps"]), g["w"])
    kl = orc.kl_sum(g["mu"], g["rho"], float(g["prior_mu"]), float(g["prior_sigma"])) / g["mu"].size
    assert abs(kl - float(g["kl_mean"])) <= 1e-5 * abs(float(g["kl_mean"]))


@pytest.mark.parametrize("name", LINEAR + ["linear_mnist_pretrained"])
def test_linear_forward(name):
    g = load_golden(name)
    w = orc.sample_affine(g["mu_w"], g["rho_w"], g["eps_w"])
    assert allclose(w, g["w"])
    b = None
    if "mu_b" in g:
        b = orc.sample_affine(g["mu_b"], g["rho_b"], g["eps_b"])
        assert allclose(b, g["b"])
    assert allclose(orc.linear(g["x"], w, b), g["y"])


@pytest.mark.parametrize("name", LINEAR)
def test_linear_kl_and_grads(name):
    g = load_golden(name)
    pm, ps = float(g["prior_mu"]), float(g["prior_sigma"])
    tensors = [(g["mu_w"], g["rho_w"], pm, ps)]
    if "mu_b" in g:
        tensors.append((g["mu_b"], g["rho_b"], pm, ps))
    for (mu, rho, _, _), want in zip(tensors, g["kl_parts"]):
        assert abs(orc.kl_sum(mu, rho, pm, ps) / mu.size - want) <= 1e-5 * (1 + abs(want))
    assert abs(orc.kl_divergence(tensors, float(g["n_batches"])) - float(g["kl"])) <= 1e-5 * (1 + abs(float(g["kl"])))
    # backward: loss = sum(y * gy) + KL  (make_golden.linear_case)
    gw = g["gy"].T.astype(np.float64) @ g["x"].astype(np.float64)
    gmu_s, grho_s = orc.sample_affine_bwd(gw.astype(np.float32), g["rho_w"], g["eps_w"])
    T = len(tensors)
    gmu_k, grho_k = orc.kl_bwd(g["mu_w"], g["rho_w"], pm, ps, 1.0 / (g["mu_w"].size * T * float(g["n_batches"])))
    assert np.allclose(gmu_s + gmu_k, g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(grho_s + grho_k, g["g_rho_w"], rtol=1e-4, atol=1e-5)
    w = orc.sample_affine(g["mu_w"], g["rho_w"], g["eps_w"])
    assert np.allclose(g["gy"].astype(np.float64) @ w.astype(np.float64), g["g_x"], rtol=1e-4, atol=1e-5)
    if "mu_b" in g:
        gb = g["gy"].sum(0)
        gmu_s, grho_s = orc.sample_affine_bwd(gb, g["rho_b"], g["eps_b"])
        gmu_k, grho_k = orc.kl_bwd(g["mu_b"], g["rho_b"], pm, ps, 1.0 / (g["mu_b"].size * T * float(g["n_batches"])))
        assert np.allclose(gmu_s + gmu_k, g["g_mu_b"], rtol=1e-4, atol=1e-5)
        assert np.allclose(grho_s + grho_k, g["g_rho_b"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", CONV)
def test_conv2d_forward_and_kl(name):
    g = load_golden(name)
    sh, sw, ph, pw, dh, dw, groups = [int(v) for v in g["conv"]]
    w = orc.sample_affine(g["mu_w"], g["rho_w"], g["eps_w"])
    assert allclose(w, g["w"])
    b = orc.sample_affine(g["mu_b"], g["rho_b"], g["eps_b"]) if "mu_b" in g else None
    y = orc.conv2d(g["x"], w, b, (sh, sw), (ph, pw), (dh, dw), groups)
    assert y.shape == g["y"].shape
    assert allclose(y, g["y"])
    tensors = [(g["mu_w"], g["rho_w"], 0.0, 0.1)]
    if "mu_b" in g:
        tensors.append((g["mu_b"], g["rho_b"], 0.0, 0.1))
    assert abs(orc.kl_divergence(tensors, 1.0) - float(g["kl"])) <= 1e-5 * (1 + abs(float(g["kl"])))


def test_conv_128_regenerated_from_seed():
    g = load_golden("conv_128_128_k3_p1")
    cin, cout, k, batch, hw = [int(v) for v in g["shape"]]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    mu_w, rho_w, mu_b, rho_b = seeded.posterior(gen, (cout, cin, k, k))
    x = torch.randn(batch, cin, hw, hw, generator=gen)
    (ew, eb), = seeded.eps_like_reference(int(g["eps_seed"]), [((cout, cin, k, k), (cout,))])[0]
    w = orc.sample_affine(mu_w.numpy(), rho_w.numpy(), ew.numpy())
    b = orc.sample_affine(mu_b.numpy(), rho_b.numpy(), eb.numpy())
    y = orc.conv2d(x.numpy(), w, b, (1, 1), (1, 1), (1, 1), 1)
    assert allclose(y, g["y"])
    kl = orc.kl_divergence([(mu_w.numpy(), rho_w.numpy(), 0.0, 0.1), (mu_b.numpy(), rho_b.numpy(), 0.0, 0.1)])
    assert abs(kl - float(g["kl"])) <= 1e-5 * (1 + abs(float(g["kl"])))


def test_mnist_pretrained_net_kl():
    """KLDivergence of the shipped MNIST net = 0.20435977."""
    g = load_golden("linear_mnist_pretrained")
    c = load_golden("conv_mnist_pretrained")
    tensors = [(c["mu_w"], c["rho_w"], 0.0, 0.1), (c["mu_b"], c["rho_b"], 0.0, 0.1),
               (g["mu_w"], g["rho_w"], 0.0, 0.1), (g["mu_b"], g["rho_b"], 0.0, 0.1)]
    for (mu, rho, pm, ps), want in zip(tensors, g["kl_parts_net"]):
        assert abs(orc.kl_sum(mu, rho, pm, ps) / mu.size - want) <= 1e-5 * (1 + abs(want))
    assert abs(orc.kl_divergence(tensors) - float(g["kl_net"])) <= 1e-5


def test_north_star_mlp_two_samples():
    """784-1200-1200-10, batch 512, 2 MC samples: oracle MC loop vs the reference."""
    g = load_golden("mlp_784_1200_1200_10")
    post = seeded.mlp_posteriors((784, 1200, 1200, 10), seed=int(g["param_seed"]))
    x = seeded.mlp_input(512, 784, seed=int(g["x_seed"])).numpy()
    shapes = [(tuple(p[0].shape), tuple(p[2].shape)) for p in post]
    eps = seeded.eps_like_reference(int(g["eps_seed"]), shapes, samples=2)
    layers = []
    for j, (mw, rw, mb, rb) in enumerate(post):
        layers.append(('linear', mw.numpy(), rw.numpy(), mb.numpy(), rb.numpy()))
        if j < len(post) - 1:
            layers.append(('relu',))
    eps_np = [[(ew.numpy(), eb.numpy()) for ew, eb in per] for per in eps]
    ys = orc.mc_forward(layers, x, eps_np)
    # output rms is 36 here: 1e-5 relative to the output scale (conftest.allclose_scaled)
    assert allclose_scaled(ys[0], g["y0"]) and allclose_scaled(ys[1], g["y1"])
    assert allclose_scaled((ys[0] + ys[1]) / 2, g["pred_mean"])
    tensors = []
    for mw, rw, mb, rb in post:
        tensors += [(mw.numpy(), rw.numpy(), 0.0, 0.1), (mb.numpy(), rb.numpy(), 0.0, 0.1)]
    assert abs(orc.kl_divergence(tensors) - float(g["kl"])) <= 1e-5 * (1 + float(g["kl"]))
    for (mu, rho, pm, ps), want in zip(tensors, g["kl_parts"]):
        assert abs(orc.kl_sum(mu, rho, pm, ps) / mu.size - want) <= 1e-5 * (1 + abs(want))


def test_reference_port_is_bit_identical_to_the_reference():
    """oracle/reference_port.py (bench.py's cpu_baseline) == the reference, same seed."""
    from oracle import reference_port as port
    g = load_golden("mlp_784_1200_1200_10")
    post = seeded.mlp_posteriors((784, 1200, 1200, 10), seed=int(g["param_seed"]))
    x = seeded.mlp_input(512, 784, seed=int(g["x_seed"]))
    torch.set_num_threads(1)
    torch.manual_seed(int(g["eps_seed"]))
    with torch.no_grad():
        ys = port.mlp_forward(x, post, 2)
        kl = port.kl_divergence_loss(post)
    assert allclose_scaled(ys[0].numpy(), g["y0"]) and allclose_scaled(ys[1].numpy(), g["y1"])
    assert abs(kl.item() - float(g["kl"])) < 1e-7
    gl = load_golden("linear_7x11")
    torch.manual_seed(int(gl["eps_seed"]))
    y = port.normal_linear(torch.from_numpy(gl["x"]), torch.from_numpy(gl["mu_w"]), torch.from_numpy(gl["rho_w"]),
                           torch.from_numpy(gl["mu_b"]), torch.from_numpy(gl["rho_b"]))
    assert np.array_equal(y.numpy(), gl["y"])


# ------------------------------------------------------------------ Flipout
@pytest.mark.parametrize("name", ["flipout_linear_12x7", "flipout_linear_64x48"])
def test_flipout_linear_is_sampled_affine_with_rank1_signs(name):
    """dense.py:78-83: x mu^T + ((x * S) sigma^T) * R == x (mu + sigma * outer(R, S))^T -- the identity the
    device path uses (eps = outer(R, S) into the K1 / K2 oracles)."""
    g = load_golden(name)
    w = orc.sample_affine(g["mu_w"], g["rho_w"], np.outer(g["R"], g["S"]).astype(np.float32))
    assert allclose(orc.linear(g["x"], w), g["y"])
    assert set(np.unique(g["R"])) <= {-1.0, 1.0} and set(np.unique(g["S"])) <= {-1.0, 1.0}


@pytest.mark.parametrize("name,stride,pad", [("flipout_conv_3_4_k3_p1", 1, 1), ("flipout_conv_4_6_k3_s2", 2, 1)])
def test_flipout_conv_two_contractions(name, stride, pad):
    """conv.py:213-227: conv(x, mu) + conv(x * S, sigma) * R with per-example signs."""
    g = load_golden(name)
    geo = dict(stride=(stride, stride), padding=(pad, pad))
    y = orc.conv2d(g["x"], g["mu_w"], None, **geo) + orc.conv2d(g["x"] * g["S"], orc.sigma(g["rho_w"]), None, **geo) * g["R"]
    assert allclose(y, g["y"])


@pytest.mark.parametrize("name", ["linear_4x3", "linear_7x11_nobias", "linear_64x48"])
def test_oracle_linear_bwd_matches_reference_autograd(name):
    """oracle.linear_bwd (the checker of the HIP backward kernels) against the reference's own autograd
    gradients of sum(y * gy) -- the KL part of the golden's loss is subtracted with oracle.kl_bwd."""
    g = load_golden(name)
    g_mu, g_rho, gx = orc.linear_bwd(g["mu_w"], g["rho_w"], g["x"][None], g["gy"][None], [g["eps_w"]])
    ntens = 2 if "mu_b" in g else 1
    km, kr = orc.kl_bwd(g["mu_w"], g["rho_w"], float(g["prior_mu"]), float(g["prior_sigma"]),
                        1.0 / (g["mu_w"].size * ntens * float(g["n_batches"])))
    assert np.allclose(g_mu + km, g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(g_rho + kr, g["g_rho_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(gx[0], g["g_x"], rtol=1e-4, atol=1e-5)


def test_oracle_prune_score_is_the_reference_log_prob():
    """prune/prune.py:11 with a tensor argument (the reference's int argument fails on current torch)."""
    g = load_golden("linear_64x48")
    mu, rho = torch.from_numpy(g["mu_w"]), torch.from_numpy(g["rho_w"])
    want = torch.distributions.Normal(mu, 1e-10 + torch.nn.functional.softplus(rho)).log_prob(torch.zeros(()))
    assert np.allclose(orc.prune_score(g["mu_w"], g["rho_w"]), want.numpy(), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- round 2: MVN head, shipped MNIST net (CPU side)
def _mvn_layer(g):
    import torch
    from bayesianneuralnetworks_amd.nn import MultivariateNormalLinear
    layer = MultivariateNormalLinear(128, 10)
    with torch.no_grad():
        layer.weight.mean.copy_(torch.from_numpy(g["mu_w"])); layer.weight.scale.copy_(torch.from_numpy(g["scale_w"]))
        layer.bias.mean.copy_(torch.from_numpy(g["mu_b"])); layer.bias.scale.copy_(torch.from_numpy(g["scale_b"]))
    return layer


def test_mvn_head_host_path_matches_reference_golden():
    """MultivariateNormalLinear(128, 10) on the CPU: forward on the reference's own uniform noise, the
    multivariate KL through KLDivergence and all gradients against the fixture the real reference produced."""
    import torch
    from bayesianneuralnetworks_amd.nn import KLDivergence, BayesianNetworkModule
    g = load_golden("mvn_linear_128x10")
    layer = _mvn_layer(g)
    layer.weight.sample_with_noise(torch.from_numpy(g["u_w"]))
    layer.bias.sample_with_noise(torch.from_numpy(g["u_b"]))
    layer.sampled = (layer.weight.sampled, layer.bias.sampled)
    assert allclose(layer.sampled[0].detach().numpy(), g["w"]) and allclose(layer.sampled[1].detach().numpy(), g["b"])
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = layer(x, sample=False)
    assert allclose(y.detach().numpy(), g["y"])

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(1, 1, 1)
            self.layers = torch.nn.Sequential(layer)

    kl = KLDivergence(number_of_batches=float(g["n_batches"]))(Net())
    assert abs(kl.item() - float(g["kl"])) <= 1e-5 * abs(float(g["kl"]))
    ((y * torch.from_numpy(g["gy"])).sum() + kl).backward()
    for got, want in ((layer.weight.mean.grad, "g_mu_w"), (layer.weight.scale.grad, "g_scale_w"),
                      (layer.bias.mean.grad, "g_mu_b"), (layer.bias.scale.grad, "g_scale_b"), (x.grad, "g_x")):
        assert allclose(got.numpy(), g[want], 2e-5), want
    # the reference's own sampling order under the global generator: weight noise, then bias noise
    torch.manual_seed(43)
    y2 = layer(torch.from_numpy(g["x"]))
    assert allclose(y2.detach().numpy(), g["y"])
