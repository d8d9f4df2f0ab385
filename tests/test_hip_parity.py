"""GPU parity tests: the HIP path (through the C-ABI, via the host modules) against
  (a) the golden fixtures produced by the real reference (eps supplied = parity mode), and
  (b) the CPU oracle running the same Philox draws (production mode).

Tolerance: atol = rtol = 1e-5, fp32 -- the reference suite's bar
(/root/reference/tests/test_nn/test_dense.py:11-12) and BASELINE.json's north_star.
Whole-network outputs with rms >> 1 use conftest.allclose_scaled (1e-5 of the output scale).
bf16 compute mode: stated per test.
"""
import numpy as np
import pytest
import torch

from conftest import allclose, allclose_scaled, assert_close_scaled, load_golden
import seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib, ops
    from oracle import oracle as orc
    lib = _lib.load()
    return dict(bnn=bnn, lib=lib, ops=ops, orc=orc, dev=torch.device("cuda:0"))


def T(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)


def N(t):
    return t.detach().cpu().numpy()


LINEAR = ["linear_4x3", "linear_7x11", "linear_7x11_nobias", "linear_64x48", "linear_1x1",
          "linear_mnist_pretrained"]
CONV = ["conv_1_1_k1", "conv_3_4_k3_p1", "conv_4_6_k3_s2_d2_g2", "conv_mnist_pretrained"]


# ------------------------------------------------------------------ library / RNG
def test_library_identity_and_counter(env):
    lib = env["lib"]
    assert lib.bnn_abi_version() == 2
    assert lib.bnn_arch() == b"gfx950"
    n0 = lib.bnn_launch_count()
    env["ops"].sigma(torch.zeros(8, device=env["dev"]))
    assert lib.bnn_launch_count() == n0 + 1


@pytest.mark.parametrize("gen", [0, 1])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 7, 8, 9, 1023, 4096, 100003])
def test_eps_stream_matches_cpu_twin(env, n, gen):
    from bayesianneuralnetworks_amd._rng import DrawKey
    key = DrawKey(0x1234567890ABCDEF, 777, 3, 4, 42, gen=gen)
    got = N(env["ops"].eps_philox((n,), key, env["dev"]))
    for s in range(4):
        want = env["orc"].eps_fill(key.seed, key.stream, key.sample0 + s, key.epoch_host, 0, (n,), key.gen)
        # eps itself: native sin/cos/sqrt on the GPU vs double on the CPU
        assert np.abs(got[s] - want).max() < 2e-5, np.abs(got[s] - want).max()


@pytest.mark.parametrize("gen", [0, 1])
def test_eps_stream_moments_and_independence(env, gen):
    from bayesianneuralnetworks_amd._rng import DrawKey
    e = N(env["ops"].eps_philox((1 << 20,), DrawKey(99, 5, 0, 2, 0, gen=gen), env["dev"]))
    for s in range(2):
        assert abs(e[s].mean()) < 5e-3 and abs(e[s].std() - 1) < 5e-3
        assert abs((e[s] ** 4).mean() - 3) < 5e-2
    assert abs(np.corrcoef(e[0], e[1])[0, 1]) < 5e-3
    assert abs(np.corrcoef(e[0][:-1], e[0][1:])[0, 1]) < 5e-3


def test_eps_stream_is_standard_normal_ks(env):
    """Kolmogorov-Smirnov against N(0, 1) on 2^20 draws per stream / sample / epoch (p > 1e-3 each), and the
    tails exist (|eps| > 4 occurs about 2^20 * 6.3e-5 = 66 times)."""
    from scipy import stats
    from bayesianneuralnetworks_amd._rng import DrawKey
    for key in (DrawKey(1, 1, 0, 1, 0), DrawKey(2 ** 63 + 5, 65535, 65535, 1, 7), DrawKey(3, 9, 4, 1, 2 ** 31)):
        e = N(env["ops"].eps_philox((1 << 20,), key, env["dev"]))[0].astype(np.float64)
        assert stats.kstest(e, "norm").pvalue > 1e-3
        assert 25 < int((np.abs(e) > 4).sum()) < 130


def test_epoch_dev_changes_the_draw(env):
    from bayesianneuralnetworks_amd._rng import DrawKey, default_generator
    from bayesianneuralnetworks_amd import _lib
    key = DrawKey(7, 1, 0, 1, 0)
    a = N(env["ops"].eps_philox((64,), key, env["dev"]))
    cell = default_generator.epoch_dev(env["dev"])
    _lib.check(env["lib"].bnn_rng_advance(_lib.ptr(cell), 1, _lib.stream_ptr(env["dev"])), "advance")
    b = N(env["ops"].eps_philox((64,), key, env["dev"]))
    key_prev = DrawKey(7, 1, 0, 1, 0, epoch_dev_delta=-1)
    c = N(env["ops"].eps_philox((64,), key_prev, env["dev"]))
    cell.zero_()
    assert not np.array_equal(a, b)
    assert np.array_equal(a, c)
    want = env["orc"].eps_fill(7, 1, 0, 0, 1, (64,))
    assert np.abs(b[0] - want).max() < 2e-5


# ------------------------------------------------------------------ K1
def test_sigma_and_sample_affine_golden(env):
    g = load_golden("weightnormal_5x6x7")
    dev = env["dev"]
    sg = N(env["ops"].sigma(T(g["rho"], dev)))
    assert np.allclose(sg, g["sigma"], rtol=1e-5, atol=0)
    w = N(env["ops"].sample_affine_eps(T(g["mu"], dev), T(g["rho"], dev), T(g["eps"], dev)))
    assert allclose(w, g["w"])


@pytest.mark.parametrize("rgen", [0, 1])
@pytest.mark.parametrize("shape", [(1,), (3, 4), (5, 6, 7), (1200, 784), (10,), (64, 64, 3, 3)])
def test_sample_affine_philox_vs_oracle(env, shape, rgen):
    from bayesianneuralnetworks_amd._rng import DrawKey
    dev = env["dev"]
    gen = torch.Generator().manual_seed(3)
    mu = torch.randn(shape, generator=gen) * 0.1
    rho = torch.randn(shape, generator=gen) * 0.15 - 2.0
    key = DrawKey(2024, 11, 1, 3, 9, gen=rgen)
    got = N(env["ops"].sample_affine_philox(mu.to(dev), rho.to(dev), key))
    for s in range(3):
        eps = env["orc"].eps_fill(key.seed, key.stream, key.sample0 + s, key.epoch_host, 0, shape, key.gen)
        assert allclose(got[s], env["orc"].sample_affine(mu.numpy(), rho.numpy(), eps))


def test_weightnormal_reference_kat_on_gpu(env):
    """tests/test_nn/test_core.py:14-39 on the device: collapsed posterior samples to 0."""
    from bayesianneuralnetworks_amd.nn import WeightNormal
    for shape in [(1,), (3, 4), (5, 6, 7)]:
        wn = WeightNormal(*shape).to(env["dev"])
        torch.nn.init.constant_(wn.mean, 0)
        torch.nn.init.constant_(wn.scale, -100)
        wn.sample()
        assert wn.sampled.is_cuda and wn.sampled.shape == shape
        assert (wn.stddev > 0).all() and (wn.stddev ** 2 == wn.variance).all()
        assert allclose(N(wn.sampled), np.zeros(shape))


# ------------------------------------------------------------------ K3
@pytest.mark.parametrize("name", ["linear_4x3", "linear_7x11", "linear_7x11_nobias", "linear_64x48", "linear_1x1"])
def test_kl_golden_small(env, name):
    g = load_golden(name)
    dev = env["dev"]
    pm, ps = float(g["prior_mu"]), float(g["prior_sigma"])
    mus, rhos = [T(g["mu_w"], dev)], [T(g["rho_w"], dev)]
    if "mu_b" in g:
        mus.append(T(g["mu_b"], dev))
        rhos.append(T(g["rho_b"], dev))
    out = N(env["ops"].kl_normal(mus, rhos, [(pm, ps)] * len(mus), float(g["n_batches"])))
    for t, want in enumerate(g["kl_parts"]):
        assert abs(out[t] / mus[t].numel() - want) <= 1e-5 * (1 + abs(want))
    assert abs(out[-1] - float(g["kl"])) <= 1e-5 * (1 + abs(float(g["kl"])))


def test_kl_corner_values_golden(env):
    g = load_golden("weightnormal_5x6x7")
    dev = env["dev"]
    out = N(env["ops"].kl_normal([T(g["mu"], dev)], [T(g["rho"], dev)],
                                 [(float(g["prior_mu"]), float(g["prior_sigma"]))], 1.0))
    assert abs(out[1] - float(g["kl_mean"])) <= 1e-5 * abs(float(g["kl_mean"]))


def test_kl_mnist_pretrained_net(env):
    """0.20435977: KLDivergence of the shipped MNIST net."""
    g = load_golden("linear_mnist_pretrained")
    c = load_golden("conv_mnist_pretrained")
    dev = env["dev"]
    mus = [T(c["mu_w"], dev), T(c["mu_b"], dev), T(g["mu_w"], dev), T(g["mu_b"], dev)]
    rhos = [T(c["rho_w"], dev), T(c["rho_b"], dev), T(g["rho_w"], dev), T(g["rho_b"], dev)]
    out = N(env["ops"].kl_normal(mus, rhos, [(0.0, 0.1)] * 4, 1.0))
    for t, want in enumerate(g["kl_parts_net"]):
        assert abs(out[t] / mus[t].numel() - want) <= 1e-5 * (1 + abs(want))
    assert abs(out[4] - float(g["kl_net"])) <= 1e-5


def test_kl_is_bitwise_reproducible(env):
    dev = env["dev"]
    gen = torch.Generator().manual_seed(0)
    mu = (torch.randn(1200 * 1200, generator=gen) * 0.05).to(dev)
    rho = (torch.randn(1200 * 1200, generator=gen) * 0.15 - 2).to(dev)
    a = N(env["ops"].kl_normal([mu], [rho], [(0.0, 0.1)], 1.0))
    b = N(env["ops"].kl_normal([mu], [rho], [(0.0, 0.1)], 1.0))
    assert np.array_equal(a, b)


def test_kl_many_tensors_both_reduction_paths(env):
    """<= 64 tensors: one partial launch; more: several, sharing the workspace.  Both must match
    the oracle on ragged tensors, call after call."""
    dev = env["dev"]
    gen = torch.Generator().manual_seed(5)
    sizes = [1, 3, 2047, 2048, 2049, 5000, 7, 64] * 9            # 72 ragged tensors
    mus = [torch.randn(n, generator=gen) * 0.05 for n in sizes]
    rhos = [torch.randn(n, generator=gen) * 0.15 - 2 for n in sizes]
    dm, dr = [m.to(dev) for m in mus], [r.to(dev) for r in rhos]
    for T_ in (72, 64, 5, 72, 5):                                  # alternate the two paths
        out = N(env["ops"].kl_normal(dm[:T_], dr[:T_], [(0.0, 0.1)] * T_, 3.0))
        want = [env["orc"].kl_sum(mus[t].numpy(), rhos[t].numpy(), 0.0, 0.1) for t in range(T_)]
        for t in range(T_):
            assert abs(out[t] - want[t]) <= 1e-5 * (1 + abs(want[t])), (T_, t)
        scalar = np.mean([w / sizes[t] for t, w in enumerate(want)]) / 3.0
        assert abs(out[T_] - scalar) <= 1e-5 * (1 + abs(scalar))


@pytest.mark.parametrize("total", [(9 << 20) + 4099, (33 << 20) + 12345])
def test_kl_large_workgroup_paths(env, total):
    """>= 8 Mi scalars: 8192-scalar workgroups; >= 32 Mi: 32768-scalar workgroups walking 4 blocks each.  Ragged
    tensor ends (a last workgroup that stops after 1, 2, 3 blocks; a last block that is not full), an unaligned
    tensor (element-wise path), against the oracle's double sums."""
    dev = env["dev"]
    gen = torch.Generator().manual_seed(11)
    n_small = 8192 * 2 + 5                          # ends inside the first block of its last workgroup
    n_odd = 32768 + 8192 * 2 + 77                   # ends inside block 3 of its second workgroup
    sizes = [total - n_small - n_odd, n_small, n_odd]
    mus = [torch.randn(n, generator=gen) * 0.05 for n in sizes]
    rhos = [torch.randn(n, generator=gen) * 0.15 - 2 for n in sizes]
    dm, dr = [m.to(dev) for m in mus], [r.to(dev) for r in rhos]
    # third tensor as an UNALIGNED view (4-byte offset): the element-wise path inside the big-workgroup kernels
    pad_m, pad_r = torch.zeros(sizes[2] + 1, device=dev), torch.zeros(sizes[2] + 1, device=dev)
    pad_m[1:] = dm[2]; pad_r[1:] = dr[2]
    dm[2], dr[2] = pad_m[1:], pad_r[1:]
    out = N(env["ops"].kl_normal(dm, dr, [(0.0, 0.1)] * 3, 2.0))
    again = N(env["ops"].kl_normal(dm, dr, [(0.0, 0.1)] * 3, 2.0))
    assert np.array_equal(out, again)               # fixed summation order
    want = [env["orc"].kl_sum(mus[t].numpy(), rhos[t].numpy(), 0.0, 0.1) for t in range(3)]
    for t in range(3):
        assert abs(out[t] - want[t]) <= 1e-5 * (1 + abs(want[t])), (t, out[t], want[t])
    scalar = np.mean([w / sizes[t] for t, w in enumerate(want)]) / 2.0
    assert abs(out[3] - scalar) <= 1e-5 * (1 + abs(scalar))


@pytest.fixture
def restore_draw_once(env):
    yield
    env["ops"].DRAW_ONCE_BF16 = True


def test_kl_second_pass_inside_mc_reduction(env, restore_draw_once):
    """ops.kl_normal_begin + ops.mc_mean(kl=...) (bnn_kl_forward_partial / bnn_mc_sum_kl: KL's second pass as one extra
    workgroup of the MC reduction) == ops.kl_normal + ops.mc_mean, bit for bit; the epoch cell is bumped once."""
    dev = env["dev"]
    ops = env["ops"]
    gen = torch.Generator().manual_seed(3)
    sizes = [1200 * 784, 1200, 1200 * 1200, 1200, 12000, 10]
    mus = [(torch.randn(n, generator=gen) * 0.05).to(dev) for n in sizes]
    rhos = [(torch.randn(n, generator=gen) * 0.15 - 2).to(dev) for n in sizes]
    pri = [(0.0, 0.1)] * len(sizes)
    y = torch.randn(8, 512, 10, device=dev)
    cell = torch.zeros(4, dtype=torch.int32, device=dev)
    want_kl = N(ops.kl_normal(mus, rhos, pri, 2.0))
    want_mean = N(ops.mc_mean(y, scale=0.25))
    out = torch.full((len(sizes) + 1,), -1.0, device=dev)
    h = ops.kl_normal_begin(mus, rhos, pri, 2.0, out=out)
    got_mean = N(ops.mc_mean(y, scale=0.25, advance=cell, kl=h))
    assert np.array_equal(N(out), want_kl)
    assert np.array_equal(got_mean, want_mean)
    assert cell.tolist() == [1, 0, 0, 0]
    with pytest.raises(Exception):
        ops.mc_mean(y, kl=h)                        # a handle is finished once
    # carry=True: the first pass rides in a narrow layer's launch (bnn_linear_forward_sampled_kl); same bits, and the
    # layer's own output is unchanged.  (The FUSED kernels: in bf16 mode the default is now the draw-once path, whose
    # draw launch carries the KL instead -- tests/test_dense_path.py::test_draw_multi_many_layers_and_kl_carry.)
    from bayesianneuralnetworks_amd._rng import DrawKey
    from bayesianneuralnetworks_amd import _lib
    ops.DRAW_ONCE_BF16 = False
    x = torch.randn(8, 512, 1200, device=dev).to(torch.bfloat16)
    mw, rw, mb, rb = mus[4].view(10, 1200), rhos[4].view(10, 1200), mus[5], rhos[5]
    kw, kb = DrawKey(7, 1, 0, 8, 0), DrawKey(7, 2, 0, 8, 0)
    y_ref = ops._linear_sampled_raw(x, 512 * 1200, 512, mw, rw, mb, rb, kw, kb, _lib.COMPUTE_BF16)
    out3 = torch.full((len(sizes) + 1,), -1.0, device=dev)
    h3 = ops.kl_normal_begin(mus, rhos, pri, 2.0, out=out3, carry=True)
    y_wide = ops._linear_sampled_raw(x, 512 * 1200, 512, mus[2].view(1200, 1200), rhos[2].view(1200, 1200), mus[3], rhos[3],
                                     kw, kb, _lib.COMPUTE_BF16)               # a wide layer does not take it
    assert not h3.launched
    before = env["lib"].bnn_launch_count()
    y_car = ops._linear_sampled_raw(x, 512 * 1200, 512, mw, rw, mb, rb, kw, kb, _lib.COMPUTE_BF16)
    assert h3.launched and env["lib"].bnn_launch_count() == before + 1          # ONE launch: layer + KL first pass
    ops.mc_mean(y, kl=h3)
    assert np.array_equal(N(out3), want_kl)
    assert np.array_equal(N(y_car), N(y_ref))
    # ... and a handle nobody carried is launched by mc_mean itself
    h4 = ops.kl_normal_begin(mus, rhos, pri, 2.0, carry=True)
    ops.mc_mean(y, kl=h4)
    assert np.array_equal(N(h4.out), want_kl)
    # fp32 layer (4-wave narrow kernel) carrying a small model's KL (2048-scalar workgroups)
    xf = torch.randn(8, 64, 1200, device=dev)
    yf_ref = ops._linear_sampled_raw(xf, 64 * 1200, 64, mw, rw, mb, rb, kw, kb, _lib.COMPUTE_F32)
    h5 = ops.kl_normal_begin(mus[3:], rhos[3:], pri[3:], 1.0, carry=True)
    yf = ops._linear_sampled_raw(xf, 64 * 1200, 64, mw, rw, mb, rb, kw, kb, _lib.COMPUTE_F32)
    ops.mc_mean(y, kl=h5)
    assert np.array_equal(N(h5.out), N(ops.kl_normal(mus[3:], rhos[3:], pri[3:], 1.0)))
    assert np.array_equal(N(yf), N(yf_ref))
    ops.DRAW_ONCE_BF16 = True
    # a large reduction (more MC workgroups than the 2048-block cap) next to a one-tensor KL
    y2 = torch.randn(3, 700001, device=dev)
    h2 = ops.kl_normal_begin(mus[:1], rhos[:1], pri[:1], 1.0)
    got2 = N(ops.mc_mean(y2, kl=h2))
    assert np.array_equal(got2, N(ops.mc_mean(y2)))
    assert np.array_equal(N(h2.out), N(ops.kl_normal(mus[:1], rhos[:1], pri[:1], 1.0)))


def test_f32_mode_bf16x3_accuracy(env):
    """The fp32 mode's wide forward layers run as three-way bf16 splits on the bf16 MFMA (kComputeBf16x3).  Against a
    float64 contraction of the same fp32 operands: error <= 2e-6 of the output scale at K = 1200 (what a k-ordered fp32
    fmaf chain gives: ~3.5e-7 * sum|a b| / scale) -- on values with a wide dynamic range and exact cancellations, where
    a plain bf16 contraction (compute='bf16') is three orders of magnitude worse."""
    dev = env["dev"]
    ops = env["ops"]
    gen = torch.Generator().manual_seed(21)
    S, M, K, Nn = 2, 512, 1200, 1200
    x = torch.randn(S, M, K, generator=gen) * torch.exp(torch.randn(S, M, K, generator=gen) * 2.0)     # wide dynamic range
    w = torch.randn(S, Nn, K, generator=gen) * 0.05
    w[:, :, 1::2] = -w[:, :, 0::2] * (1 + 2.0 ** -12)                                                    # near-cancelling pairs
    x[:, :, 1::2] = x[:, :, 0::2]
    b = torch.randn(S, Nn, generator=gen)
    want = torch.einsum("smk,snk->smn", x.double(), w.double()) + b.double()[:, None, :]
    scale = float(torch.einsum("smk,snk->smn", x.double().abs(), w.double().abs()).mean())
    y = ops.linear_plain(x.to(dev), w.to(dev), b.to(dev), False, "f32").cpu().double()
    err = float((y - want).abs().max())
    assert err <= 2e-6 * scale * 8, (err, scale)            # max over 1.2 M outputs vs the MEAN sum|a b|: factor 8 head-room
    yb = ops.linear_plain(x.to(dev), w.to(dev), b.to(dev), False, "bf16").cpu().double()
    errb = float((yb - want).abs().max())
    assert errb > 100 * err                                  # the test would notice a silent fall-back to one bf16 term


@pytest.mark.parametrize("mode", ["f32", "bf16", "bf16_act"])
def test_large_batch_draws_once(env, mode):
    """>= ops.DRAW_ONCE_MIN_ROWS rows: the weights are drawn once by K1 and the kernel runs on explicit weights -- the
    same bits as the fused in-kernel draw (same DrawKey), ReLU and bias included; fp32 mode, bf16 mode, bf16 mode with
    bf16 activations in and out."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    from bayesianneuralnetworks_amd import _lib
    dev = env["dev"]
    ops = env["ops"]
    gen = torch.Generator().manual_seed(31)
    S, M, K, Nn = 2, 2048, 256, 160
    x = torch.randn(S, M, K, generator=gen).to(dev)
    mu, rho = (torch.randn(Nn, K, generator=gen) * 0.05).to(dev), (torch.randn(Nn, K, generator=gen) * 0.15 - 2).to(dev)
    mb, rb = torch.randn(Nn, generator=gen).to(dev), (torch.randn(Nn, generator=gen) * 0.15 - 2).to(dev)
    kw, kb = DrawKey(5, 1, 0, S, 0), DrawKey(5, 2, 0, S, 0)
    comp = _lib.COMPUTE_F32 if mode == "f32" else _lib.COMPUTE_BF16
    odt = torch.bfloat16 if mode == "bf16_act" else torch.float32
    if mode == "bf16_act":
        x = x.bfloat16()
    old = ops.DRAW_ONCE_MIN_ROWS
    try:
        ops.DRAW_ONCE_BF16 = False                 # (round-1 kernels: fused draw against K1 + the same kernel on explicit weights)
        ops.DRAW_ONCE_MIN_ROWS = 1 << 30
        fused = ops._linear_sampled_raw(x, M * K, M, mu, rho, mb, rb, kw, kb, comp, relu=True, out_dtype=odt)
        ops.DRAW_ONCE_MIN_ROWS = 2048
        before = env["lib"].bnn_launch_count()
        once = ops._linear_sampled_raw(x, M * K, M, mu, rho, mb, rb, kw, kb, comp, relu=True, out_dtype=odt)
        assert env["lib"].bnn_launch_count() == before + 3           # K1 (weights), K1 (bias), the contraction
    finally:
        ops.DRAW_ONCE_MIN_ROWS = old
        ops.DRAW_ONCE_BF16 = True
    assert np.array_equal(N(once.float()), N(fused.float()))
    assert float(once.float().min()) == 0.0                            # the fused ReLU ran


def test_mc_mean_with_epoch_advance(env):
    """bnn_mc_sum's advance_epoch: the reduction bumps the device epoch cell in the same launch."""
    dev = env["dev"]
    y = torch.randn(8, 512, 10, device=dev)
    cell = torch.zeros(4, dtype=torch.int32, device=dev)
    out = env["ops"].mc_mean(y, advance=cell)
    assert allclose(N(out), N(y).mean(0))
    out2 = env["ops"].mc_mean(y, scale=0.5, advance=cell)
    assert allclose(N(out2), 0.5 * N(y).sum(0))
    assert cell.tolist() == [2, 0, 0, 0]
    env["ops"].mc_mean(y)
    assert cell.tolist() == [2, 0, 0, 0]


# ------------------------------------------------------------------ K2 linear
def _layer_from_golden(g, dev, cls, *args):
    layer = cls(*args)
    with torch.no_grad():
        layer.weight.mean.copy_(torch.from_numpy(g["mu_w"]))
        layer.weight.scale.copy_(torch.from_numpy(g["rho_w"]))
        if "mu_b" in g:
            layer.bias.mean.copy_(torch.from_numpy(g["mu_b"]))
            layer.bias.scale.copy_(torch.from_numpy(g["rho_b"]))
    return layer.to(dev)


@pytest.mark.parametrize("name", LINEAR)
def test_linear_forward_golden_parity_mode(env, name):
    """eps from the reference's own draw -> w, b, y must match the reference (1e-5)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    g = load_golden(name)
    dev = env["dev"]
    o, i = g["mu_w"].shape
    layer = _layer_from_golden(g, dev, NormalLinear, i, o, "mu_b" in g)
    layer.weight.sample_with_eps(T(g["eps_w"], dev))
    if "mu_b" in g:
        layer.bias.sample_with_eps(T(g["eps_b"], dev))
    n0 = env["lib"].bnn_launch_count()
    y = layer(T(g["x"], dev), sample=False)
    assert env["lib"].bnn_launch_count() > n0
    w, b = layer.sampled
    assert allclose(N(w), g["w"])
    if "mu_b" in g:
        assert allclose(N(b), g["b"])
    assert allclose(N(y), g["y"])


@pytest.mark.parametrize("name", ["linear_4x3", "linear_7x11", "linear_7x11_nobias", "linear_64x48", "linear_1x1"])
def test_linear_backward_golden(env, name):
    """loss = sum(y * gy) + KL(n_batches=3): grads of mean / scale / x vs the reference's autograd."""
    from bayesianneuralnetworks_amd.nn import NormalLinear, KLDivergence, BayesianNetworkModule
    g = load_golden(name)
    dev = env["dev"]
    o, i = g["mu_w"].shape
    prior = torch.distributions.Normal(float(g["prior_mu"]), float(g["prior_sigma"]))
    layer = _layer_from_golden(g, dev, NormalLinear, i, o, "mu_b" in g, prior)

    class Net(BayesianNetworkModule):
        def __init__(self, L):
            super().__init__(1, 1, 1)
            self.layers = torch.nn.Sequential(L)

        def _forward(self, x):
            return self.layers(x)

    layer.weight.sample_with_eps(T(g["eps_w"], dev))
    if "mu_b" in g:
        layer.bias.sample_with_eps(T(g["eps_b"], dev))
    x = T(g["x"], dev).requires_grad_(True)
    y = layer(x, sample=False)
    kl = KLDivergence(number_of_batches=3)(Net(layer))
    assert abs(kl.item() - float(g["kl"])) <= 1e-5 * (1 + abs(float(g["kl"])))
    ((y * T(g["gy"], dev)).sum() + kl).backward()
    assert np.allclose(N(layer.weight.mean.grad), g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(layer.weight.scale.grad), g["g_rho_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(x.grad), g["g_x"], rtol=1e-4, atol=1e-5)
    if "mu_b" in g:
        assert np.allclose(N(layer.bias.mean.grad), g["g_mu_b"], rtol=1e-4, atol=1e-5)
        assert np.allclose(N(layer.bias.scale.grad), g["g_rho_b"], rtol=1e-4, atol=1e-5)


def _oracle_layer_draw(orc, layer, s, epoch_dev=0):
    kw = layer.weight.draw_key
    ew = orc.eps_fill(kw.seed, kw.stream, kw.sample0 + s, kw.epoch_host, epoch_dev, tuple(layer.weight.shape), kw.gen)
    w = orc.sample_affine(N(layer.weight.mean), N(layer.weight.scale), ew)
    b = None
    if layer.bias is not None:
        kb = layer.bias.draw_key
        eb = orc.eps_fill(kb.seed, kb.stream, kb.sample0 + s, kb.epoch_host, epoch_dev, tuple(layer.bias.shape), kb.gen)
        b = orc.sample_affine(N(layer.bias.mean), N(layer.bias.scale), eb)
    return w, b


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 3, 4), (6, 11, 7), (33, 48, 64), (130, 100, 90),
                                   (512, 784, 1200), (64, 1200, 10), (300, 36, 200)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_fused_philox_vs_oracle(env, shape, bias):
    """Production path: in-kernel draw (fused) == oracle Philox twin + exact GEMM; and the
    lazily materialised .sampled equals what the fused kernel used."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    M, K, Nn = shape
    dev = env["dev"]
    torch.manual_seed(M * 7 + K)
    layer = NormalLinear(K, Nn, bias).to(dev)
    env["bnn"].manual_seed(555)
    x = torch.randn(M, K, device=dev)
    n0 = env["lib"].bnn_launch_count()
    y = layer(x)
    assert env["lib"].bnn_launch_count() == n0 + 1          # ONE fused launch
    w, b = _oracle_layer_draw(env["orc"], layer, 0)
    # outputs of the K >= 784 shapes have rms 3-5: 1e-5 of the output scale (allclose_scaled);
    # unit-scale shapes use the plain 1e-5 bar.
    want = env["orc"].linear(N(x), w, b)
    assert allclose_scaled(N(y), want) if K >= 784 else allclose(N(y), want)
    ws, bs = layer.sampled
    assert allclose(N(ws), w)
    if bias:
        assert allclose(N(bs), b)
    y2 = layer(x, sample=False)                             # reuses the recorded draw
    assert torch.equal(y, y2)
    y3 = layer(x)                                           # new draw
    assert not torch.equal(y, y3)


def test_linear_reference_kat_collapsed_posterior(env):
    """tests/test_nn/test_dense.py:38-70 on the device."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    from torch.nn import init
    for (i, o, b) in [(1, 1, True), (3, 4, True), (11, 7, True), (1, 1, False), (3, 4, False), (11, 7, False)]:
        nl = NormalLinear(i, o, b).to(env["dev"])
        init.constant_(nl.weight.mean, 1)
        init.constant_(nl.weight.scale, -100)
        if b:
            init.constant_(nl.bias.mean, 3)
            init.constant_(nl.bias.scale, -100)
        nl.sample()
        x = torch.ones_like(nl.weight.mean)
        result = nl(x)
        assert isinstance(nl.sampled, tuple) and len(nl.sampled) == 2
        assert allclose(N(result), np.full(result.shape, i + (3 if b else 0)))


def test_linear_bf16_mode_tolerance(env):
    """bf16 operands / fp32 accumulate vs the fp32 path on the same draw: operands carry
    2^-9 relative rounding each; with K = 1200 random-sign terms the observed error is
    < 1% of the output rms.  Bound used: 2e-2 * rms(y)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    dev = env["dev"]
    torch.manual_seed(1)
    layer = NormalLinear(1200, 1200).to(dev)
    x = torch.randn(512, 1200, device=dev)
    y32 = layer(x)
    layer.compute = "bf16"
    y16 = layer(x, sample=False)
    rms = y32.pow(2).mean().sqrt().item()
    err = (y16 - y32).abs().max().item()
    assert err < 2e-2 * rms, (err, rms)
    assert err > 0


# ------------------------------------------------------------------ K2 conv
@pytest.mark.parametrize("name", CONV)
def test_conv_forward_and_backward_golden(env, name):
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    g = load_golden(name)
    dev = env["dev"]
    sh, sw, ph, pw, dh, dw, groups = [int(v) for v in g["conv"]]
    O, Cg, KH, KW = g["mu_w"].shape
    layer = _layer_from_golden(g, dev, NormalConv2d, Cg * groups, O, (KH, KW), (sh, sw), (ph, pw), (dh, dw),
                               groups, "mu_b" in g)
    layer.weight.sample_with_eps(T(g["eps_w"], dev))
    if "mu_b" in g:
        layer.bias.sample_with_eps(T(g["eps_b"], dev))
    x = T(g["x"], dev).requires_grad_(True)
    y = layer(x, sample=False)
    assert allclose(N(y), g["y"])
    (y * T(g["gy"], dev)).sum().backward()
    # the fixture's loss also holds KL/1: add its gradient through the oracle-checked kernel
    from bayesianneuralnetworks_amd.nn import KLDivergence, BayesianNetworkModule

    class Net(BayesianNetworkModule):
        def __init__(self, L):
            super().__init__(1, 1, 1)
            self.layers = torch.nn.Sequential(L)

        def _forward(self, x):
            return self.layers(x)

    KLDivergence()(Net(layer)).backward()
    assert np.allclose(N(layer.weight.mean.grad), g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(layer.weight.scale.grad), g["g_rho_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(x.grad), g["g_x"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cfg", [
    # (B, C, H, W, O, k, stride, pad, dil, groups, bias)
    (1, 1, 10, 10, 1, 1, 1, 0, 1, 1, True),          # reference test shape (1,1,k1)
    (1, 3, 10, 10, 4, 3, 1, 1, 1, 1, True),          # reference test shape (3,4,k3,pad1)
    (3, 4, 9, 9, 6, 3, 2, 2, 2, 2, False),
    (16, 64, 6, 6, 64, 3, 2, 1, 1, 1, True),         # MNIST/FMNIST implicit-GEMM shape
    (8, 128, 4, 4, 128, 3, 1, 1, 1, 1, True),        # CIFAR10 shape
])
def test_conv_fused_philox_vs_oracle(env, cfg):
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    B, C, H, W, O, k, s, p, d, groups, bias = cfg
    dev = env["dev"]
    torch.manual_seed(B + C)
    layer = NormalConv2d(C, O, k, s, p, d, groups, bias).to(dev)
    env["bnn"].manual_seed(31)
    x = torch.randn(B, C, H, W, device=dev)
    n0 = env["lib"].bnn_launch_count()
    y = layer(x)
    # generic implicit-GEMM kernel: 1 launch; fast path (groups 1, K % 8 == 0, O >= 16): im2col + fused GEMM
    fast = groups == 1 and (C * k * k) % 8 == 0 and C * k * k >= 32 and O >= 16
    assert env["lib"].bnn_launch_count() == n0 + (2 if fast else 1)
    w, b = _oracle_layer_draw(env["orc"], layer, 0)
    want = env["orc"].conv2d(N(x), w, b, (s, s), (p, p), (d, d), groups)
    assert y.shape == want.shape
    assert allclose(N(y), want)
    assert torch.equal(layer(x, sample=False), y)


def test_conv_reference_kat_and_errors(env):
    """tests/test_nn/test_conv.py:97-120 on the device + conv.py:15-18 errors."""
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from torch.nn import init
    for (i, o, k, s, pad, d, g_, b) in [(1, 1, 1, 1, 0, 1, 1, True), (3, 4, 3, 1, 1, 1, 1, True),
                                        (1, 1, 1, 1, 0, 1, 1, False), (3, 4, 3, 1, 1, 1, 1, False)]:
        nc = NormalConv2d(i, o, k, s, pad, d, g_, b).to(env["dev"])
        init.constant_(nc.weight.mean, 1)
        init.constant_(nc.weight.scale, -100)
        if b:
            init.constant_(nc.bias.mean, 3)
            init.constant_(nc.bias.scale, -100)
        nc.sample()
        x = torch.ones(1, i, 10, 10, device=env["dev"])
        expected = torch.nn.functional.conv2d(x.cpu(), torch.ones(o, i // g_, k, k), None, s, pad, d, g_)
        assert allclose(N(nc(x)), expected.numpy() + (3 if b else 0))
    with pytest.raises(ValueError):
        NormalConv2d(3, 4, 3, groups=2)
    with pytest.raises(ValueError):
        NormalConv2d(4, 3, 3, groups=2)


# ------------------------------------------------------------------ MC loop / network
class _MLP:
    @staticmethod
    def build(dev, dims=(784, 1200, 1200, 10), samples=2, seed=0):
        from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule
        post = seeded.mlp_posteriors(dims, seed=seed)

        class Net(BayesianNetworkModule):
            def __init__(self):
                super().__init__(dims[0], dims[-1], samples)
                mods = []
                for j, (mw, rw, mb, rb) in enumerate(post):
                    L = NormalLinear(mw.shape[1], mw.shape[0])
                    with torch.no_grad():
                        L.weight.mean.copy_(mw)
                        L.weight.scale.copy_(rw)
                        L.bias.mean.copy_(mb)
                        L.bias.scale.copy_(rb)
                    mods.append(L)
                    if j < len(post) - 1:
                        mods.append(torch.nn.ReLU())
                self.layers = torch.nn.Sequential(*mods)

            def _forward(self, x):
                return self.layers(x)

        return Net().to(dev), post


def test_north_star_mlp_vs_reference_golden(env):
    """784-1200-1200-10, batch 512, 2 MC samples, eps = the reference's own mt19937 draws."""
    from bayesianneuralnetworks_amd.nn import KLDivergence
    g = load_golden("mlp_784_1200_1200_10")
    dev = env["dev"]
    net, post = _MLP.build(dev)
    x = seeded.mlp_input(512, 784, seed=int(g["x_seed"])).to(dev)
    shapes = [(tuple(p[0].shape), tuple(p[2].shape)) for p in post]
    eps = seeded.eps_like_reference(int(g["eps_seed"]), shapes, samples=2)
    linears = [m for m in net.layers if hasattr(m, "weight")]
    ys = []
    for s in range(2):
        h = x
        for li, L in enumerate(linears):
            L.weight.sample_with_eps(eps[s][li][0].to(dev))
            L.bias.sample_with_eps(eps[s][li][1].to(dev))
            h = L(h, sample=False)
            if li < len(linears) - 1:
                h = torch.relu(h)
        ys.append(N(h))
    assert allclose_scaled(ys[0], g["y0"]) and allclose_scaled(ys[1], g["y1"])
    kl = KLDivergence()(net)
    assert abs(kl.item() - float(g["kl"])) <= 1e-5 * (1 + float(g["kl"]))
    pm = N(env["ops"].mc_mean(torch.stack([T(ys[0], dev), T(ys[1], dev)])))
    assert allclose_scaled(pm, g["pred_mean"])


def test_mc_batched_equals_oracle_loop(env):
    """All S samples in one grid per layer == the serial MC loop (container.py:36-37) run by
    the oracle on the same per-sample draws; independent of how samples are split."""
    dev = env["dev"]
    net, post = _MLP.build(dev, dims=(96, 200, 120, 10), samples=4, seed=3)
    net.mc_batched = True
    env["bnn"].manual_seed(77)
    x = torch.randn(40, 96, device=dev)
    n0 = env["lib"].bnn_launch_count()
    ys = net(x)
    assert env["lib"].bnn_launch_count() == n0 + 3           # one launch per Bayesian layer
    assert isinstance(ys, list) and len(ys) == 4
    linears = [m for m in net.layers if hasattr(m, "weight")]
    for s in range(4):
        h = N(x)
        for li, L in enumerate(linears):
            w, b = _oracle_layer_draw(env["orc"], L, s)
            h = env["orc"].linear(h, w, b)
            if li < 2:
                h = np.maximum(h, 0)
        assert allclose_scaled(N(ys[s]), h)
    # sample ids, not grid position, decide the draw: samples 2..3 alone give the same outputs
    from bayesianneuralnetworks_amd import _mc
    keys = [(L.weight.draw_key.epoch_host) for L in linears]
    with _mc.McContext(2, 40, sample0=2):
        h = x
        for li, L in enumerate(linears):
            L.weight.sample(2, 2, keys[li])
            L.bias.sample(2, 2, keys[li])
            h = L(h, sample=False)
            if li < 2:
                h = torch.relu(h)
    assert torch.equal(h.view(2, 40, 10)[0], ys[2]) and torch.equal(h.view(2, 40, 10)[1], ys[3])


def test_serial_loop_api_and_item_or_list(env):
    dev = env["dev"]
    net, _ = _MLP.build(dev, dims=(16, 32, 8), samples=3, seed=4)
    x = torch.randn(5, 16, device=dev)
    ys = net(x)
    assert isinstance(ys, list) and len(ys) == 3 and not torch.equal(ys[0], ys[1])
    y1 = net(x, samples=1)
    assert isinstance(y1, torch.Tensor) and y1.shape == (5, 8)


def test_training_step_gradients_match_oracle(env):
    """fwd (fused, Philox) + KL + backward on a small MLP: grads vs the oracle's formulas
    evaluated on the same draws."""
    from bayesianneuralnetworks_amd.nn import KLDivergence
    orc = env["orc"]
    dev = env["dev"]
    net, _ = _MLP.build(dev, dims=(24, 40, 6), samples=2, seed=5)
    net.mc_batched = True
    env["bnn"].manual_seed(5)
    x = torch.randn(9, 24, device=dev)
    gy = torch.randn(2, 9, 6, device=dev)
    ys = net(x)
    kl = KLDivergence(number_of_batches=4)(net)
    loss = sum((ys[s] * gy[s]).sum() for s in range(2)) + kl
    loss.backward()
    L0, L1 = net.layers[0], net.layers[2]
    g_mu0 = np.zeros(L0.weight.shape, np.float64)
    g_rho0 = np.zeros(L0.weight.shape, np.float64)
    g_mu1 = np.zeros(L1.weight.shape, np.float64)
    g_rho1 = np.zeros(L1.weight.shape, np.float64)
    g_rho_b1 = np.zeros(L1.bias.shape, np.float64)
    for s in range(2):
        w0, b0 = _oracle_layer_draw(orc, L0, s)
        w1, b1 = _oracle_layer_draw(orc, L1, s)
        h0 = orc.linear(N(x), w0, b0)
        a0 = np.maximum(h0, 0)
        g1 = N(gy[s]).astype(np.float64)
        gw1 = g1.T @ a0
        ga0 = g1 @ w1
        gh0 = ga0 * (h0 > 0)
        gw0 = gh0.T @ N(x)
        for (gw, L, gm, gr) in ((gw0, L0, g_mu0, g_rho0), (gw1, L1, g_mu1, g_rho1)):
            kw = L.weight.draw_key
            ew = orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, 0, tuple(L.weight.shape), kw.gen)
            a, b_ = orc.sample_affine_bwd(gw.astype(np.float32), N(L.weight.scale), ew)
            gm += a
            gr += b_
        kb = L1.bias.draw_key
        eb = orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, 0, tuple(L1.bias.shape), kb.gen)
        g_rho_b1 += orc.sample_affine_bwd(g1.sum(0).astype(np.float32), N(L1.bias.scale), eb)[1]
    sc = 1.0 / (4 * 4)
    for (L, gm, gr) in ((L0, g_mu0, g_rho0), (L1, g_mu1, g_rho1)):
        km, kr = orc.kl_bwd(N(L.weight.mean), N(L.weight.scale), 0.0, 0.1, sc / L.weight.mean.numel())
        assert np.allclose(N(L.weight.mean.grad), gm + km, rtol=1e-4, atol=1e-4)
        assert np.allclose(N(L.weight.scale.grad), gr + kr, rtol=1e-4, atol=1e-4)
    km, kr = orc.kl_bwd(N(L1.bias.mean), N(L1.bias.scale), 0.0, 0.1, sc / L1.bias.mean.numel())
    assert np.allclose(N(L1.bias.scale.grad), g_rho_b1 + kr, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ Flipout
@pytest.mark.parametrize("name", ["flipout_linear_12x7", "flipout_linear_64x48"])
def test_flipout_linear_golden_forward_backward(env, name):
    """dense.py:63-83 on the device = K1 with eps = outer(R, S) + ONE HIP contraction; outputs and the
    reference's autograd gradients (mean, scale, x)."""
    from bayesianneuralnetworks_amd.nn import FlipoutNormalLinear
    g = load_golden(name)
    dev = env["dev"]
    o, i = g["mu_w"].shape
    layer = FlipoutNormalLinear(i, o).to(dev)
    with torch.no_grad():
        layer.weight.mean.copy_(T(g["mu_w"], dev))
        layer.weight.scale.copy_(T(g["rho_w"], dev))
    layer.R, layer.S = T(g["R"], dev), T(g["S"], dev)
    x = T(g["x"], dev).requires_grad_(True)
    n0 = env["lib"].bnn_launch_count()
    y = layer(x, sample=False)
    assert env["lib"].bnn_launch_count() >= n0 + 2
    assert allclose(N(y), g["y"])
    (y * T(g["gy"], dev)).sum().backward()
    assert np.allclose(N(layer.weight.mean.grad), g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(layer.weight.scale.grad), g["g_rho_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(x.grad), g["g_x"], rtol=1e-4, atol=1e-5)
    layer(x)                                            # sample=True draws fresh signs
    assert set(layer.R.unique().tolist()) <= {-1.0, 1.0} and layer.S.shape == (i,)


@pytest.mark.parametrize("name,stride", [("flipout_conv_3_4_k3_p1", 1), ("flipout_conv_4_6_k3_s2", 2)])
def test_flipout_conv2d_golden_forward_backward(env, name, stride):
    """conv.py:199-227 on the device: both contractions through the HIP implicit GEMM."""
    from bayesianneuralnetworks_amd.nn import FlipOutNormalConv2d
    g = load_golden(name)
    dev = env["dev"]
    o, c = g["mu_w"].shape[:2]
    layer = FlipOutNormalConv2d(c, o, 3, stride=stride, padding=1).to(dev)
    with torch.no_grad():
        layer.weight.mean.copy_(T(g["mu_w"], dev))
        layer.weight.scale.copy_(T(g["rho_w"], dev))
    layer.R, layer.S = T(g["R"], dev), T(g["S"], dev)
    x = T(g["x"], dev).requires_grad_(True)
    n0 = env["lib"].bnn_launch_count()
    y = layer(x, sample=False)
    assert env["lib"].bnn_launch_count() >= n0 + 2
    assert allclose(N(y), g["y"])
    (y * T(g["gy"], dev)).sum().backward()
    assert np.allclose(N(layer.weight.mean.grad), g["g_mu_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(layer.weight.scale.grad), g["g_rho_w"], rtol=1e-4, atol=1e-5)
    assert np.allclose(N(x.grad), g["g_x"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ backward kernels
def _bwd_case(env, S, M, N, K, seed, shared_x=False):
    from bayesianneuralnetworks_amd._rng import DrawKey
    gen = torch.Generator().manual_seed(seed)
    mu = torch.randn(N, K, generator=gen) * 0.1
    rho = torch.randn(N, K, generator=gen) * 0.15 - 2.0
    x = torch.randn((M, K) if shared_x else (S, M, K), generator=gen)
    gy = torch.randn(S, M, N, generator=gen)
    key = DrawKey(99 + seed, 21, 0, S, 7)
    orc = env["orc"]
    eps = [orc.eps_fill(key.seed, key.stream, s, key.epoch_host, 0, (N, K), key.gen) for s in range(S)]
    return mu, rho, x, gy, key, eps


def _oracle_linear_bwd(orc, mu, rho, x, gy, eps, shared_x, rounder=None):
    """oracle.linear_bwd: float64 restatement of autograd through F.linear (dense.py:60) and mu + sigma * eps (core.py:45)."""
    return orc.linear_bwd(N(mu), N(rho), N(x), N(gy), eps, shared_x, rounder)


BWD_SHAPES = [(1, 1, 1, 4), (2, 9, 6, 24), (3, 33, 10, 1200), (2, 70, 130, 260), (8, 64, 48, 136),
              (2, 5, 7, 11), (1, 40, 3, 9), (2, 300, 10, 72), (2, 520, 7, 40)]     # (the last two: narrow layer, 2 and 3 row slices)


@pytest.mark.parametrize("S,M,Nn,K", BWD_SHAPES)
@pytest.mark.parametrize("shared_x", [False, True])
def test_weight_gradient_kernel_vs_oracle_f32(env, S, M, Nn, K, shared_x):
    """bnn_linear_backward_weight_sampled, exact fp32: g_mu / g_rho vs the oracle on the same draws
    (ragged tiles, odd K, sample split through the workspace for small outputs)."""
    mu, rho, x, gy, key, eps = _bwd_case(env, S, M, Nn, K, 1, shared_x)
    dev = env["dev"]
    xd = x.to(dev)
    mud, rhod = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
    y = env["ops"].linear_sampled(xd, mud, rhod, None, None, key, None, shared_x)
    g_mu, g_rho = torch.autograd.grad(y, (mud, rhod), gy.to(dev))
    want_mu, want_rho, _ = _oracle_linear_bwd(env["orc"], mu, rho, x, gy, eps, shared_x)
    assert allclose_scaled(N(g_mu), want_mu)
    assert allclose_scaled(N(g_rho), want_rho)


@pytest.mark.parametrize("S,M,Nn,K", BWD_SHAPES)
def test_input_gradient_kernel_vs_oracle_f32(env, S, M, Nn, K):
    """bnn_linear_backward_input_sampled (fused re-draw, aligned shapes) / bnn_linear_backward_input
    (explicit weights, the rest): gx vs the oracle, per sample and summed for a shared input."""
    for shared_x in (False, True):
        mu, rho, x, gy, key, eps = _bwd_case(env, S, M, Nn, K, 2, shared_x)
        dev = env["dev"]
        xd = x.to(dev).requires_grad_(True)
        y = env["ops"].linear_sampled(xd, mu.to(dev), rho.to(dev), None, None, key, None, shared_x)
        (gx,) = torch.autograd.grad(y, (xd,), gy.to(dev))
        _, _, want = _oracle_linear_bwd(env["orc"], mu, rho, x, gy, eps, shared_x)
        if shared_x:
            want = want.sum(0)
        assert allclose_scaled(N(gx), want)


@pytest.mark.parametrize("S,M,Nn,K", [(2, 9, 8, 24), (3, 33, 16, 1200), (8, 64, 48, 136), (2, 70, 136, 264)])
def test_backward_kernels_bf16_mode(env, S, M, Nn, K):
    """bf16 operands (x, gy and the re-drawn W rounded to bf16), fp32 accumulate; transposed LDS reads
    in the weight gradient.  Checked against the oracle fed the same bf16-rounded operands: 2e-3 of the
    output scale (fp32 accumulation order only), and against the exact gradient at 2e-2."""
    orc = env["orc"]
    dev = env["dev"]
    mu, rho, x, gy, key, eps = _bwd_case(env, S, M, Nn, K, 3)
    for act_dtype in (torch.float32, torch.bfloat16):
        xd = x.to(dev).to(act_dtype).requires_grad_(True)
        mud, rhod = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
        y = env["ops"].linear_sampled(xd, mud, rhod, None, None, key, None, False, compute="bf16", out_dtype=act_dtype)
        gx, g_mu, g_rho = torch.autograd.grad(y, (xd, mud, rhod), gy.to(dev).to(act_dtype))
        want_mu, want_rho, want_x = _oracle_linear_bwd(orc, mu, rho, x, gy, eps, False, rounder=orc.bf16_round)
        exact_mu, exact_rho, exact_x = _oracle_linear_bwd(orc, mu, rho, x, gy, eps, False)
        if Nn > 16:
            assert allclose_scaled(N(g_mu), want_mu, 2e-3)
            assert allclose_scaled(N(g_rho), want_rho, 2e-3)
            assert allclose_scaled(N(gx.float()), want_x, 2e-3 if act_dtype == torch.float32 else 1e-2)
        # (N <= 16 runs the narrow-layer kernel, which keeps fp32 operands: closer to the exact gradient
        # than to the bf16-rounded restatement)
        assert allclose_scaled(N(g_mu), exact_mu, 2e-2) and allclose_scaled(N(g_rho), exact_rho, 2e-2)
        assert allclose_scaled(N(gx.float()), exact_x, 2e-2)


@pytest.mark.parametrize("S,rows,cols,ld_in", [(1, 8, 8, 8), (3, 70, 136, 192), (2, 1200, 1200, 1216), (2, 130, 64, 64), (1, 10, 1200, 1216)])
def test_transpose_drawn_weights(env, S, rows, cols, ld_in):
    """bnn_transpose_bf16: (S, rows, cols) bf16 of pitch ld_in -> (S, cols, roundup(rows, 64)), zeros beyond column `rows`;
    bit-exact (a copy)."""
    dev = env["dev"]
    w = torch.randn(S, rows, ld_in, device=dev).bfloat16()
    out = env["ops"]._transpose_drawn_raw(w[:, :, :ld_in], cols)
    ldn = (rows + 63) // 64 * 64
    assert out.shape == (S, cols, ldn)
    assert torch.equal(out[:, :, :rows], w[:, :, :cols].transpose(1, 2))
    assert not out[:, :, rows:].any()


@pytest.mark.parametrize("wgen", [0, 1])
@pytest.mark.parametrize("S,M,Nn,K", [(2, 70, 136, 264), (8, 64, 48, 136), (2, 128, 1200, 784), (3, 256, 24, 8), (2, 384, 200, 1200), (5, 100, 72, 72),
                                      (1, 1, 328, 40)])
def test_input_gradient_on_the_drawn_weights_equals_the_redraw_path(env, S, M, Nn, K, wgen):
    """bf16 training: gx = gy . w_s contracts on the weights the forward drew (transpose + dense kernel) -- same bf16 operands as
    the kernel that re-draws them inside the contraction, so the two agree to fp32 accumulation order, and both with the oracle on
    bf16-rounded operands."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    orc, dev, ops = env["orc"], env["dev"], env["ops"]
    mu, rho, x, gy, _, _ = _bwd_case(env, S, M, Nn, K, 4)
    key = DrawKey(41, 3, 0, S, 5, gen=wgen)
    eps = [orc.eps_fill(key.seed, key.stream, s, key.epoch_host, 0, (Nn, K), key.gen) for s in range(S)]
    got = {}
    n_launch = {}
    for on in (True, False):
        ops.DGRAD_ON_DRAWN = on
        try:
            xd = x.to(dev).bfloat16().requires_grad_(True)
            y = ops.linear_sampled(xd, mu.to(dev), rho.to(dev), None, None, key, None, False, compute="bf16", out_dtype=torch.bfloat16)
            n0 = env["lib"].bnn_launch_count()
            (got[on],) = torch.autograd.grad(y, (xd,), gy.to(dev).bfloat16())
            n_launch[on] = env["lib"].bnn_launch_count() - n0
        finally:
            ops.DGRAD_ON_DRAWN = True
    assert n_launch[True] == 2 and n_launch[False] == 1          # transpose + dense launch; the fused re-draw kernel
    _, _, want = _oracle_linear_bwd(orc, mu, rho, x, gy, eps, False, rounder=orc.bf16_round)
    assert_close_scaled(N(got[True].float()), want, 1e-2)
    assert_close_scaled(N(got[False].float()), want, 1e-2)
    assert_close_scaled(N(got[True].float()), N(got[False].float()), 1e-2)


@pytest.mark.parametrize("mode,M,wgen", [("f32", 70, 0), ("bf16", 256, 0), ("bf16", 256, 1), ("bf16", 72, 0), ("bf16", 72, 1)])
def test_bias_gradient_fused_into_weight_gradient(env, mode, M, wgen):
    """>= 64 output tiles: the k-tile-0 workgroups of the weight-gradient launch also produce the bias
    gradient (column sums by an MFMA against ones, bias draw's backward in the epilogue).  bf16 with
    M % 256 == 0 takes the LDS-DMA kernel, M = 72 the register-staged one.  wgen = 1: the draws keyed with the 16-bit stream
    (the bf16 mode's default) -- the LDS-DMA kernel's helper waves then take whole 8-eps blocks, shared by two lane groups."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    orc, dev = env["orc"], env["dev"]
    S, Nn, K = 2, 520, 1024
    gen = torch.Generator().manual_seed(21)
    mu = torch.randn(Nn, K, generator=gen) * 0.05
    rho = torch.randn(Nn, K, generator=gen) * 0.15 - 2.0
    mub = torch.randn(Nn, generator=gen) * 0.05
    rhob = torch.randn(Nn, generator=gen) * 0.15 - 2.0
    x = torch.randn(S, M, K, generator=gen)
    gy = torch.randn(S, M, Nn, generator=gen)
    kw, kb = DrawKey(77, 5, 0, S, 2, gen=wgen), DrawKey(77, 6, 0, S, 2, gen=wgen)
    adt = torch.float32 if mode == "f32" else torch.bfloat16
    xd = x.to(dev).to(adt)
    mbd, rbd = mub.to(dev).requires_grad_(True), rhob.to(dev).requires_grad_(True)
    n0 = env["lib"].bnn_launch_count()
    y = env["ops"].linear_sampled(xd, mu.to(dev), rho.to(dev), mbd, rbd, kw, kb, False, compute=mode, out_dtype=adt)
    # bias gradient alone: needs neither the weight-gradient launch nor the fused path
    g_mu_b0, g_rho_b0 = torch.autograd.grad(y, (mbd, rbd), gy.to(dev).to(adt), retain_graph=True)
    md, rd = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
    y = env["ops"].linear_sampled(xd, md, rd, mbd, rbd, kw, kb, False, compute=mode, out_dtype=adt)
    n1 = env["lib"].bnn_launch_count()
    g_mu_w, g_rho_w, g_mu_b, g_rho_b = torch.autograd.grad(y, (md, rd, mbd, rbd), gy.to(dev).to(adt))
    assert env["lib"].bnn_launch_count() == n1 + 1            # ONE launch: weights and bias together
    rnd = None if mode == "f32" else orc.bf16_round
    want_mu = np.zeros(Nn)
    want_rho = np.zeros(Nn)
    for s_ in range(S):
        cs = (N(gy[s_]) if rnd is None else rnd(N(gy[s_]))).astype(np.float64).sum(0)
        eb = orc.eps_fill(kb.seed, kb.stream, s_, kb.epoch_host, 0, (Nn,), kb.gen)
        a, b = orc.sample_affine_bwd(np.ones(Nn, np.float32), N(rhob), eb)
        want_mu += cs * a
        want_rho += cs * b
    tol = 1e-5 if mode == "f32" else 2e-3
    for got in (g_mu_b, g_mu_b0):
        assert allclose_scaled(N(got), want_mu, tol)
    for got in (g_rho_b, g_rho_b0):
        assert allclose_scaled(N(got), want_rho, tol)
    ew = [orc.eps_fill(kw.seed, kw.stream, s_, kw.epoch_host, 0, (Nn, K), kw.gen) for s_ in range(S)]
    wm, wr, _ = _oracle_linear_bwd(orc, mu, rho, x, gy, ew, False, rounder=rnd)
    assert allclose_scaled(N(g_mu_w), wm, tol) and allclose_scaled(N(g_rho_w), wr, tol)
    del n0


def test_narrow_layer_backward_accumulates_and_folds_bias(env):
    """bnn_linear_backward_narrow_sampled through the C-ABI: weights, bias (column sums folded over the row slices, the bias draw's
    backward) and the fused KL gradient leave in ONE tail launch; accumulate = 1 adds to what is there; values against the oracle."""
    import ctypes
    from bayesianneuralnetworks_amd import _lib
    from bayesianneuralnetworks_amd._rng import DrawKey
    orc, dev, ops, lib = env["orc"], env["dev"], env["ops"], env["lib"]
    S, M, Nn, K = 3, 300, 10, 72
    mu, rho, x, gy, _, _ = _bwd_case(env, S, M, Nn, K, 8)
    g = torch.Generator().manual_seed(5)
    rhob = torch.randn(Nn, generator=g) * 0.15 - 2.0
    kw, kb = DrawKey(19, 3, 0, S, 2), DrawKey(19, 4, 0, S, 2)
    _lib.ensure_workspace(dev)
    xd, gyd, mud, rhod, rbd = x.to(dev), gy.to(dev), mu.to(dev), rho.to(dev), rhob.to(dev)
    gm, gr = torch.zeros(Nn, K, device=dev), torch.zeros(Nn, K, device=dev)
    gmb, grb = torch.zeros(Nn, device=dev), torch.zeros(Nn, device=dev)
    gx = torch.empty(S, M, K, device=dev)
    rw, rb = ops._rng_struct(kw, dev), ops._rng_struct(kb, dev)
    P = _lib.ptr

    def call(acc):
        n0 = lib.bnn_launch_count()
        _lib.check(lib.bnn_linear_backward_narrow_sampled(P(xd), M * K, K, P(gyd), M * Nn, Nn, P(mud), P(rhod), P(gx), M * K, K, P(gm), P(gr),
                                                          P(rbd), P(gmb), P(grb), M, Nn, K, S, ctypes.byref(rw), ctypes.byref(rb), None, 0, acc,
                                                          _lib.stream_ptr(dev)), "narrow")
        assert lib.bnn_launch_count() == n0 + 2                   # the pass over the activations + ONE tail launch
    call(0)
    first = [t.clone() for t in (gm, gr, gmb, grb)]
    eps = [orc.eps_fill(kw.seed, kw.stream, s, kw.epoch_host, 0, (Nn, K), kw.gen) for s in range(S)]
    want_mu, want_rho, want_x = _oracle_linear_bwd(orc, mu, rho, x, gy, eps, False)
    assert_close_scaled(N(gm), want_mu, 1e-5, "g_mu")
    assert_close_scaled(N(gr), want_rho, 1e-5, "g_rho")
    assert_close_scaled(N(gx), want_x, 1e-5, "gx")
    wb_mu, wb_rho = np.zeros(Nn), np.zeros(Nn)
    for s in range(S):
        cs = N(gy[s]).astype(np.float64).sum(0)
        eb = orc.eps_fill(kb.seed, kb.stream, s, kb.epoch_host, 0, (Nn,), kb.gen)
        a, b = orc.sample_affine_bwd(np.ones(Nn, np.float32), N(rhob), eb)
        wb_mu += cs * a
        wb_rho += cs * b
    assert_close_scaled(N(gmb), wb_mu, 1e-5, "g_mu_b")
    assert_close_scaled(N(grb), wb_rho, 1e-5, "g_rho_b")
    call(1)
    for t, f in zip((gm, gr, gmb, grb), first):
        assert torch.allclose(t, 2 * f, rtol=1e-6, atol=1e-7)


def test_bias_colsum_and_relu_mask_kernels(env):
    dev = env["dev"]
    gy = torch.randn(3, 37, 130, device=dev)
    out = env["ops"]._colsum_raw(gy)
    assert allclose_scaled(N(out), N(gy).astype(np.float64).sum(1))
    for nn_ in (1, 10, 16, 17, 136):                    # narrow, scalar and 16-B-vector kernels
        for dt in (torch.float32, torch.bfloat16):
            gq = torch.randn(2, 515, nn_, device=dev).to(dt)
            assert allclose_scaled(N(env["ops"]._colsum_raw(gq)), N(gq.float()).astype(np.float64).sum(1))
    outh = env["ops"]._colsum_raw(gy.bfloat16())
    assert allclose_scaled(N(outh), N(gy.bfloat16().float()).astype(np.float64).sum(1))
    y = torch.randn(3, 37, 130, device=dev)
    for gd in (torch.float32, torch.bfloat16):
        for yd in (torch.float32, torch.bfloat16):
            m = env["ops"]._relu_backward_raw(gy.to(gd), y.to(yd))
            assert torch.equal(m, gy.to(gd) * (y.to(yd) > 0).to(gd))
    # 16-B form (both bf16, n % 8 == 0), with +-0 and denormals in y
    g8 = torch.randn(4, 64, 48, device=dev).bfloat16()
    y8 = torch.randn(4, 64, 48, device=dev).relu_().bfloat16()
    y8.view(-1)[:4] = torch.tensor([0.0, -0.0, 1e-40, -1e-40], device=dev).bfloat16()
    assert torch.equal(env["ops"]._relu_backward_raw(g8, y8), g8 * (y8 > 0).to(torch.bfloat16))


def test_weight_gradient_is_bitwise_reproducible_and_accumulates(env):
    """Fixed-order reductions (incl. the split over samples): two runs agree bit for bit; accumulate adds."""
    from bayesianneuralnetworks_amd import _lib
    from bayesianneuralnetworks_amd.ops import _rng_struct
    import ctypes
    dev = env["dev"]
    for (S, M, Nn, K) in ((8, 64, 10, 1200), (2, 100, 200, 264)):
        mu, rho, x, gy, key, eps = _bwd_case(env, S, M, Nn, K, 4)
        xd, gyd, rhod = x.to(dev), gy.to(dev), rho.to(dev)
        _lib.ensure_workspace(dev)
        r = _rng_struct(key, dev)
        outs = []
        for acc in (0, 0, 1):
            gm = torch.full((Nn, K), 1.0, device=dev)
            gr = torch.full((Nn, K), 2.0, device=dev)
            _lib.check(env["lib"].bnn_linear_backward_weight_sampled(
                _lib.ptr(xd), M * K, K, _lib.ptr(gyd), M * Nn, Nn, _lib.ptr(rhod), _lib.ptr(gm), _lib.ptr(gr), None, None, None,
                M, Nn, K, S, ctypes.byref(r), None, None, _lib.COMPUTE_F32, 0, acc, _lib.stream_ptr(dev)), "wgrad")
            outs.append((gm, gr))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert torch.allclose(outs[2][0], outs[0][0] + 1.0, rtol=1e-6, atol=1e-6)
        assert torch.allclose(outs[2][1], outs[0][1] + 2.0, rtol=1e-6, atol=1e-6)


def test_full_size_backward_properties(env):
    """BASELINE layer (8 x 512 x 1200 x 1200, bf16 activations): the weight gradient is linear in gy and
    the input gradient of a frozen draw is linear in gy (size-independent properties)."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    dev = env["dev"]
    gen = torch.Generator().manual_seed(11)
    mu = (torch.randn(1200, 1200, generator=gen) * 0.03).to(dev).requires_grad_(True)
    rho = (torch.randn(1200, 1200, generator=gen) * 0.15 - 2).to(dev).requires_grad_(True)
    x = torch.randn(8, 512, 1200, generator=gen).to(dev).bfloat16().requires_grad_(True)
    g1 = torch.randn(8, 512, 1200, generator=gen).to(dev).bfloat16()
    g2 = torch.randn(8, 512, 1200, generator=gen).to(dev).bfloat16()
    key = DrawKey(5, 3, 0, 8, 1)
    y = env["ops"].linear_sampled(x, mu, rho, None, None, key, None, False, compute="bf16", out_dtype=torch.bfloat16)
    ga = torch.autograd.grad(y, (x, mu, rho), g1, retain_graph=True)
    gb = torch.autograd.grad(y, (x, mu, rho), g2, retain_graph=True)
    gs = torch.autograd.grad(y, (x, mu, rho), (g1.float() + g2.float()).bfloat16())
    # g1 + g2 and the bf16 outputs are each rounded once more (2^-9 relative per element): compare
    # in the Frobenius norm at the bf16 bar, and element-wise at 5 % of the output scale
    for a, b, c in zip(ga, gb, gs):
        want = a.float() + b.float()
        assert float((c.float() - want).norm() / want.norm()) < 5e-3
        assert allclose_scaled(N(c.float()), N(want), 5e-2)
    assert all(torch.isfinite(t.float()).all() for t in gs)


# ------------------------------------------------------------------ conv2d backward through the panel
@pytest.mark.parametrize("cfg", [
    # (S, B, C, H, W, O, k, stride, pad, dil, bias, shared_x)
    (2, 3, 8, 6, 6, 16, 3, 2, 1, 1, True, False),
    (3, 2, 8, 7, 5, 24, 3, 1, 1, 2, False, True),
    (2, 4, 64, 6, 6, 64, 3, 2, 1, 1, True, False),       # MNIST / FMNIST conv shape
    # ragged / odd: 1x1 window, 5x5 window, odd channel counts (generic kernel + torch backward), wide rows
    (1, 2, 16, 5, 5, 16, 1, 1, 0, 1, True, False),
    (2, 1, 8, 9, 9, 32, 5, 2, 2, 1, False, True),
    (2, 3, 5, 6, 6, 7, 3, 1, 1, 1, True, False),
    (1, 5, 24, 4, 8, 40, 3, 1, 1, 1, True, True),
    (3, 2, 16, 8, 3, 16, 3, 2, 0, 1, False, False),
])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_sampled_conv2d_backward_vs_float64_autograd(env, cfg, mode):
    """Sampled conv2d backward, all HIP (NCHW -> rows, im2col panel, fused weight / input gradient kernels,
    col2im): against float64 autograd through F.conv2d (conv.py:116) on the oracle's draws."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    S, B, C, H, W, O, k, st, pd, dl, bias, shared = cfg
    orc, dev = env["orc"], env["dev"]
    gen = torch.Generator().manual_seed(17)
    mu = torch.randn(O, C, k, k, generator=gen) * 0.1
    rho = torch.randn(O, C, k, k, generator=gen) * 0.15 - 2.0
    mub = torch.randn(O, generator=gen) * 0.1
    rhob = torch.randn(O, generator=gen) * 0.15 - 2.0
    x = torch.randn((B, C, H, W) if shared else (S, B, C, H, W), generator=gen)
    kw, kb = DrawKey(5, 40, 0, S, 3), DrawKey(5, 41, 0, S, 3)
    xd = x.to(dev).requires_grad_(True)
    md, rd = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
    mbd, rbd = (mub.to(dev).requires_grad_(True), rhob.to(dev).requires_grad_(True)) if bias else (None, None)
    n0 = env["lib"].bnn_launch_count()
    y = env["ops"].conv2d_sampled(xd, md, rd, mbd, rbd, kw, kb if bias else None, shared, (st, st), (pd, pd), (dl, dl), 1, mode)
    gy = torch.randn(y.shape, generator=gen)
    grads = torch.autograd.grad(y, [xd, md, rd] + ([mbd, rbd] if bias else []), gy.to(dev))
    if (C * k * k) % 8 == 0 and C * k * k >= 32 and O >= 16 and O % 8 == 0:
        assert env["lib"].bnn_launch_count() >= n0 + 7       # fwd 2 + rows, im2col, wgrad, dgrad, col2im
    # float64 restatement
    x64 = x.double().requires_grad_(True)
    m64, r64 = mu.double().requires_grad_(True), rho.double().requires_grad_(True)
    mb64, rb64 = mub.double().requires_grad_(True), rhob.double().requires_grad_(True)
    ys = []
    for s_ in range(S):
        ew = torch.from_numpy(orc.eps_fill(kw.seed, kw.stream, s_, kw.epoch_host, 0, tuple(mu.shape), kw.gen)).double()
        w = m64 + (1e-10 + torch.nn.functional.softplus(r64)) * ew
        b_ = None
        if bias:
            eb = torch.from_numpy(orc.eps_fill(kb.seed, kb.stream, s_, kb.epoch_host, 0, (O,), kb.gen)).double()
            b_ = mb64 + (1e-10 + torch.nn.functional.softplus(rb64)) * eb
        ys.append(torch.nn.functional.conv2d(x64 if shared else x64[s_], w, b_, st, pd, dl))
    y64 = torch.stack(ys)
    tol_y = 1e-5 if mode == "f32" else 2e-2
    assert allclose_scaled(N(y), y64.detach().numpy(), tol_y)
    want = torch.autograd.grad(y64, [x64, m64, r64] + ([mb64, rb64] if bias else []), gy.double())
    tol = 2e-5 if mode == "f32" else 3e-2
    for got, w_ in zip(grads, want):
        assert got.shape == w_.shape
        assert allclose_scaled(N(got), w_.numpy(), tol)


@pytest.mark.parametrize("shared", [False, True])
def test_plain_conv2d_backward_panel_vs_float64_autograd(env, shared):
    """Explicit-weight conv2d (parity mode, Flipout): backward through the panel -- rows, im2col, per-sample
    weight gradient, plain input gradient, col2im, column sums -- against float64 autograd of F.conv2d."""
    dev = env["dev"]
    gen = torch.Generator().manual_seed(23)
    S, B, C, H, W, O, k, st, pd = 2, 3, 8, 7, 6, 12, 3, 2, 1
    w = torch.randn(S, O, C, k, k, generator=gen) * 0.2
    b = torch.randn(S, O, generator=gen) * 0.1
    x = torch.randn((B, C, H, W) if shared else (S, B, C, H, W), generator=gen)
    xd, wd, bd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    n0 = env["lib"].bnn_launch_count()
    y = env["ops"].conv2d_plain(xd, wd, bd, shared, (st, st), (pd, pd), (1, 1), 1, "f32")
    gy = torch.randn(y.shape, generator=gen)
    gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), gy.to(dev))
    assert env["lib"].bnn_launch_count() >= n0 + 6
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = torch.stack([torch.nn.functional.conv2d(x64 if shared else x64[s_], w64[s_], b64[s_], st, pd) for s_ in range(S)])
    assert allclose_scaled(N(y), y64.detach().numpy())
    want = torch.autograd.grad(y64, (x64, w64, b64), gy.double())
    for got, w_ in zip((gx, gw, gb), want):
        assert got.shape == w_.shape and allclose_scaled(N(got), w_.numpy(), 2e-5)


def _random_linear_shapes():
    rng = np.random.RandomState(1234)
    shapes = []
    for _ in range(28):
        S = int(rng.randint(1, 5))
        M = int(rng.choice([1, 2, 7, 16, 31, 33, 64, 65, 130]))
        Nn = int(rng.choice([1, 2, 9, 15, 16, 17, 24, 31, 47, 48, 49, 80, 97]))
        K = int(rng.choice([1, 3, 4, 5, 8, 12, 20, 32, 36, 60, 64, 68, 100, 128, 132, 200]))
        shapes.append((S, M, Nn, K, bool(rng.randint(0, 2)), bool(rng.randint(0, 2))))
    return shapes


@pytest.mark.parametrize("S,M,Nn,K,bias,shared", _random_linear_shapes())
def test_random_shapes_forward_and_backward_vs_oracle(env, S, M, Nn, K, bias, shared):
    """Seeded sweep over ragged / tiny / unaligned shapes (every dispatch path: draw-paced tiles, narrow head,
    generic kernel; fused and explicit-weight input gradient; tile, sample-split and narrow weight gradient):
    exact-fp32 forward against oracle.linear on the oracle's draws, backward against oracle.linear_bwd."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    orc, dev = env["orc"], env["dev"]
    gen = torch.Generator().manual_seed(S * 1000003 + M * 1009 + Nn * 31 + K)
    mu = torch.randn(Nn, K, generator=gen) * 0.2
    rho = torch.randn(Nn, K, generator=gen) * 0.3 - 2.0
    mub = torch.randn(Nn, generator=gen) * 0.2
    rhob = torch.randn(Nn, generator=gen) * 0.3 - 2.0
    x = torch.randn((M, K) if shared else (S, M, K), generator=gen)
    gy = torch.randn(S, M, Nn, generator=gen)
    kw, kb = DrawKey(11, 3, 2, S, 5), DrawKey(11, 4, 2, S, 5)
    xd = x.to(dev).requires_grad_(True)
    md, rd = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
    mbd, rbd = (mub.to(dev).requires_grad_(True), rhob.to(dev).requires_grad_(True)) if bias else (None, None)
    y = env["ops"].linear_sampled(xd, md, rd, mbd, rbd, kw, kb if bias else None, shared)
    eps = [orc.eps_fill(kw.seed, kw.stream, 2 + s_, kw.epoch_host, 0, (Nn, K), kw.gen) for s_ in range(S)]
    epsb = [orc.eps_fill(kb.seed, kb.stream, 2 + s_, kb.epoch_host, 0, (Nn,), kb.gen) for s_ in range(S)]
    for s_ in range(S):
        w = orc.sample_affine(mu.numpy(), rho.numpy(), eps[s_])
        b_ = orc.sample_affine(mub.numpy(), rhob.numpy(), epsb[s_]) if bias else None
        assert allclose_scaled(N(y[s_]), orc.linear(N(x if shared else x[s_]), w, b_)), (s_,)
    grads = torch.autograd.grad(y, [xd, md, rd] + ([mbd, rbd] if bias else []), gy.to(dev))
    want_mu, want_rho, want_x = orc.linear_bwd(mu.numpy(), rho.numpy(), x.numpy(), gy.numpy(), eps, shared)
    assert allclose_scaled(N(grads[0]), want_x.sum(0) if shared else want_x, 2e-5)
    assert allclose_scaled(N(grads[1]), want_mu, 2e-5)
    assert allclose_scaled(N(grads[2]), want_rho, 2e-5)
    if bias:
        gmb = np.zeros(Nn)
        grb = np.zeros(Nn)
        for s_ in range(S):
            cs = N(gy[s_]).astype(np.float64).sum(0)
            a, b_ = orc.sample_affine_bwd(np.ones(Nn, np.float32), rhob.numpy(), epsb[s_])
            gmb += cs * a
            grb += cs * b_
        assert allclose_scaled(N(grads[3]), gmb, 2e-5) and allclose_scaled(N(grads[4]), grb, 2e-5)


@pytest.mark.parametrize("S,M,Nn,K,bias,shared", _random_linear_shapes()[:14])
@pytest.mark.parametrize("acts", ["f32", "bf16"])
def test_random_shapes_bf16_mode_vs_exact_oracle(env, S, M, Nn, K, bias, shared, acts):
    """Same sweep in bf16 compute mode (bf16 operands, fp32 accumulate; hidden activations fp32 or bf16 where the
    shape allows): forward and backward within 3e-2 of the output scale of the EXACT oracle."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    orc, dev = env["orc"], env["dev"]
    if acts == "bf16" and K % 8 != 0:
        pytest.skip("bf16 activations need K % 8 == 0 (BNN_FLAG_X_BF16)")
    gen = torch.Generator().manual_seed(S * 1000003 + M * 1009 + Nn * 31 + K + 7)
    mu = torch.randn(Nn, K, generator=gen) * 0.2
    rho = torch.randn(Nn, K, generator=gen) * 0.3 - 2.0
    x = torch.randn((M, K) if shared else (S, M, K), generator=gen)
    gy = torch.randn(S, M, Nn, generator=gen)
    kw = DrawKey(12, 3, 0, S, 6)
    adt = torch.bfloat16 if acts == "bf16" else torch.float32
    xr = x.to(adt).float()                                   # the operand the kernel really sees
    xd = x.to(dev).to(adt).requires_grad_(True)
    md, rd = mu.to(dev).requires_grad_(True), rho.to(dev).requires_grad_(True)
    y = env["ops"].linear_sampled(xd, md, rd, None, None, kw, None, shared, compute="bf16", out_dtype=adt)
    eps = [orc.eps_fill(kw.seed, kw.stream, s_, kw.epoch_host, 0, (Nn, K), kw.gen) for s_ in range(S)]
    for s_ in range(S):
        w = orc.sample_affine(mu.numpy(), rho.numpy(), eps[s_])
        assert allclose_scaled(N(y[s_].float()), orc.linear(N(xr if shared else xr[s_]), w, None), 3e-2)
    gx, g_mu, g_rho = torch.autograd.grad(y, (xd, md, rd), gy.to(dev).to(adt))
    gyr = gy.to(adt).float()
    want_mu, want_rho, want_x = orc.linear_bwd(mu.numpy(), rho.numpy(), xr.numpy(), gyr.numpy(), eps, shared)
    assert allclose_scaled(N(gx.float()), want_x.sum(0) if shared else want_x, 3e-2)
    assert allclose_scaled(N(g_mu), want_mu, 3e-2) and allclose_scaled(N(g_rho), want_rho, 3e-2)


def test_kl_gradient_fusion_is_opt_in_and_equivalent(env):
    """nn.fuse_kl_gradient: off (default) KLDivergence's backward returns its own gradient (torch.autograd.grad
    works); on, the layers' weight-gradient launches add it -- same .grad after (likelihood + kl).backward();
    a layer that takes no part in the backward still gets its KL gradient (flush)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, KLDivergence, fuse_kl_gradient
    dev = env["dev"]

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(24, 5, 3)
            self.layers = torch.nn.Sequential(NormalLinear(24, 48), torch.nn.ReLU(), NormalLinear(48, 5))
            self.unused = NormalLinear(8, 8)                     # only the KL term sees it

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(3)
    net = Net().to(dev)
    net.mc_batched = True
    x = torch.randn(16, 24, device=dev)
    gy = torch.randn(3, 16, 5, device=dev)
    kld = KLDivergence(number_of_batches=7)
    # default path: functional gradient of the KL term alone
    g = torch.autograd.grad(kld(net), list(net.parameters()))
    assert all(t is not None and torch.isfinite(t).all() for t in g)
    results = []
    for fuse in (False, True):
        fuse_kl_gradient(fuse)
        try:
            env["bnn"].manual_seed(9)
            for p_ in net.parameters():
                p_.grad = None
            n0 = env["lib"].bnn_launch_count()
            ys = net.forward_stacked(x, 3)
            ((ys * gy).sum() + 2.0 * kld(net)).backward()
            results.append(([p_.grad.clone() for p_ in net.parameters()], env["lib"].bnn_launch_count() - n0))
        finally:
            fuse_kl_gradient(False)
    (g0, l0), (g1, l1) = results
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert l0 > 0 and l1 > 0          # (the saving is autograd's per-parameter accumulation adds, which are torch launches)
    from bayesianneuralnetworks_amd import ops
    assert not ops._tls.kl_pending


# ------------------------------------------------------------------ pruning
def test_prune_normal_on_device(env):
    """prune/prune.py:7-22 on the device: HIP score == oracle; half of every tensor pruned to (0, -30); the
    pruned model still runs and its state_dict round-trips (examples/MNIST/prune.py:47-58)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, KLDivergence
    from bayesianneuralnetworks_amd.prune import PruneNormal
    dev, orc = env["dev"], env["orc"]
    gen = torch.Generator().manual_seed(12)
    mu = torch.randn(257, 33, generator=gen) * 0.3
    rho = torch.randn(257, 33, generator=gen) * 2 - 2
    rho[0, :4] = torch.tensor([-30.0, 25.0, 19.9, 0.0])
    got = N(env["ops"].prune_score(mu.to(dev), rho.to(dev)))
    want = orc.prune_score(mu.numpy(), rho.numpy())
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5)

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(12, 5, 2)
            self.layers = torch.nn.Sequential(NormalLinear(12, 40), torch.nn.ReLU(), NormalLinear(40, 5))

        def _forward(self, x):
            return self.layers(x)

    net = Net().to(dev)
    n0 = env["lib"].bnn_launch_count()
    PruneNormal()(net, 0.5)
    assert env["lib"].bnn_launch_count() >= n0 + 4           # one score launch per posterior tensor
    for L in (net.layers[0], net.layers[2]):
        frac = (L.weight.scale == -30).float().mean().item()
        assert abs(frac - 0.5) < 0.02
        assert bool(((L.weight.scale == -30) == (L.weight.mean == 0)).all())
    y = net(torch.randn(7, 12, device=dev))
    assert len(y) == 2 and torch.isfinite(y[0]).all()
    assert torch.isfinite(KLDivergence()(net))
    net2 = Net().to(dev)
    net2.load_state_dict(net.state_dict())
    assert torch.equal(net2.layers[0].weight.scale, net.layers[0].weight.scale)


# ------------------------------------------------------------------ training-loop callers
def test_fused_adam_matches_torch_adam(env):
    """optim.Adam (one HIP launch for all tensors) against torch.optim.Adam (train.py:41,65) over 6 steps,
    ragged sizes, with and without weight decay."""
    from bayesianneuralnetworks_amd import optim
    dev = env["dev"]
    for wd in (0.0, 0.01):
        gen = torch.Generator().manual_seed(8)
        shapes = [(1,), (3, 5), (2049,), (1200, 37), (7,)]
        ref = [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
        mine = [r.detach().clone().requires_grad_(True) for r in ref]
        o_ref = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=wd)
        o_mine = optim.Adam(mine, lr=3e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=wd)
        for step in range(6):
            for r, m in zip(ref, mine):
                g = torch.randn(r.shape, generator=gen).to(dev)
                r.grad, m.grad = g.clone(), g.clone()
            o_ref.step()
            o_mine.step()
            for r, m in zip(ref, mine):
                assert torch.allclose(r, m, rtol=1e-5, atol=1e-6), (wd, step)
        assert float(o_mine.param_groups[0]["step"]) == 6.0


def test_softmax_cross_entropy_matches_torch(env):
    dev = env["dev"]
    gen = torch.Generator().manual_seed(9)
    for R, C in ((1, 2), (300, 10), (4096, 10), (257, 33), (8192, 10), (9000, 7)):      # (<= 8192 rows: the one-launch kernel)
        x = (torch.randn(R, C, generator=gen) * 4).to(dev).requires_grad_(True)
        y = torch.randint(0, C, (R,), generator=gen).to(dev)
        xr = x.detach().clone().requires_grad_(True)
        l_ref = torch.nn.functional.cross_entropy(xr, y)
        l = env["ops"].cross_entropy(x, y)
        (l * 1.5).backward()
        (l_ref * 1.5).backward()
        assert abs(l.item() - l_ref.item()) <= 1e-5 * (1 + abs(l_ref.item()))
        assert torch.allclose(x.grad, xr.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_training_loop_learns_a_separable_task(env, mode):
    """The reference's loop body (examples/MNIST/train.py:53-65) end to end on the device -- MC-batched
    forward, fused ReLU / bf16 activations, KLDivergence, HIP cross-entropy, HIP backward with the KL
    gradient folded into the weight-gradient launch, HIP Adam: a linearly separable 3-class task is
    learnt (loss falls, accuracy > 95 %), and the posterior scales move (the KL / draw-backward path)."""
    from bayesianneuralnetworks_amd import optim
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, KLDivergence, fuse_activations
    dev = env["dev"]

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(16, 3, 4)
            self.layers = torch.nn.Sequential(NormalLinear(16, 64), torch.nn.ReLU(), NormalLinear(64, 3))

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(0)
    env["bnn"].manual_seed(0)
    env["bnn"].set_compute(mode)
    try:
        net = Net().to(dev)
        net.mc_batched = True
        fuse_activations(net, bf16_activations=(mode == "bf16"))
        gen = torch.Generator().manual_seed(4)
        centers = torch.randn(3, 16, generator=gen) * 3
        y = torch.randint(0, 3, (512,), generator=gen)
        x = (centers[y] + torch.randn(512, 16, generator=gen)).to(dev)
        y = y.to(dev)
        kld = KLDivergence(number_of_batches=50)
        opt = optim.Adam(net.parameters(), lr=5e-3)
        rho0 = net.layers[0].weight.scale.detach().clone()
        losses = []
        for step in range(120):
            opt.zero_grad(set_to_none=True)
            ys = net.forward_stacked(x, 4)                                  # (4, 512, 3)
            loss = env["ops"].cross_entropy(ys.reshape(4 * 512, 3).float(), y.repeat(4)) + kld(net)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        with torch.no_grad():
            pred = net.forward_stacked(x, 4).float().mean(0).argmax(-1)
        assert losses[-1] < 0.5 * losses[0]
        assert (pred == y).float().mean().item() > 0.95
        assert (net.layers[0].weight.scale.detach() - rho0).abs().max().item() > 1e-3
        assert all(torch.isfinite(p).all() for p in net.parameters())
    finally:
        env["bnn"].set_compute("f32")


def test_empty_batch_and_single_row_edges(env):
    """Empty batch (M = 0): nothing is launched for the contraction, outputs are empty, parameter gradients are
    exact zeros.  One row, one column, K = 4 (the smallest aligned layer): fused path == oracle."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    dev, orc = env["dev"], env["orc"]
    mu = torch.randn(8, 12).to(dev).requires_grad_(True)
    rho = (torch.randn(8, 12) * 0.1 - 2).to(dev).requires_grad_(True)
    key = DrawKey(3, 9, 0, 2, 1)
    x0 = torch.zeros(2, 0, 12, device=dev, requires_grad=True)
    y0 = env["ops"].linear_sampled(x0, mu, rho, None, None, key, None, False)
    assert y0.shape == (2, 0, 8)
    gx, gm, gr = torch.autograd.grad(y0, (x0, mu, rho), torch.zeros_like(y0))
    assert gx.shape == x0.shape and float(gm.abs().max()) == 0.0 and float(gr.abs().max()) == 0.0
    for (Nn, K) in ((1, 4), (1, 8), (3, 4)):
        m1 = torch.randn(Nn, K) * 0.1
        r1 = torch.randn(Nn, K) * 0.1 - 2
        k1 = DrawKey(4, 2, 5, 1, 0)
        xx = torch.randn(1, 1, K)
        got = env["ops"].linear_sampled(xx.to(dev), m1.to(dev), r1.to(dev), None, None, k1, None, False)
        w = orc.sample_affine(m1.numpy(), r1.numpy(), orc.eps_fill(k1.seed, k1.stream, 5, k1.epoch_host, 0, (Nn, K), k1.gen))
        assert allclose(N(got)[0], orc.linear(xx[0].numpy(), w))


# ------------------------------------------------------------------ properties at full size
def test_full_size_linearity_property(env):
    """BASELINE size (512 x 1200 x 1200): with the draw frozen (sample=False) and no bias the
    layer is linear in x: f(a x1 + x2) = a f(x1) + f(x2) up to fp32 rounding."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    dev = env["dev"]
    torch.manual_seed(2)
    layer = NormalLinear(1200, 1200, False).to(dev)
    x1 = torch.randn(512, 1200, device=dev)
    x2 = torch.randn(512, 1200, device=dev)
    y1 = layer(x1)
    y2 = layer(x2, sample=False)
    y12 = layer(2.0 * x1 + x2, sample=False)
    assert allclose_scaled(N(y12), N(2.0 * y1 + y2))


def test_kl_error_and_priors(env):
    """loss.py:34-36: ValueError without Bayesian modules; tests/test_nn/test_loss.py:23-34."""
    from bayesianneuralnetworks_amd.nn import KLDivergence, BayesianNetworkModule, NormalLinear

    class Net(BayesianNetworkModule):
        def __init__(self, arch):
            super().__init__(3, 4, 1)
            self.arch = arch

        def _forward(self, x):
            return self.arch(x)

    dev = env["dev"]
    with pytest.raises(ValueError):
        KLDivergence()(Net(torch.nn.Linear(3, 4)).to(dev))
    net = Net(torch.nn.Sequential(torch.nn.Linear(3, 4), NormalLinear(4, 2), torch.nn.Linear(2, 1))).to(dev)
    r = KLDivergence()(net)
    assert isinstance(r, torch.Tensor) and r.is_cuda and r > 0


def test_fused_relu_matches_separate_relu(env):
    """nn.fuse_activations: ReLU in the kernel epilogue == a ReLU module after the layer,
    forward and backward (same draw)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, fuse_activations
    dev = env["dev"]

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(24, 8, 3)
            self.layers = torch.nn.Sequential(NormalLinear(24, 40), torch.nn.ReLU(), NormalLinear(40, 8))

        def _forward(self, x):
            return self.layers(x)

    torch.manual_seed(0)
    net = Net().to(dev)
    net.mc_batched = True
    x = torch.randn(10, 24, device=dev, requires_grad=True)
    env["bnn"].manual_seed(9)
    ya = net.forward_stacked(x)
    ya.square().sum().backward()
    ga = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
    for p in net.parameters():
        p.grad = None
    x.grad = None
    assert fuse_activations(net) == 1 and isinstance(net.layers[1], torch.nn.Identity)
    env["bnn"].manual_seed(9)
    yb = net.forward_stacked(x)
    yb.square().sum().backward()
    gb = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
    assert torch.equal(ya, yb)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_bf16_hidden_activations_are_result_neutral(env):
    """bf16 compute mode: emitting hidden activations in bf16 (fuse_activations(...,
    bf16_activations=True)) feeds the consumer the SAME bf16 operand values as fp32 hidden
    activations would (it rounds its A operand to bf16 either way); only the k order inside the
    MFMA differs between the two A layouts, i.e. fp32 summation order: 1e-5 of the output scale."""
    from bayesianneuralnetworks_amd.nn import fuse_activations
    dev = env["dev"]
    net, _ = _MLP.build(dev, dims=(96, 200, 120, 10), samples=4, seed=3)
    net.mc_batched = True
    fuse_activations(net)
    x = torch.randn(40, 96, device=dev)
    env["bnn"].set_compute("bf16")
    try:
        env["bnn"].manual_seed(11)
        ya = net.forward_stacked(x).clone()
        fuse_activations(net, bf16_activations=True)
        assert net.layers[0].out_dtype == torch.bfloat16 and net.layers[2].out_dtype == torch.bfloat16
        assert net.layers[4].out_dtype is None
        env["bnn"].manual_seed(11)
        yb = net.forward_stacked(x)
        assert yb.dtype == torch.float32
        assert allclose_scaled(N(yb), N(ya))
        # and in fp32 mode the attribute is ignored (exact path untouched)
        env["bnn"].set_compute("f32")
        env["bnn"].manual_seed(11)
        yc = net.forward_stacked(x)
        assert yc.dtype == torch.float32 and not torch.equal(yc, yb)
    finally:
        env["bnn"].set_compute("f32")


def test_config5_wide_layer_properties(env):
    """BASELINE configs[4] shape (4096 x 4096 layer, batch 4096, fp32, MFMA-bound): too big for the
    scalar oracle, so size-independent properties: (1) linearity in x with the draw frozen,
    (2) a 64-row slice of the batch equals the oracle on that slice, (3) MC-batched samples equal
    the same samples drawn one at a time.
    Tolerance for (1), (2): 5e-5 of the output scale instead of 1e-5 -- at K = 4096 the fp32 MFMA
    is a 4096-long k-ordered fmaf chain whose rounding error is 3.5e-7 * sum|a b| ~ 1e-4 here
    (MI355X_MICROARCH.md, F32 MFMA numerics; SURVEY.md 7 "borderline at K=4096")."""
    from bayesianneuralnetworks_amd.nn import NormalLinear
    from bayesianneuralnetworks_amd import _mc
    dev = env["dev"]
    torch.manual_seed(4)
    layer = NormalLinear(4096, 4096, True).to(dev)
    env["bnn"].manual_seed(21)
    x1 = torch.randn(4096, 4096, device=dev)
    x2 = torch.randn(4096, 4096, device=dev)
    y1 = layer(x1)
    y2 = layer(x2, sample=False)
    y12 = layer(0.5 * x1 - x2, sample=False)
    b = layer.sampled[1]
    assert allclose_scaled(N(y12 - b), N(0.5 * (y1 - b) - (y2 - b)), tol=5e-5)
    w, bo = _oracle_layer_draw(env["orc"], layer, 0)
    want = env["orc"].linear(N(x1[1000:1064]), w, bo)
    assert allclose_scaled(N(y1[1000:1064]), want, tol=5e-5)
    # two samples in one launch == the same two sample ids launched separately
    key = layer.weight.draw_key
    with _mc.McContext(2, 4096, sample0=0):
        layer.weight.sample(2, 0, key.epoch_host)
        layer.bias.sample(2, 0, key.epoch_host)
        yb = layer(x1, sample=False)
    assert torch.equal(yb[:4096], y1)
    with _mc.McContext(1, 4096, sample0=1):
        layer.weight.sample(1, 1, key.epoch_host)
        layer.bias.sample(1, 1, key.epoch_host)
        ys1 = layer(x1, sample=False)
    assert torch.equal(yb[4096:], ys1)


def test_config5_wide_layer_no_grad_takes_the_three_plane_dense_path(env):
    """The same configs[4] layer under torch.no_grad(): the fp32 parity mode then runs draw-once + bnn_dense_forward_x3 on
    the 256 x 128 tile (k_dense_bf16<4, 8, 4, 1, 3> on three-plane operands) -- the launch bench.py's roofline_wide_f32 leg
    times.  (1) rows from every kind of tile equal the oracle in double on the Philox draw of the recorded key at 1e-5 of the
    output scale (the bf16x3 contraction has no 4096-long fp32 chain: the tighter bar holds), (2) linearity with the draw
    frozen, (3) it agrees with the grad-mode (fused-kernel) result of the same key."""
    from conftest import assert_close_scaled
    from bayesianneuralnetworks_amd.nn import NormalLinear
    from bayesianneuralnetworks_amd import ops
    dev, lib = env["dev"], env["lib"]
    assert ops.DENSE_X3_F32
    torch.manual_seed(4)
    layer = NormalLinear(4096, 4096, True).to(dev)
    env["bnn"].manual_seed(21)
    x1 = torch.randn(4096, 4096, device=dev)
    x2 = torch.randn(4096, 4096, device=dev)
    with torch.no_grad():
        n0 = lib.bnn_launch_count()
        y1 = layer(x1)
        assert lib.bnn_launch_count() == n0 + 3, "draw + split of the input + one dense launch expected"
        y2 = layer(x2, sample=False)
        y12 = layer(0.5 * x1 - x2, sample=False)
    b = layer.sampled[1]
    assert_close_scaled(N(y12 - b), N(0.5 * (y1 - b) - (y2 - b)), 2e-5, "linearity")
    w, bo = _oracle_layer_draw(env["orc"], layer, 0)
    rows = torch.cat([torch.arange(r, r + 16) for r in (0, 240, 256, 1000, 2040, 2176, 3824, 4080)])
    want = N(x1[rows.to(dev)]).astype(np.float64) @ w.astype(np.float64).T + bo.astype(np.float64)
    assert_close_scaled(N(y1[rows.to(dev)]), want, 1e-5, "x3 dense path at 4096 vs oracle")
    yg = layer(x1, sample=False)                       # grad mode: the fused kernel on the same key
    assert_close_scaled(N(yg[rows.to(dev)]), want, 5e-5, "fused kernel at 4096 vs oracle")


def test_config4_cifar_conv_mc_batched(env):
    """BASELINE configs[3] shape: NormalConv2d(128, 128, 3, padding=1) on (256, 128, 4, 4), 8 MC
    samples in one launch; sample s of the batched launch == oracle conv on a batch slice with the
    Philox draw of sample s."""
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from bayesianneuralnetworks_amd import _mc
    dev = env["dev"]
    torch.manual_seed(6)
    layer = NormalConv2d(128, 128, 3, padding=1).to(dev)
    env["bnn"].manual_seed(8)
    x = torch.randn(256, 128, 4, 4, device=dev)
    with _mc.McContext(8, 256, sample0=0):
        y = layer(x)                                   # shared input, 8 samples -> (8*256, 128, 4, 4)
    assert y.shape == (8 * 256, 128, 4, 4)
    for s in (0, 7):
        w, b = _oracle_layer_draw(env["orc"], layer, s)
        want = env["orc"].conv2d(N(x[10:14]), w, b, (1, 1), (1, 1), (1, 1), 1)
        assert allclose(N(y[s * 256 + 10:s * 256 + 14]), want)
