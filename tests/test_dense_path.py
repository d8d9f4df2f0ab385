"""GPU parity of the draw-once path (csrc/bnn_dense.hip): bnn_draw_multi and bnn_dense_forward through the C-ABI
against the CPU oracle on the same Philox draws, and against the fused kernel / the standalone sampler.

bf16 compute mode: the oracle is fed what the MFMA is fed (bf16-rounded activations and drawn weights, fp32 bias) and
accumulates in double; the kernel accumulates the exact bf16 products in fp32 -> 1e-5 of the output scale
(conftest.allclose_scaled), the same bar as the fp32 path."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import allclose_scaled, assert_close_scaled
import seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd import _lib, ops
    from oracle import oracle as orc
    return dict(bnn=bnn, lib=_lib.load(), _lib=_lib, ops=ops, orc=orc, dev=torch.device("cuda:0"))


def N(t):
    return t.detach().float().cpu().numpy()


def _post(shape, seed, dev, bias=True):
    gen = torch.Generator().manual_seed(seed)
    mw, rw, mb, rb = seeded.posterior(gen, shape, bias)
    return [t.to(dev) if t is not None else None for t in (mw, rw, mb, rb)]


@pytest.mark.parametrize("gen", [0, 1])
@pytest.mark.parametrize("shape", [(1200, 784), (10, 1200), (64, 48), (7, 16), (80, 264), (600, 1208)])
@pytest.mark.parametrize("S", [1, 8])
def test_draw_multi_equals_standalone_sampler_and_oracle(env, shape, S, gen):
    """(600 x 1208: 4530 groups -- past the small-tensor threshold, the one-thread-per-group path with a ragged row end.)"""
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, orc, dev = env["ops"], env["orc"], env["dev"]
    mw, rw, mb, rb = _post(shape, 5, dev)
    kw, kb = DrawKey(99, 11, 2, S, 7, gen=gen), DrawKey(99, 12, 2, S, 7, gen=gen)
    n0 = env["lib"].bnn_launch_count()
    pre = ops.draw_layers([(mw, rw, mb, rb, kw, kb)], S)[0]
    assert env["lib"].bnn_launch_count() == n0 + 1            # weight AND bias, all samples: one launch
    Nn, K = shape
    kp = (K + 63) // 64 * 64
    assert pre.w.shape == (S, Nn, kp) and pre.b.shape == (S, Nn)
    # (a) bit-identical to K1 (same device function, same key), zero padding
    w1 = ops._sample_affine_philox_raw(mw, rw, kw, out_dtype=torch.bfloat16)
    assert torch.equal(pre.w[:, :, :K], w1)
    assert (pre.w[:, :, K:] == 0).all()
    assert torch.equal(pre.b, ops._sample_affine_philox_raw(mb, rb, kb))
    # (b) the oracle's draw, rounded to bf16: at most one bf16 ulp apart (the eps twin agrees to ~1e-6)
    for s in range(S):
        ew = orc.eps_fill(kw.seed, kw.stream, kw.sample0 + s, kw.epoch_host, 0, shape, kw.gen)
        want = orc.sample_affine(N(mw), N(rw), ew)
        got = N(pre.w[s, :, :K])
        assert (np.abs(got - want) <= np.abs(want) * 2.0 ** -8 + 1e-6).all()
        eb = orc.eps_fill(kb.seed, kb.stream, kb.sample0 + s, kb.epoch_host, 0, (Nn,), kb.gen)
        assert np.allclose(N(pre.b[s]), orc.sample_affine(N(mb), N(rb), eb), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("O,C,k,S,gen,x3", [(64, 64, 3, 8, 1, False), (128, 128, 3, 3, 0, True), (16, 8, 5, 2, 1, False), (5, 16, 7, 2, 0, False),
                                               (24, 40, 2, 3, 1, True), (3, 8, 1, 2, 0, False), (10, 12, 3, 2, 1, False)])
def test_conv_weight_draw_tap_major(env, O, C, k, S, gen, x3):
    """A conv weight (O, C, k, k) drawn tap-major (element (o, c, t) -> column t C + c; C % 8 == 0: groups of 8 channels x taps through
    LDS, 16-B chunks; C = 12: the general body): the values of K1 on the same key, permuted -- bit for bit --, zero padding, for
    bf16 and three-plane outputs and both eps streams."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, dev = env["ops"], env["dev"]
    g = torch.Generator().manual_seed(O + C + k)
    mw = (torch.randn(O, C, k, k, generator=g) * 0.1).to(dev)
    rw = (torch.randn(O, C, k, k, generator=g) * 0.2 - 2.0).to(dev)
    kw = DrawKey(31, 7, 1, S, 4, gen=gen)
    K, taps = C * k * k, k * k
    pre = ops.draw_layers([(mw.reshape(O, K), rw.reshape(O, K), None, None, kw, None, taps)], S, x3=x3)[0]
    kp = (K + 63) // 64 * 64
    w = ops._sample_affine_philox_raw(mw, rw, kw)                                  # (S, O, C, k, k) fp32, the same key
    want = w.reshape(S, O, C, taps).transpose(2, 3).reshape(S, O, K)              # tap-major
    if x3:
        assert pre.w.shape == (3, S, O, kp)
        h = want.bfloat16()
        m = (want - h.float()).bfloat16()
        l = (want - h.float() - m.float()).bfloat16()
        for got, ref in zip(pre.w, (h, m, l)):
            assert torch.equal(got[:, :, :K], ref)
            assert not got[:, :, K:].any()
    else:
        assert pre.w.shape == (S, O, kp)
        assert torch.equal(pre.w[:, :, :K], want.bfloat16())
        assert not pre.w[:, :, K:].any()


def test_draw_multi_many_layers_and_kl_carry(env):
    """Three layers (6 tensors) + the KL first pass in ONE launch; the KL finished by mc_mean equals kl_normal's."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, dev = env["ops"], env["dev"]
    posts = [_post(sh, 20 + i, dev) for i, sh in enumerate([(96, 64), (40, 96), (10, 40)])]
    layers, mus, rhos = [], [], []
    for i, (mw, rw, mb, rb) in enumerate(posts):
        layers.append((mw, rw, mb, rb, DrawKey(5, 2 * i + 1, 0, 4, 3), DrawKey(5, 2 * i + 2, 0, 4, 3)))
        mus += [mw, mb]
        rhos += [rw, rb]
    priors = [(0.0, 0.1)] * 6
    ref = ops.kl_normal(mus, rhos, priors, 2.0)
    h = ops.kl_normal_begin(mus, rhos, priors, 2.0, carry=True)
    n0 = env["lib"].bnn_launch_count()
    pre = ops.draw_layers(layers, 4, kl=h)
    assert env["lib"].bnn_launch_count() == n0 + 1 and h.launched
    ops._tls.kl_carry = None
    ys = torch.zeros(4, 8, device=dev)
    ops.mc_mean(ys, kl=h)
    assert torch.equal(h.out, ref)
    for (mw, rw, mb, rb, kw, kb), p_ in zip(layers, pre):
        assert torch.equal(p_.w[:, :, :mw.shape[1]], ops._sample_affine_philox_raw(mw, rw, kw, out_dtype=torch.bfloat16))
        assert torch.equal(p_.b, ops._sample_affine_philox_raw(mb, rb, kb))


DENSE_SHAPES = [  # S, M, N, K, shared_x, relu, y_bf16
    (2, 17, 80, 72, False, False, False),
    (3, 130, 96, 264, False, True, False),        # ragged rows and columns, K tail (264 = 4 x 64 + 8)
    (2, 300, 200, 128, True, False, True),
    (1, 256, 160, 64, False, False, False),
    (8, 512, 1200, 784, True, True, True),        # BASELINE layer 1 (shared input, ReLU, bf16 hidden activation)
    (8, 512, 1200, 1200, False, True, True),      # BASELINE layer 2
    (8, 512, 10, 1200, False, False, False),      # BASELINE head
    (2, 33, 5, 40, False, False, False),
    (3, 40, 16, 2048, True, True, False),
    (1, 1, 10, 8, False, False, False),
    (2, 260, 256, 320, False, False, False),      # 128-column tiles
    (4, 512, 1200, 1200, False, True, True),      # a rank of a 2-GPU job: 64 x 160 tiles
    (2, 512, 1200, 784, True, True, True),        # ... of a 4-GPU job: 32 x 160 tiles
    (1, 512, 1200, 1200, False, True, False),     # ... of an 8-GPU job
    (1, 100, 160, 72, True, False, False),        # 64-row tiles with a ragged last tile
    (3, 1100, 1300, 136, False, True, False),     # S % 8 != 0, 55 tiles of 256 x 128 per sample: XCD-contiguous grouped tile order, ragged
    (8, 700, 1300, 72, True, False, True),        # S % 8 == 0, 33 tiles per sample: sample -> XCD with grouped tiles inside
    (1, 48, 240, 200, False, True, True),         # 32-row tiles, ragged rows
]


@pytest.mark.parametrize("S,M,Nn,K,shared,relu,ybf", DENSE_SHAPES)
def test_dense_forward_vs_oracle(env, S, M, Nn, K, shared, relu, ybf):
    """bnn_dense_forward (through the C-ABI) on given bf16 operands against the oracle's F.linear (double accumulate)."""
    lib, _lib, orc, dev = env["lib"], env["_lib"], env["orc"], env["dev"]
    g = torch.Generator().manual_seed(S * 1000 + M + Nn + K)
    kp = (K + 63) // 64 * 64
    x = torch.randn((M, K) if shared else (S, M, K), generator=g).bfloat16()
    w = torch.zeros(S, Nn, kp, dtype=torch.bfloat16)
    w[:, :, :K] = (torch.randn(S, Nn, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(S, Nn, generator=g) * 0.1
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = torch.full((S, M, Nn), float("nan"), dtype=torch.bfloat16 if ybf else torch.float32, device=dev)
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_Y_BF16 if ybf else 0)
    rc = lib.bnn_dense_forward(_lib.ptr(xd), 0 if shared else M * K, K, _lib.ptr(wd), Nn * kp, kp, _lib.ptr(bd), Nn,
                               _lib.ptr(y), M * Nn, Nn, M, Nn, K, S, flags, _lib.stream_ptr(dev))
    assert rc == 0, lib.bnn_last_error()
    torch.cuda.synchronize()
    got = N(y)
    xf, wf = x.float().numpy(), w.float().numpy()[:, :, :K]
    for s in range(S):
        xs = xf if shared else xf[s]
        if M * Nn * K <= 50_000_000:
            want = orc.linear(xs, wf[s], b[s].numpy())
        else:
            # orc_linear's arithmetic (exact products, double accumulation, one rounding) evaluated by BLAS
            want = (xs.astype(np.float64) @ wf[s].astype(np.float64).T + b[s].numpy().astype(np.float64)).astype(np.float32)
        if relu:
            want = np.maximum(want, 0)
        if ybf:
            assert (np.abs(got[s] - want) <= np.abs(want) * 2.0 ** -8 + 1e-5 * max(1.0, float(np.sqrt((want ** 2).mean())))).all()
        else:
            assert allclose_scaled(got[s], want), np.abs(got[s] - want).max()


def _random_dense_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        S = int(rng.choice([1, 2, 3, 4, 5, 8, 9]))
        M = int(rng.choice([1, 7, 31, 33, 64, 100, 129, 257, 300, 513]))
        Nn = int(rng.choice([17, 24, 80, 81, 100, 159, 160, 161, 200, 256, 330, 640]))
        K = 8 * int(rng.randint(1, 90))
        out.append((S, M, Nn, K, bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))))
    return out


@pytest.mark.parametrize("S,M,Nn,K,shared,relu,ybf", _random_dense_shapes(24, 20261005))
def test_dense_forward_random_shapes_vs_oracle(env, S, M, Nn, K, shared, relu, ybf):
    """A seeded sweep of ragged shapes over the tile choices (256 x 80, 128 / 64 / 32 x 160, 256 x 128), sample counts that do and
    do not fill the XCDs, K tails, unaligned output pitches (N % 8 != 0: the scalar store path): the whole matrix against the oracle."""
    test_dense_forward_vs_oracle(env, S, M, Nn, K, shared, relu, ybf)


def test_dense_forward_rejects_unpadded_weights_and_bad_alignment(env):
    lib, _lib, dev = env["lib"], env["_lib"], env["dev"]
    x = torch.zeros(4, 72, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(1, 16, 72, dtype=torch.bfloat16, device=dev)            # ldw = 72 < roundup(72, 64) = 128
    y = torch.zeros(1, 4, 16, device=dev)
    n0 = lib.bnn_launch_count()
    assert lib.bnn_dense_forward(_lib.ptr(x), 0, 72, _lib.ptr(w), 16 * 72, 72, None, 0, _lib.ptr(y), 64, 16, 4, 16, 72, 1, 0,
                                 _lib.stream_ptr(dev)) == _lib.E_UNSUPPORTED
    assert lib.bnn_dense_forward(_lib.ptr(x), 0, 72, _lib.ptr(w), 16 * 72, 72, None, 0, _lib.ptr(y), 64, 16, 4, 16, 70, 1, 0,
                                 _lib.stream_ptr(dev)) < 0
    assert lib.bnn_launch_count() == n0


@pytest.mark.parametrize("dims,B,S", [((784, 1200, 1200, 10), 512, 8), ((96, 200, 120, 10), 40, 4), ((64, 80, 16), 33, 2)])
def test_layer_on_draw_once_path_equals_fused_kernel_and_oracle(env, dims, B, S):
    """The Module path in bf16 mode (draw plan -> one draw launch for the whole net + dense GEMMs) against (a) the same
    net on the round-1 fused kernels (same keys; same bf16 products, another fp32 summation order) and (b) the oracle."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, fuse_activations
    ops, orc, dev = env["ops"], env["orc"], env["dev"]
    posts = seeded.mlp_posteriors(dims, seed=3)

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(dims[0], dims[-1], S)
            mods = []
            for j, (mw, rw, mb, rb) in enumerate(posts):
                L = NormalLinear(mw.shape[1], mw.shape[0])
                with torch.no_grad():
                    L.weight.mean.copy_(mw); L.weight.scale.copy_(rw); L.bias.mean.copy_(mb); L.bias.scale.copy_(rb)
                mods.append(L)
                if j < len(posts) - 1:
                    mods.append(torch.nn.ReLU())
            self.layers = torch.nn.Sequential(*mods)

        def _forward(self, x):
            return self.layers(x)

    net = Net().to(dev)
    net.mc_batched = True
    fuse_activations(net, bf16_activations=True)
    x = torch.randn(B, dims[0], generator=torch.Generator().manual_seed(9)).to(dev)
    bnn.set_compute("bf16")
    try:
        bnn.manual_seed(31)
        n0 = env["lib"].bnn_launch_count()
        with torch.no_grad():
            y_new = net.forward_stacked(x.bfloat16(), S)
        nl = len(posts)
        assert env["lib"].bnn_launch_count() == n0 + 1 + nl       # ONE draw launch + one contraction per layer
        keys = [(L.weight.draw_key, L.bias.draw_key) for L in net.layers if hasattr(L, "weight")]
        ops.DRAW_ONCE_BF16 = False
        bnn.manual_seed(31)
        with torch.no_grad():
            y_old = net.forward_stacked(x.bfloat16(), S)
        assert [(a.epoch_host, a.stream) for a, _ in keys] == [(L.weight.draw_key.epoch_host, L.weight.draw_key.stream)
                                                                for L in net.layers if hasattr(L, "weight")]
    finally:
        ops.DRAW_ONCE_BF16 = True
        bnn.set_compute("f32")
    a, b = N(y_new), N(y_old)
    rms = float(np.sqrt((b ** 2).mean()))
    # hidden activations are stored in bf16: a sum that differs in the last fp32 bit can round to the neighbouring bf16
    # value on one path -- one bf16 ulp of a few hidden units, ~1e-3 of the output scale after the next layer
    assert np.abs(a - b).max() <= 2.0 ** -8 * max(1.0, rms), (np.abs(a - b).max(), rms)
    # (b) oracle, 48 rows, bf16-rounded operands
    rows = min(B, 48)
    h0 = orc.bf16_round(N(x[:rows]))
    for s in range(S):
        h = h0
        for li, ((mw, rw, mb, rb), (kw, kb)) in enumerate(zip(posts, keys)):
            ew = orc.eps_fill(kw.seed, kw.stream, kw.sample0 + s, kw.epoch_host, 0, tuple(mw.shape), kw.gen)
            eb = orc.eps_fill(kb.seed, kb.stream, kb.sample0 + s, kb.epoch_host, 0, tuple(mb.shape), kb.gen)
            h = orc.linear(h, orc.bf16_round(orc.sample_affine(mw.numpy(), rw.numpy(), ew)), orc.sample_affine(mb.numpy(), rb.numpy(), eb))
            if li < len(posts) - 1:
                h = orc.bf16_round(np.maximum(h, 0))
        r = float(np.sqrt((h ** 2).mean()))
        assert np.abs(a[s, :rows] - h).max() <= 2.0 ** -7 * max(1.0, r), (np.abs(a[s, :rows] - h).max(), r)


def test_draw_plan_is_consumed_once_and_matches_per_layer_draws(env):
    """Bitwise: the network-level draw plan (one launch for all layers) == every layer drawing for itself."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule
    from bayesianneuralnetworks_amd import _mc
    dev = env["dev"]
    torch.manual_seed(4)

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(64, 10, 4)
            self.layers = torch.nn.Sequential(NormalLinear(64, 80), torch.nn.ReLU(), NormalLinear(80, 10))

        def _forward(self, x):
            return self.layers(x)

    net = Net().to(dev)
    net.mc_batched = True
    x = torch.randn(24, 64, device=dev)
    bnn.set_compute("bf16")
    try:
        bnn.manual_seed(8)
        with torch.no_grad():
            y_plan = net.forward_stacked(x, 4)
        assert all(getattr(m, "_predrawn", None) is None for m in net.modules())      # nothing left behind
        bnn.manual_seed(8)
        with torch.no_grad(), _mc.McContext(4, 24, 0):                                # no plan: layers draw for themselves
            y_layer = net._forward(x).view(4, 24, 10)
        assert torch.equal(y_plan, y_layer)
        # training through the plan: gradients exist and are finite
        bnn.manual_seed(8)
        ys = net.forward_stacked(x, 4)
        ys.sum().backward()
        g = net.layers[0].weight.mean.grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
    finally:
        bnn.set_compute("f32")


# ------------------------------------------------------------------ conv2d as implicit GEMM on drawn weights
CONV_CASES = [  # B, C, O, H, W, k, stride, pad, dil, shared_x
    (5, 64, 64, 6, 6, 3, 2, 1, 1, False),          # the MNIST / FashionMNIST layer (configs[2] shape), ragged last tile
    (1024, 64, 64, 6, 6, 3, 2, 1, 1, False),       # configs[2] at full size (checked on a slice)
    (256, 128, 128, 4, 4, 3, 1, 1, 1, False),      # configs[3]
    (9, 128, 128, 4, 4, 3, 1, 1, 1, True),         # shared input (deterministic prefix), ragged tile
    (3, 64, 128, 7, 5, 3, 1, 0, 1, False),         # no padding, non-square, H * W % 4 != 0
    (4, 128, 64, 6, 6, 2, 2, 1, 2, False),         # dilation, even kernel
    (2, 256, 64, 3, 3, 1, 1, 0, 1, False),         # 1 x 1 kernel, C = 256
]


def _random_conv_cases(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        C, O = int(rng.choice([64, 128, 256])), int(rng.choice([64, 128]))
        H, W = int(rng.randint(2, 9)), int(rng.randint(2, 9))
        k, st, pad, dil = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2])), int(rng.choice([0, 1, 2])), int(rng.choice([1, 2]))
        OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
        if OH < 1 or OW < 1 or OH * OW > 128 or pad >= k * dil:
            continue
        out.append((int(rng.choice([1, 3, 9, 17])), C, O, H, W, k, st, pad, dil, bool(rng.randint(2))))
    return out


@pytest.mark.parametrize("B,C,O,H,W,k,st,pad,dil,shared", _random_conv_cases(16, 4242))
def test_conv_dense_path_random_shapes(env, B, C, O, H, W, k, st, pad, dil, shared):
    """A seeded sweep of window / stride / padding / dilation / image sizes over the implicit-GEMM kernel (shapes it does not take
    are skipped: they run the round-1 kernels, which have their own tests)."""
    ops = env["ops"]
    sh, OH, OW = ops._conv_shape((B, C, H, W), (O, C, k, k), (st, st), (pad, pad), (dil, dil), 1)
    if not ops.conv_dense_eligible(sh, OH, OW):
        pytest.skip("not a shape of the implicit-GEMM kernel")
    test_conv_dense_path_vs_oracle(env, B, C, O, H, W, k, st, pad, dil, shared)


@pytest.mark.parametrize("B,C,O,H,W,k,st,pad,dil,shared", CONV_CASES)
def test_conv_dense_path_vs_oracle(env, B, C, O, H, W, k, st, pad, dil, shared):
    """NormalConv2d in bf16 mode (draw once, tap-major + k_conv_bf16) against the oracle's F.conv2d on the same Philox
    draws with bf16-rounded inputs / weights (double accumulate) -- 1e-5 of the output scale; and the launch count: one
    draw + one contraction, no im2col panel."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from bayesianneuralnetworks_amd import _mc
    orc, dev = env["orc"], env["dev"]
    S = 8 if B >= 256 else 3
    torch.manual_seed(B + C + O)
    layer = NormalConv2d(C, O, k, stride=st, padding=pad, dilation=dil).to(dev)
    x = torch.randn(B if shared else S * B, C, H, W, generator=torch.Generator().manual_seed(1))
    xd = x.to(dev)
    bnn.set_compute("bf16")
    try:
        bnn.manual_seed(77)
        n0 = env["lib"].bnn_launch_count()
        with torch.no_grad(), _mc.McContext(S, B, 0):
            y = layer(xd)
        assert env["lib"].bnn_launch_count() == n0 + 2
    finally:
        bnn.set_compute("f32")
    OH, OW = y.shape[-2:]
    y = N(y).reshape(S, B, O, OH, OW)
    kw, kb = layer.weight.draw_key, layer.bias.draw_key
    # the weights the launch drew (same keys -> same launch), back from tap-major to (O, C, KH, KW): bit-identical to K1 ...
    K = C * k * k
    pre = env["ops"].draw_layers([(layer.weight.mean.detach().reshape(O, K), layer.weight.scale.detach().reshape(O, K),
                                   layer.bias.mean.detach(), layer.bias.scale.detach(), kw, kb, k * k)], S)[0]
    wt = pre.w[:, :, :K].reshape(S, O, k * k, C).permute(0, 1, 3, 2).reshape(S, O, C, k, k)
    w1 = env["ops"]._sample_affine_philox_raw(layer.weight.mean.detach(), layer.weight.scale.detach(), kw, out_dtype=torch.bfloat16)
    assert torch.equal(wt, w1) and (pre.w[:, :, K:] == 0).all()
    # ... and within one bf16 ulp of the oracle's draw (a weight within the eps twin's 1e-6 of a rounding boundary rounds the
    # other way on one side, which is why the contraction below is checked on the DEVICE's weights)
    ew = orc.eps_fill(kw.seed, kw.stream, kw.sample0, kw.epoch_host, 0, tuple(layer.weight.shape), kw.gen)
    w_or = orc.sample_affine(N(layer.weight.mean), N(layer.weight.scale), ew)
    assert (np.abs(N(wt[0]) - w_or) <= np.abs(w_or) * 2.0 ** -8 + 1e-6).all()
    nb = min(B, 6)                                              # images the SCALAR oracle checks per sample (first and last ones)
    sel = sorted(set(list(range(nb // 2)) + list(range(B - (nb - nb // 2), B))))
    xr = orc.bf16_round(x.numpy())
    for s in range(S):
        xs = xr[sel] if shared else xr[s * B:(s + 1) * B][sel]
        want = orc.conv2d(xs, N(wt[s]), N(pre.b[s]), stride=(st, st), padding=(pad, pad), dilation=(dil, dil))
        assert allclose_scaled(y[s][sel], want), np.abs(y[s][sel] - want).max()
    # EVERY image of EVERY workgroup tile (the slice above leaves most tiles of configs[2] / [3] unseen): the whole
    # batch against torch's conv2d in float64 on the CPU, fed the device's own bf16 weights and the bf16-rounded inputs --
    # pinned to the scalar oracle on the slice just checked
    xr64 = torch.from_numpy(xr).double()
    worst = 0.0
    for s in range(S):
        xs = xr64 if shared else xr64[s * B:(s + 1) * B]
        full = torch.nn.functional.conv2d(xs, wt[s].double().cpu(), pre.b[s].double().cpu(), stride=st, padding=pad, dilation=dil).numpy()
        assert np.abs(full[sel] - orc.conv2d(xs[sel].float().numpy(), N(wt[s]), N(pre.b[s]), stride=(st, st), padding=(pad, pad),
                                             dilation=(dil, dil))).max() < 1e-4
        assert_close_scaled(y[s], full, 1e-5, "conv %s sample %d, all %d images" % ((B, C, O, H, W, k, st, pad, dil), s, B))
        worst = max(worst, float(np.abs(y[s] - full).max()))
    assert np.isfinite(worst)


def test_conv_dense_path_equals_panel_path_and_trains(env):
    """Same keys: the implicit-GEMM path and the round-1 panel path contract the same bf16 products (fp32 sums in another
    order); the backward (unchanged: panel kernels, re-created draws) runs behind the new forward."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from bayesianneuralnetworks_amd import _mc
    ops, dev = env["ops"], env["dev"]
    torch.manual_seed(3)
    layer = NormalConv2d(64, 64, 3, stride=2, padding=1).to(dev)
    x = torch.randn(4 * 10, 64, 6, 6, device=dev)
    bnn.set_compute("bf16")
    try:
        bnn.manual_seed(5)
        with torch.no_grad(), _mc.McContext(4, 10, 0):
            y_new = layer(x)
        ops.DRAW_ONCE_BF16 = False
        bnn.manual_seed(5)
        with torch.no_grad(), _mc.McContext(4, 10, 0):
            y_old = layer(x)
        ops.DRAW_ONCE_BF16 = True
        assert allclose_scaled(N(y_new), N(y_old))
        bnn.manual_seed(5)
        xg = x.clone().requires_grad_(True)
        with _mc.McContext(4, 10, 0):
            yg = layer(xg)
        yg.square().sum().backward()
        assert torch.isfinite(layer.weight.mean.grad).all() and layer.weight.scale.grad.abs().sum() > 0 and xg.grad is not None
    finally:
        ops.DRAW_ONCE_BF16 = True
        bnn.set_compute("f32")


@pytest.mark.parametrize("B,C,O,HW,k,st,pad", [(37, 64, 64, 6, 3, 2, 1), (5, 128, 32, 5, 3, 1, 1), (1024, 64, 64, 6, 3, 2, 1)])
def test_flipout_conv_fused_kernel_vs_oracle(env, B, C, O, HW, k, st, pad):
    """FlipOutNormalConv2d in bf16 mode without grad = ONE contraction launch (both operands share the A tile; S in the
    fragment's sign bits, R in the epilogue) against the reference's expression (conv.py:207-221) evaluated by the oracle
    on bf16-rounded x / mean / stddev -- and against this package's two-convolution fp32 path on the same signs."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import FlipOutNormalConv2d
    orc, dev = env["orc"], env["dev"]
    torch.manual_seed(B + C)
    layer = FlipOutNormalConv2d(C, O, k, stride=st, padding=pad).to(dev)
    x = torch.randn(B, C, HW, HW, generator=torch.Generator().manual_seed(2)).to(dev)
    bnn.set_compute("bf16")
    try:
        n0 = env["lib"].bnn_launch_count()
        with torch.no_grad():
            y = layer(x)
        assert env["lib"].bnn_launch_count() == n0 + 2            # weights prep + ONE contraction
        R, S = layer.sampled
        bnn.set_compute("f32")
        with torch.no_grad():
            y32 = layer(x, sample=False)                            # same signs, fp32 two-convolution path
    finally:
        bnn.set_compute("f32")
    sel = sorted(set([0, 1, B // 2, B - 1]))
    # the weight operand the launch contracts on: bf16 mean / stddev, tap-major -- within one bf16 ulp of the rounded reference
    # values (a stddev within ~1e-6 of a rounding boundary rounds the other way), and the contraction is checked on THEM
    K = C * k * k
    w2 = env["ops"].flipout_conv_weights(layer.weight.mean, layer.weight.scale)
    assert (w2[:, K:] == 0).all()
    w2 = N(w2[:, :K]).reshape(2 * O, k * k, C).transpose(0, 2, 1).reshape(2 * O, C, k, k)
    mean, std = w2[:O], w2[O:]
    assert (np.abs(mean - N(layer.weight.mean)) <= np.abs(N(layer.weight.mean)) * 2.0 ** -8 + 1e-30).all()
    assert (np.abs(std - N(layer.weight.stddev)) <= N(layer.weight.stddev) * 2.0 ** -8).all()
    xs = orc.bf16_round(N(x)[sel])
    Sn, Rn = N(S).reshape(B, C, 1, 1)[sel], N(R).reshape(B, O, 1, 1)[sel]
    want = (orc.conv2d(xs, mean, None, stride=(st, st), padding=(pad, pad)) +
            orc.conv2d(xs * Sn, std, None, stride=(st, st), padding=(pad, pad)) * Rn)
    got = N(y)[sel]
    assert allclose_scaled(got, want), np.abs(got - want).max()
    rms = float(np.sqrt((N(y32) ** 2).mean()))
    assert np.abs(N(y) - N(y32)).max() <= 2e-2 * max(1.0, rms)     # bf16 operands against fp32 operands


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_draw_plan_leaves_layer_state_alone_for_sample_false(env, mode):
    """`_forward` calling a layer with sample=False must reuse the user-assigned `.sampled` (dense.py:56-58:
    `if sample: self.sample()`) on the batched path too -- the plan draws ahead of the call but must not re-key the layer --
    and a layer `_forward` never reaches keeps its recorded draw.  Batched path == serial loop."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule
    dev = env["dev"]
    torch.manual_seed(14)

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(64, 16, 3)
            self.a = NormalLinear(64, 80)
            self.frozen = NormalLinear(80, 24)
            self.unused = NormalLinear(24, 24)

        def _forward(self, x):
            return self.frozen(torch.relu(self.a(x)), sample=False)

    net = Net().to(dev)
    w = torch.randn(24, 80, device=dev) * 0.1
    b = torch.randn(24, device=dev)
    x = torch.randn(70, 64, device=dev)
    bnn.set_compute(mode)
    try:
        bnn.manual_seed(5)
        net.unused.sample()
        key_unused = net.unused.weight.draw_key
        net.frozen.sampled = (w, b)
        with torch.no_grad():
            net.mc_batched = True
            ys = net.forward_stacked(x, 3)
            assert net.frozen.weight._explicit is w and net.frozen.bias._explicit is b      # still the assigned weights
            assert net.unused.weight.draw_key is key_unused                                 # never reached: not re-keyed
            assert net.a.weight.draw_key.nsamples == 3                                      # consumed its planned draw
            ha = torch.relu(_layer_outputs(env, net.a, x, 3, mode))                         # (3, 70, 80) on the recorded keys
        want = ha.double() @ w.double().T + b.double()
        tol = 1e-5 if mode == "f32" else 2.0 ** -7
        err = (ys.double() - want).abs().max().item()
        assert err <= tol * max(1.0, want.pow(2).mean().sqrt().item()), err
        # no two samples alike (layer a was drawn per sample), and the serial loop agrees on what `frozen` does
        assert not torch.equal(ys[0], ys[1])
        net.mc_batched = False
        with torch.no_grad():
            y_serial = net.forward(x, 1)
        assert net.frozen.weight._explicit is w
        ha1 = torch.relu(_layer_outputs(env, net.a, x, 1, mode))[0]
        want1 = ha1.double() @ w.double().T + b.double()
        assert (y_serial.double() - want1).abs().max().item() <= tol * max(1.0, want1.pow(2).mean().sqrt().item())
    finally:
        bnn.set_compute("f32")


def _layer_outputs(env, layer, x, S, mode):
    """layer(x) for every sample of the layer's RECORDED draw key, from the standalone sampler (K1) in float64."""
    ops = env["ops"]
    kw, kb = layer.weight.draw_key, layer.bias.draw_key
    w = ops._sample_affine_philox_raw(layer.weight.mean.detach(), layer.weight.scale.detach(), kw)
    b = ops._sample_affine_philox_raw(layer.bias.mean.detach(), layer.bias.scale.detach(), kb)
    xx = x
    if mode == "bf16":
        w, xx = w.bfloat16().float(), x.bfloat16().float()
    w = w.reshape(-1, *layer.weight.mean.shape)[-S:]
    b = b.reshape(-1, layer.bias.mean.shape[0])[-S:]
    return (torch.einsum("mk,snk->smn", xx.double(), w.double()) + b.double().unsqueeze(1)).float()


def test_single_row_samples_chain_through_padded_rows(env):
    """M = 1 row per sample, S > 1, N = 1200 (row pitch 1216): a dense call's padded bf16 output fed straight into
    the next dense call -- the (S, 1, K) view has strides (ldy, ldy, 1) and sample s must be read ldy elements on, not K."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, dev = env["ops"], env["dev"]
    S = 4
    m1 = _post((1200, 784), 31, dev)
    m2 = _post((1200, 1200), 32, dev)
    k1 = (DrawKey(7, 21, 0, S, 3), DrawKey(7, 22, 0, S, 3))
    k2 = (DrawKey(7, 23, 0, S, 3), DrawKey(7, 24, 0, S, 3))
    x = torch.randn(1, 784, device=dev).bfloat16()
    with torch.no_grad():
        h = ops.linear_sampled(x, *m1, *k1, True, "bf16", relu=True, out_dtype=torch.bfloat16)         # (S, 1, 1200)
        assert h.shape == (S, 1, 1200)
        y = ops.linear_sampled(h, *m2, *k2, False, "bf16", relu=False, out_dtype=torch.float32)
        hc = h.contiguous()
        y_ref = ops.linear_sampled(hc, *m2, *k2, False, "bf16", relu=False, out_dtype=torch.float32)
    assert torch.equal(y, y_ref)
    w2 = ops._sample_affine_philox_raw(m2[0], m2[1], k2[0]).bfloat16().double()
    b2 = ops._sample_affine_philox_raw(m2[2], m2[3], k2[1]).double()
    want = torch.einsum("smk,snk->smn", hc.double(), w2) + b2.unsqueeze(1)
    assert (y.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.pow(2).mean().sqrt().item())


HEAD_SHAPES = [  # S, M, N (hidden), K, Nh, shared_x, relu
    (8, 512, 1200, 1200, 10, False, True),         # the BASELINE pair: 128 x 160 tiles, 16 partials
    (8, 512, 1200, 784, 10, True, True),
    (4, 512, 1200, 1200, 10, False, True),         # 64 x 160 tiles
    (1, 300, 1200, 264, 16, True, False),          # 32 x 160 tiles, ragged rows, 16 outputs, no ReLU
    (2, 70, 200, 72, 3, False, True),              # ragged everything: N = 200 is 1.25 panels of 160, K tail
    (3, 130, 72, 200, 10, True, True),             # N <= 80: 256 x 80 tile, one partial per panel
    (2, 260, 264, 328, 7, False, False),           # 256 x 128 tile (N % 80 != 0, N >= 128)
]


def _random_head_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([1, 2, 3, 5, 8])), int(rng.choice([1, 17, 64, 100, 129, 300])),
                    8 * int(rng.choice([3, 9, 10, 11, 20, 21, 33, 41])), 8 * int(rng.randint(1, 60)), int(rng.randint(1, 17)),
                    bool(rng.randint(2)), bool(rng.randint(2))))
    return out


@pytest.mark.parametrize("S,M,Nn,K,Nh,shared,relu", _random_head_shapes(14, 314))
def test_fused_head_random_shapes(env, S, M, Nn, K, Nh, shared, relu):
    """A seeded sweep over hidden widths (every tile choice), head widths 1..16, ragged rows and K tails."""
    test_fused_head_equals_two_dense_launches_and_double(env, S, M, Nn, K, Nh, shared, relu)


@pytest.mark.parametrize("S,M,Nn,K,Nh,shared,relu", HEAD_SHAPES)
def test_fused_head_equals_two_dense_launches_and_double(env, S, M, Nn, K, Nh, shared, relu):
    """bnn_dense_forward_head (hidden layer + classifier head in one launch, the hidden activation never stored) against (a) the
    two plain launches on the same drawn weights -- same bf16 products, fp32 sums in another order -- and (b) float64 on the
    device's own bf16 operands with the hidden activation rounded to bf16; the MC reduction over the partial logits equals the
    mean of the summed logits."""
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, dev, lib = env["ops"], env["dev"], env["lib"]
    hid = _post((Nn, K), 41, dev)
    head = _post((Nh, Nn), 42, dev)
    pre = ops.draw_layers([(*hid, DrawKey(3, 31, 0, S, 5, gen=1), DrawKey(3, 32, 0, S, 5, gen=1)),
                           (*head, DrawKey(3, 33, 0, S, 5, gen=1), DrawKey(3, 34, 0, S, 5, gen=1))], S)
    g = torch.Generator().manual_seed(S * 100 + M)
    x = torch.randn((M, K) if shared else (S, M, K), generator=g).to(dev).bfloat16()
    xs = 0 if shared else M * K
    n0 = lib.bnn_launch_count()
    hp = ops._dense_head_raw(x, xs, M, pre[0], K, relu, pre[1])
    assert lib.bnn_launch_count() == n0 + 1
    parts = lib.bnn_dense_head_parts(M, Nn, S)
    assert hp.p.shape == (parts, S, M, Nh) and parts >= 1
    got = hp.logits()
    # (a) the two plain launches
    h = ops._dense_raw(x, xs, M, pre[0], K, relu, torch.bfloat16, pad_rows=True)
    ld = h.stride(-2) if M > 1 else Nn
    y2 = ops._dense_raw(h, h.stride(0), M, pre[1], Nn, False, torch.float32, ldx=ld)
    assert_close_scaled(N(got), N(y2), 1e-5, "fused head vs two launches")
    # (b) float64 on the same bf16 operands
    xd = x.double().cpu() if not shared else x.double().cpu().unsqueeze(0).expand(S, M, K)
    w1 = pre[0].w[:, :, :K].double().cpu(); b1 = pre[0].b.double().cpu()
    w2 = pre[1].w[:, :, :Nn].double().cpu(); b2 = pre[1].b.double().cpu()
    hh = torch.einsum("smk,snk->smn", xd, w1) + b1.unsqueeze(1)
    if relu:
        hh = hh.clamp_min(0)
    hh = hh.float().bfloat16().double()
    want = torch.einsum("smk,snk->smn", hh, w2) + b2.unsqueeze(1)
    # a hidden value within fp32 rounding of a bf16 boundary may round the other way: one bf16 ulp of one addend
    tol = 1e-5 + 2.0 ** -8 * float((w2.abs().max() * hh.abs().max()) / max(1.0, float(want.pow(2).mean().sqrt())))
    assert_close_scaled(N(got), want.numpy(), tol, "fused head vs float64")
    # the step's tail: ONE launch reduces the partials over (part, sample)
    n1 = lib.bnn_launch_count()
    pm = ops.mc_mean(hp)
    assert lib.bnn_launch_count() == n1 + 1 and pm.shape == (M, Nh)
    assert_close_scaled(N(pm), N(got).astype(np.float64).mean(0), 1e-5, "mc_mean over partial logits")
    assert torch.equal(pm, ops.mc_mean(hp))                                     # fixed order: bitwise reproducible


def test_predictive_mean_fuses_the_head_and_matches_forward_stacked(env):
    """BayesianNetworkModule.predictive_mean on a fuse_activations'ed MLP in the bf16 mode: 4 launches (draw, layer 1, layer 2 +
    head, reduction) instead of 5, the same predictive mean as forward_stacked(...).mean(0) on the same seed, the head's keys
    recorded as its own sample() would; outside predictive_mean (forward / forward_stacked) nothing changes."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, fuse_activations
    dev, lib = env["dev"], env["lib"]
    torch.manual_seed(9)

    class Net(BayesianNetworkModule):
        def __init__(self):
            super().__init__(96, 10, 4)
            self.layers = torch.nn.Sequential(NormalLinear(96, 160), torch.nn.ReLU(), NormalLinear(160, 240), torch.nn.ReLU(), NormalLinear(240, 10))

        def _forward(self, x):
            return self.layers(x)

    net = Net().to(dev)
    net.mc_batched = True
    fuse_activations(net, bf16_activations=True)
    assert net.layers[2].__dict__.get("_fuse_head") is None                         # opt-in: nothing fused without fuse_head=True
    fuse_activations(net, bf16_activations=True, fuse_head=True)
    assert net.layers[2].__dict__.get("_fuse_head") is net.layers[4] and net.layers[0].__dict__.get("_fuse_head") is None
    x = torch.randn(70, 96, device=dev)
    bnn.set_compute("bf16")
    try:
        with torch.no_grad():
            bnn.manual_seed(21)
            n0 = lib.bnn_launch_count()
            ys = net.forward_stacked(x, 4)
            assert lib.bnn_launch_count() == n0 + 4 and ys.shape == (4, 70, 10)          # draw + three layers
            keys_plain = net.layers[4].weight.draw_key
            bnn.manual_seed(21)
            n0 = lib.bnn_launch_count()
            pm = net.predictive_mean(x, 4)
            assert lib.bnn_launch_count() == n0 + 4                                       # draw, layer 1, layer 2 + head, reduction
            kf = net.layers[4].weight.draw_key
            assert (kf.seed, kf.stream, kf.sample0, kf.nsamples, kf.epoch_host, kf.gen) == \
                   (keys_plain.seed, keys_plain.stream, keys_plain.sample0, keys_plain.nsamples, keys_plain.epoch_host, keys_plain.gen)
        assert_close_scaled(N(pm), N(ys).astype(np.float64).mean(0), 1e-5, "predictive_mean vs forward_stacked().mean(0)")
        # with gradients wanted the pair is not fused (the backward needs the hidden activation)
        bnn.manual_seed(21)
        pm_g = net.predictive_mean(x, 4)
        assert pm_g.requires_grad or net.layers[4].weight.mean.grad is None
        assert_close_scaled(N(pm_g), N(pm), 1e-5, "predictive_mean with grad enabled")
    finally:
        bnn.set_compute("f32")


@pytest.mark.parametrize("B,C,O,HW,k,st,pad", [(37, 64, 64, 6, 3, 2, 1), (9, 128, 128, 4, 3, 1, 1), (1024, 64, 64, 6, 3, 2, 1), (5, 128, 32, 5, 3, 1, 1),
                                               (3, 64, 32, 7, 2, 1, 0), (11, 256, 64, 3, 1, 1, 0)])
def test_flipout_conv_f32_mode_without_the_panel_vs_double(env, B, C, O, HW, k, st, pad):
    """FlipOutNormalConv2d in the fp32 PARITY mode at inference: mean and stddev as three bf16 planes (one launch), two
    implicit-GEMM contractions on three-plane operands, no im2col panel -- conv.py:207-221 evaluated by torch in float64 on the
    layer's own mean / stddev / signs, EVERY image, 1e-5 of the output scale; with gradients wanted the panel path, same values."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import FlipOutNormalConv2d
    dev, lib = env["dev"], env["lib"]
    torch.manual_seed(B + C + 3)
    layer = FlipOutNormalConv2d(C, O, k, stride=st, padding=pad).to(dev)
    x = torch.randn(B, C, HW, HW, generator=torch.Generator().manual_seed(4)).to(dev)
    bnn.set_compute("f32")
    n0 = lib.bnn_launch_count()
    with torch.no_grad():
        y = layer(x)
    n_hip = lib.bnn_launch_count() - n0
    R, S = layer.sampled
    F = torch.nn.functional
    x64, m64, s64 = x.double().cpu(), layer.weight.mean.detach().double().cpu(), layer.weight.stddev.detach().double().cpu()
    want = (F.conv2d(x64, m64, None, st, pad) + F.conv2d(x64 * S.double().cpu().expand_as(x64), s64, None, st, pad) * R.double().cpu())
    assert_close_scaled(N(y), want.numpy(), 1e-5, "fp32-mode Flipout conv vs float64")
    xg = x.clone().requires_grad_(True)
    yg = layer(xg, sample=False)                          # gradients wanted: the two-convolution panel path on the same signs
    assert yg.requires_grad
    assert_close_scaled(N(yg), N(y), 2e-5, "panel path vs implicit GEMM")
    # (K1's stddev launch may or may not be cached by the layer.)  2 O <= 128: the planes launch + ONE contraction that shares the A
    # fragment between the two convolutions (bnn_conv2d_flipout_forward_x3); O = 128: the planes launch + two contractions
    assert n_hip <= (3 if O <= 64 else 5), n_hip


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,C,O,L,k,st,pad,dil", [(16, 64, 64, 40, 3, 2, 1, 1), (3, 5, 7, 11, 3, 1, 0, 1), (4, 128, 128, 9, 5, 1, 2, 1), (2, 8, 12, 17, 3, 2, 2, 2)])
def test_normal_conv1d_on_the_device_kernels(env, B, C, O, L, k, st, pad, dil, mode):
    """NormalConv1d (conv.py:76-96) on a CUDA input = the conv2d kernels on images of height 1 (implicit GEMM where the shape
    allows it), same draws: torch's conv1d in float64 on the K1 draw of the recorded keys, every sample, every row; gradients flow
    to the (O, C, k) Parameters and agree with float64 autograd on the same draws."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalConv1d
    from bayesianneuralnetworks_amd import _mc
    ops, dev = env["ops"], env["dev"]
    S = 3
    torch.manual_seed(B + C + L)
    layer = NormalConv1d(C, O, k, stride=st, padding=pad, dilation=dil).to(dev)
    x = torch.randn(B, C, L, generator=torch.Generator().manual_seed(6))
    xd = x.to(dev)
    bnn.set_compute(mode)
    try:
        bnn.manual_seed(5)
        with torch.no_grad(), _mc.McContext(S, B, 0):
            y = layer(xd)
        OL = y.shape[-1]
        assert y.shape == (S * B, O, OL)
        kw, kb = layer.weight.draw_key, layer.bias.draw_key
        w = ops._sample_affine_philox_raw(layer.weight.mean.detach(), layer.weight.scale.detach(), kw).double().cpu()      # (S, O, C, k)
        b = ops._sample_affine_philox_raw(layer.bias.mean.detach(), layer.bias.scale.detach(), kb).double().cpu()
        tol = 1e-5 if mode == "f32" else 2e-2
        for s in range(S):
            want = torch.nn.functional.conv1d(x.double(), w[s], b[s], st, pad, dil)
            assert_close_scaled(N(y[s * B:(s + 1) * B]), want.numpy(), tol, "NormalConv1d %s sample %d" % (mode, s))
        # backward: same draws (sample=False), gradient w.r.t. the (O, C, k) mean against float64 autograd
        with _mc.McContext(S, B, 0):
            yg = layer(xd, sample=False)
        gy = torch.randn(yg.shape, generator=torch.Generator().manual_seed(8)).to(dev)
        (g_mu,) = torch.autograd.grad(yg, (layer.weight.mean,), gy)
        assert g_mu.shape == (O, C, k)
        x64 = x.double()
        want_g = torch.zeros(O, C, k, dtype=torch.float64)
        for s in range(S):
            ws = w[s].clone().requires_grad_(True)
            ys_ = torch.nn.functional.conv1d(x64, ws, b[s], st, pad, dil)
            (gw,) = torch.autograd.grad(ys_, (ws,), gy[s * B:(s + 1) * B].double().cpu())
            want_g += gw
        assert_close_scaled(N(g_mu), want_g.numpy(), 1e-4 if mode == "f32" else 3e-2, "NormalConv1d g_mu")
    finally:
        bnn.set_compute("f32")


@pytest.mark.parametrize("B,C,O,HW,k,st,pad,bias", [(9, 64, 64, 6, 3, 2, 1, True), (5, 128, 128, 4, 3, 1, 1, False), (3, 64, 128, 5, 1, 1, 0, True)])
def test_plain_conv_f32_mode_without_the_panel_vs_double(env, B, C, O, HW, k, st, pad, bias):
    """ops.conv2d_plain with ONE explicit weight at inference in the fp32 parity mode: the weight as three bf16 planes (kind 1,
    one launch) + the three-plane implicit GEMM -- no im2col panel; float64 conv2d on every image, 1e-5 of the output scale.
    With a gradient wanted the call stays on the panel kernels (autograd) and agrees."""
    ops, dev, lib = env["ops"], env["dev"], env["lib"]
    g = torch.Generator().manual_seed(B + C + O)
    x = torch.randn(B, C, HW, HW, generator=g)
    w = torch.randn(1, O, C, k, k, generator=g) * 0.05
    b = torch.randn(1, O, generator=g) if bias else None
    xd, wd, bd = x.to(dev), w.to(dev), (b.to(dev) if bias else None)
    geo = ((st, st), (pad, pad), (1, 1), 1)
    assert ops.conv2d_plain_x3_eligible(xd, wd, *geo, "f32")
    n0 = lib.bnn_launch_count()
    with torch.no_grad():
        y = ops.conv2d_plain(xd, wd, bd, True, *geo, "f32")
    assert lib.bnn_launch_count() == n0 + 2                           # planes + contraction
    want = torch.nn.functional.conv2d(x.double(), w[0].double(), b[0].double() if bias else None, st, pad)
    assert y.shape == (1,) + tuple(want.shape)
    assert_close_scaled(N(y[0]), want.numpy(), 1e-5, "plain conv, fp32 mode, implicit GEMM")
    wg = wd.clone().requires_grad_(True)
    yg = ops.conv2d_plain(xd, wg, bd, True, *geo, "f32")
    assert yg.requires_grad
    assert_close_scaled(N(yg[0]), N(y[0]), 2e-5, "panel path vs implicit GEMM")


@pytest.mark.parametrize("B,C,O,H,W,k,st,pad,dil,shared", _random_conv_cases(12, 777))
def test_conv_f32_mode_random_shapes(env, B, C, O, H, W, k, st, pad, dil, shared):
    """The same sweep over the three-plane implicit GEMM of the fp32 parity mode."""
    ops = env["ops"]
    sh, OH, OW = ops._conv_shape((B, C, H, W), (O, C, k, k), (st, st), (pad, pad), (dil, dil), 1)
    if not ops.conv_dense_x3_eligible(sh, OH, OW):
        pytest.skip("not a shape of the three-plane implicit-GEMM kernel")
    test_conv_f32_mode_without_the_panel_vs_double(env, B, C, O, H, W, k, st, pad, dil, shared)


@pytest.mark.parametrize("B,C,O,H,W,k,st,pad,dil,shared", CONV_CASES)
def test_conv_f32_mode_without_the_panel_vs_double(env, B, C, O, H, W, k, st, pad, dil, shared):
    """NormalConv2d in the fp32 PARITY mode at inference (bnn_conv2d_dense_forward_x3: weights drawn as three bf16 planes,
    images split into planes in LDS, six plane pairs per 64-k block, no im2col panel): 1e-5 of the output scale against
    torch's conv2d in float64 on the K1 draw of the recorded keys -- EVERY image -- and pinned to the scalar oracle on a slice;
    two launches (draw + contraction); with gradients wanted the layer stays on the panel kernels and agrees."""
    import bayesianneuralnetworks_amd as bnn
    from bayesianneuralnetworks_amd.nn import NormalConv2d
    from bayesianneuralnetworks_amd import _mc
    ops, orc, dev, lib = env["ops"], env["orc"], env["dev"], env["lib"]
    assert ops.CONV_X3_F32
    S = 8 if B >= 256 else 3
    torch.manual_seed(B + C + O + 1)
    layer = NormalConv2d(C, O, k, stride=st, padding=pad, dilation=dil).to(dev)
    x = torch.randn(B if shared else S * B, C, H, W, generator=torch.Generator().manual_seed(2))
    xd = x.to(dev)
    bnn.set_compute("f32")
    bnn.manual_seed(78)
    n0 = lib.bnn_launch_count()
    with torch.no_grad(), _mc.McContext(S, B, 0):
        y = layer(xd)
    assert lib.bnn_launch_count() == n0 + 2, "draw (three planes) + one contraction, no im2col"
    OH, OW = y.shape[-2:]
    y = N(y).reshape(S, B, O, OH, OW)
    kw, kb = layer.weight.draw_key, layer.bias.draw_key
    assert kw.gen == 0                                                           # the fp32 mode draws from the default stream
    w = ops._sample_affine_philox_raw(layer.weight.mean.detach(), layer.weight.scale.detach(), kw).double().cpu()    # (S, O, C, k, k)
    b = ops._sample_affine_philox_raw(layer.bias.mean.detach(), layer.bias.scale.detach(), kb).double().cpu()
    x64 = x.double()
    for s in range(S):
        xs = x64 if shared else x64[s * B:(s + 1) * B]
        full = torch.nn.functional.conv2d(xs, w[s], b[s], stride=st, padding=pad, dilation=dil).numpy()
        assert_close_scaled(y[s], full, 1e-5, "fp32-mode conv %s sample %d" % ((B, C, O, H, W, k, st, pad, dil), s))
    sel = [0, B - 1]
    want = orc.conv2d((x if shared else x[:B])[sel].numpy(), w[0].float().numpy(), b[0].float().numpy(), stride=(st, st), padding=(pad, pad), dilation=(dil, dil))
    assert_close_scaled(y[0][sel], want, 1e-5, "fp32-mode conv vs scalar oracle")
    # training-time forward (gradients wanted): the panel kernels, same keys -> same values to 1e-5
    with _mc.McContext(S, B, 0):
        n1 = lib.bnn_launch_count()
        yg = layer(xd, sample=False)
    assert yg.requires_grad
    assert_close_scaled(N(yg).reshape(S, B, O, OH, OW), y, 2e-5, "panel path vs implicit GEMM")
