"""GPU parity of the fp32 PARITY mode on the dense kernel (csrc/bnn_dense.hip: bnn_split_bf16x3, bnn_draw_multi with
BNN_BF16X3 output, bnn_dense_forward_x3): every operand as three bf16 planes, six partial products per k-block on the
bf16 MFMA.  Tolerance: 1e-5 of the output scale against the CPU oracle in double on the same Philox draws -- the fp32
mode's bar (north_star), same as the fused kernel's."""
import numpy as np
import pytest
import torch

from conftest import assert_close_scaled
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


def _post(shape, seed, dev):
    gen = torch.Generator().manual_seed(seed)
    return [t.to(dev) for t in seeded.posterior(gen, shape, True)]


def _planes_sum(p):
    return p[0].double() + p[1].double() + p[2].double()


@pytest.mark.parametrize("M,K", [(512, 784), (3, 8), (65, 1200)])
def test_split_planes_are_the_exact_three_term_expansion(env, M, K):
    ops, dev = env["ops"], env["dev"]
    x = torch.randn(M, K, device=dev) * torch.logspace(-6, 6, K, device=dev)
    p = ops.split_x3(x)
    assert p.shape == (3, 1, M, (K + 63) // 64 * 64) and p.dtype == torch.bfloat16
    h = x.bfloat16()
    assert torch.equal(p[0, 0, :, :K], h)                                       # plane 0 IS the bf16 rounding
    r = x - h.float()
    assert torch.equal(p[1, 0, :, :K], r.bfloat16())
    assert torch.equal(p[2, 0, :, :K], (r - r.bfloat16().float()).bfloat16())
    err = (_planes_sum(p[:, 0, :, :K]) - x.double()).abs()
    assert (err <= x.abs().double() * 2.0 ** -24 + 1e-300).all()


@pytest.mark.parametrize("shape,S", [((1200, 784), 8), ((80, 264), 2), ((40, 16), 1)])
def test_three_plane_draw_sums_to_the_fp32_draw(env, shape, S):
    from bayesianneuralnetworks_amd._rng import DrawKey
    ops, dev = env["ops"], env["dev"]
    mw, rw, mb, rb = _post(shape, 5, dev)
    kw, kb = DrawKey(99, 11, 2, S, 7), DrawKey(99, 12, 2, S, 7)
    n0 = env["lib"].bnn_launch_count()
    pre = ops.draw_layers([(mw, rw, mb, rb, kw, kb)], S, x3=True)[0]
    assert env["lib"].bnn_launch_count() == n0 + 1
    Nn, K = shape
    kp = (K + 63) // 64 * 64
    assert pre.w.shape == (3, S, Nn, kp)
    w32 = ops._sample_affine_philox_raw(mw, rw, kw)                             # the fp32 draw of the same key (K1)
    assert torch.equal(pre.w[0, :, :, :K], w32.bfloat16())
    err = (_planes_sum(pre.w[:, :, :, :K]) - w32.double()).abs()
    assert (err <= w32.abs().double() * 2.0 ** -24).all()
    assert (pre.w[:, :, :, K:] == 0).all()
    assert torch.equal(pre.b, ops._sample_affine_philox_raw(mb, rb, kb))


X3_SHAPES = [  # S, M, N, K, shared, relu, planes_out
    (8, 512, 1200, 784, True, True, True),
    (8, 512, 1200, 1200, False, True, False),
    (2, 100, 200, 72, True, False, False),
    (1, 64, 48, 8, True, True, True),
    (3, 130, 170, 200, False, False, True),
    (1, 300, 4096, 256, True, False, False),
    (1, 512, 1200, 784, True, True, True),         # one sample: 32 x 160 tiles
    (4, 512, 1200, 1200, False, True, False),      # four: 64 x 160
    (8, 512, 10, 1200, False, False, False),       # the classifier head: K-split kernel, six passes
    (2, 70, 16, 2048, True, True, False),
    # the 256 x 128 tile (k_dense_bf16<4, 8, 4, 1, 3>: N >= 128 and N % 80 != 0) on three-plane operands -- the instantiation
    # bench.py's roofline_wide_f32 leg times (configs[4]: 4096 x 4096 at batch 4096), checked on rows of every kind of tile
    (1, 4096, 4096, 4096, True, False, False),
    (1, 4096, 4096, 4096, True, True, True),
    (2, 300, 264, 328, False, True, False),        # ragged M, N and a K tail (328 = 5 x 64 + 8) on that tile
    (3, 257, 136, 72, True, False, True),
    (2, 90, 1200, 520, False, True, True),         # 64 x 160 tiles, ragged, K tail, planes out
    (1, 40, 240, 136, True, False, False),         # 32 x 160
]


def _check_rows(M):
    """Rows compared against float64: all of them up to 600, else whole 16-row groups from both ends, the middle, and around
    every 256- and 128-row tile seam nearby (the CPU einsum in double over 4096 x 4096 x 4096 would take minutes)."""
    if M <= 600:
        return torch.arange(M)
    picks = [0, 112, 240, 256, 1008, 1024, 2040, 2048 + 128, M - 272, M - 16]
    return torch.cat([torch.arange(r, r + 16) for r in picks])


@pytest.mark.parametrize("S,M,Nn,K,shared,relu,planes_out", X3_SHAPES)
def test_dense_x3_vs_double(env, S, M, Nn, K, shared, relu, planes_out):
    ops, dev = env["ops"], env["dev"]
    g = torch.Generator().manual_seed(S * 1000 + M)
    x = torch.randn((M, K) if shared else (S, M, K), generator=g).to(dev)
    w = (torch.randn(S, Nn, K, generator=g) * 0.1).to(dev)
    b = torch.randn(S, Nn, generator=g).to(dev)
    kp = (K + 63) // 64 * 64
    wp = torch.zeros(3, S, Nn, kp, dtype=torch.bfloat16, device=dev)
    wp[:, :, :, :K] = ops.split_x3(w.reshape(S * Nn, K))[:, 0, :, :K].reshape(3, S, Nn, K)
    pre = ops.Predrawn(wp, b, None, None)
    xp = ops.split_x3(x.reshape(-1, K))
    if not shared:
        xp = xp.view(3, S, M, xp.shape[3])
    n0 = env["lib"].bnn_launch_count()
    y = ops._dense_raw_x3(xp, shared, M, pre, K, relu, planes_out)
    assert env["lib"].bnn_launch_count() == n0 + 1
    if planes_out:
        assert isinstance(y, ops.X3Activation) and y.planes.shape[:3] == (3, S, M)
        got = y.float().reshape(S, M, Nn)
    else:
        got = y
    rows = _check_rows(M)
    xd = x.double().cpu() if not shared else x.double().cpu().unsqueeze(0).expand(S, M, K)
    want = torch.einsum("smk,snk->smn", xd[:, rows], w.double().cpu()) + b.double().cpu().unsqueeze(1)
    if relu:
        want = want.clamp_min(0)
    assert_close_scaled(N(got[:, rows.to(dev)]), want.numpy(), 1e-5, "x3 dense %s" % ((S, M, Nn, K),))
    if planes_out:
        # the planes are the split of an fp32 value: plane 0 is its bf16 rounding, plane 1 the rounding of what is left, ...
        v = y.float().reshape(S, M, Nn)
        h = v.bfloat16()
        r = v - h.float()
        assert torch.equal(y.planes[0, ..., :Nn], h) and torch.equal(y.planes[1, ..., :Nn], r.bfloat16())
        assert torch.equal(y.planes[2, ..., :Nn], (r - r.bfloat16().float()).bfloat16())
        # ... and the three planes carry the result to fp32 precision: against float64 directly, in double
        p64 = (y.planes[0].double() + y.planes[1].double() + y.planes[2].double())[..., :Nn]
        assert_close_scaled(p64[:, rows.to(dev)].cpu().numpy(), want.numpy(), 1e-5, "x3 planes")


def test_dense_x3_argument_errors(env):
    lib, _lib, dev = env["lib"], env["_lib"], env["dev"]
    a = torch.zeros(3, 1, 64, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(3, 1, 32, 64, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(1, 64, 32, device=dev)
    sp = _lib.stream_ptr(dev)
    ok = lib.bnn_dense_forward_x3(_lib.ptr(a), 64 * 64, 0, 64, _lib.ptr(w), 32 * 64, 32 * 64, 64, None, 0, _lib.ptr(y), 0, 64 * 32, 32, 64, 32, 64, 1, 0, sp)
    assert ok == 0
    # N <= 16: fp32 outputs only
    assert lib.bnn_dense_forward_x3(_lib.ptr(a), 64 * 64, 0, 64, _lib.ptr(w), 32 * 64, 32 * 64, 64, None, 0, _lib.ptr(y), 64 * 16, 64 * 16, 16, 64, 16, 64, 1,
                                    _lib.FLAG_Y_BF16, sp) == _lib.E_UNSUPPORTED
    # a plane stride smaller than one plane
    assert lib.bnn_dense_forward_x3(_lib.ptr(a), 8, 0, 64, _lib.ptr(w), 32 * 64, 32 * 64, 64, None, 0, _lib.ptr(y), 0, 64 * 32, 32, 64, 32, 64, 1, 0, sp) != 0
    x = torch.zeros(4, 12, device=dev)
    o = torch.zeros(3, 4, 64, dtype=torch.bfloat16, device=dev)
    assert lib.bnn_split_bf16x3(_lib.ptr(x), 4, 12, 12, _lib.ptr(o), 64, 4 * 64, sp) == _lib.E_UNSUPPORTED   # cols % 8


@pytest.mark.parametrize("dims,B,S", [((784, 1200, 1200, 10), 512, 8), ((96, 200, 120, 10), 70, 4)])
def test_fp32_mode_network_on_dense_path_equals_fused_kernels_and_oracle(env, dims, B, S):
    """The same MLP forward in the fp32 parity mode on both implementations -- three-plane dense path (draw plan + X3
    activations between the hidden layers) and the fused in-kernel-draw kernels -- same DrawKeys: both within 1e-5 of the
    oracle in double, and of each other."""
    bnn, ops, orc, dev = env["bnn"], env["ops"], env["orc"], env["dev"]
    from bayesianneuralnetworks_amd.nn import NormalLinear, BayesianNetworkModule, fuse_activations
    gen = torch.Generator().manual_seed(3)
    post = [seeded.posterior(gen, (dims[i + 1], dims[i]), True) for i in range(len(dims) - 1)]

    class MLP(BayesianNetworkModule):
        def __init__(self):
            super().__init__(dims[0], dims[-1], S)
            mods = []
            for j, (mw, rw, mb, rb) in enumerate(post):
                L = NormalLinear(mw.shape[1], mw.shape[0])
                with torch.no_grad():
                    L.weight.mean.copy_(mw); L.weight.scale.copy_(rw); L.bias.mean.copy_(mb); L.bias.scale.copy_(rb)
                mods.append(L)
                if j < len(post) - 1:
                    mods.append(torch.nn.ReLU())
            self.layers = torch.nn.Sequential(*mods)

        def _forward(self, x):
            return self.layers(x)

    net = MLP().to(dev)
    net.mc_batched = True
    fuse_activations(net, bf16_activations=True)
    x = torch.randn(B, dims[0], generator=gen).to(dev)
    old_mode = bnn.get_compute() if hasattr(bnn, "get_compute") else "f32"
    bnn.set_compute("f32")
    saved = ops.DENSE_X3_F32
    try:
        outs = {}
        for x3 in (True, False):
            ops.DENSE_X3_F32 = x3
            bnn.manual_seed(11)
            n0 = env["lib"].bnn_launch_count()
            with torch.no_grad():
                outs[x3] = net.forward_stacked(x, S)
            launches = env["lib"].bnn_launch_count() - n0
            if x3:
                # draw (every layer; the split of the input into planes rides in it: kind 3) + one dense launch per layer
                assert launches == 1 + (len(dims) - 1), launches
            keys = [(L.weight.draw_key, L.bias.draw_key) for L in net.layers if hasattr(L, "weight")]
            outs[(x3, "keys")] = keys
    finally:
        ops.DENSE_X3_F32 = saved
        bnn.set_compute(old_mode)
    assert_close_scaled(N(outs[True]), N(outs[False]), 2e-5, "dense path vs fused kernels")
    # oracle on the recorded keys (sample by sample, double accumulation)
    keys = outs[(True, "keys")]
    xs = N(x)[:48]
    for s in range(S):
        h = xs.astype(np.float64)
        for j, (mw, rw, mb, rb) in enumerate(post):
            kw, kb = keys[j]
            ew = orc.eps_fill(kw.seed, kw.stream, kw.sample0 + s, kw.epoch_host, 0, tuple(mw.shape), kw.gen)
            eb = orc.eps_fill(kb.seed, kb.stream, kb.sample0 + s, kb.epoch_host, 0, tuple(mb.shape), kb.gen)
            w = orc.sample_affine(mw.numpy(), rw.numpy(), ew).astype(np.float64)
            b = orc.sample_affine(mb.numpy(), rb.numpy(), eb).astype(np.float64)
            h = h @ w.T + b
            if j < len(post) - 1:
                h = np.maximum(h, 0)
        assert_close_scaled(N(outs[True][s, :48]), h, 1e-5, "dense path vs oracle, sample %d" % s)
        assert_close_scaled(N(outs[False][s, :48]), h, 1e-5, "fused kernels vs oracle, sample %d" % s)


X3_HEAD_SHAPES = [  # S, M, N (hidden), K, Nh, shared_x, relu
    (8, 512, 1200, 1200, 10, False, True),         # the BASELINE pair
    (8, 512, 1200, 784, 10, True, True),
    (4, 512, 1200, 1200, 10, False, True),         # 64 x 160 tiles
    (1, 300, 1200, 264, 16, True, False),          # 32 x 160 tiles, ragged rows
    (2, 70, 120, 72, 3, False, True),              # N < 128: one ragged panel
]


@pytest.mark.parametrize("S,M,Nn,K,Nh,shared,relu", X3_HEAD_SHAPES)
def test_fused_head_x3_vs_double(env, S, M, Nn, K, Nh, shared, relu):
    """bnn_dense_forward_x3_head (fp32 parity mode: hidden layer + classifier head in one launch on three-plane operands) against
    float64 on the fp32 operands the planes encode -- 1e-5 of the output scale -- and against the two plain three-plane launches."""
    ops, dev, lib = env["ops"], env["dev"], env["lib"]
    g = torch.Generator().manual_seed(S * 100 + M + 7)
    x = torch.randn((M, K) if shared else (S, M, K), generator=g).to(dev)
    w1 = (torch.randn(S, Nn, K, generator=g) * 0.1).to(dev); b1 = torch.randn(S, Nn, generator=g).to(dev)
    w2 = (torch.randn(S, Nh, Nn, generator=g) * 0.1).to(dev); b2 = torch.randn(S, Nh, generator=g).to(dev)

    def planes(w):
        s_, n_, k_ = w.shape
        kp = (k_ + 63) // 64 * 64
        wp = torch.zeros(3, s_, n_, kp, dtype=torch.bfloat16, device=dev)
        wp[:, :, :, :k_] = ops.split_x3(w.reshape(s_ * n_, k_))[:, 0, :, :k_].reshape(3, s_, n_, k_)
        return wp
    pre1, pre2 = ops.Predrawn(planes(w1), b1, None, None), ops.Predrawn(planes(w2), b2, None, None)
    xp = ops.split_x3(x.reshape(-1, K))
    if not shared:
        xp = xp.view(3, S, M, xp.shape[3])
    assert ops.dense_head_x3_eligible(M, Nn, pre1, pre2)
    n0 = lib.bnn_launch_count()
    hp = ops._dense_head_raw_x3(xp, shared, M, pre1, K, relu, pre2)
    assert lib.bnn_launch_count() == n0 + 1
    got = hp.logits()
    xd = x.double().cpu() if not shared else x.double().cpu().unsqueeze(0).expand(S, M, K)
    hh = torch.einsum("smk,snk->smn", xd, w1.double().cpu()) + b1.double().cpu().unsqueeze(1)
    if relu:
        hh = hh.clamp_min(0)
    want = torch.einsum("smk,snk->smn", hh, w2.double().cpu()) + b2.double().cpu().unsqueeze(1)
    assert_close_scaled(N(got), want.numpy(), 1e-5, "x3 fused head vs float64")
    y1 = ops._dense_raw_x3(xp, shared, M, pre1, K, relu, True)                  # planes out
    y2 = ops._dense_raw_x3(y1.planes, False, M, pre2, Nn, False, False)
    assert_close_scaled(N(got), N(y2), 1e-5, "x3 fused head vs two launches")


def test_inference_paths_are_not_taken_when_only_scale_or_bias_trains(env):
    """A frozen posterior mean with a trainable scale / bias still needs autograd: the fp32 mode's three-plane dense path and the
    one-launch Flipout / plain-conv routes have no autograd node and must step aside (they used to look at weight.mean only)."""
    from bayesianneuralnetworks_amd.nn import NormalLinear, FlipOutNormalConv2d
    dev, bnn = env["dev"], env["bnn"]
    bnn.set_compute("f32")
    torch.manual_seed(4)
    lin = NormalLinear(64, 32).to(dev)
    lin.weight.mean.requires_grad_(False)
    lin.bias.mean.requires_grad_(False)
    x = torch.randn(128, 64, device=dev)                      # >= 64 rows: eligible for the three-plane path at inference
    y = lin(x)
    assert y.requires_grad
    y.square().sum().backward()
    assert lin.weight.scale.grad is not None and lin.weight.scale.grad.abs().sum() > 0
    assert lin.bias.scale.grad is not None and lin.weight.mean.grad is None
    with torch.no_grad():
        assert not lin(x).requires_grad                       # ... and with no gradient wanted the inference path runs as before
    conv = FlipOutNormalConv2d(64, 64, 3, stride=2, padding=1).to(dev)
    conv.weight.mean.requires_grad_(False)
    xc = torch.randn(8, 64, 6, 6, device=dev)
    for mode in ("f32", "bf16"):
        bnn.set_compute(mode)
        try:
            conv.weight.scale.grad = None
            yc = conv(xc)
            assert yc.requires_grad, mode
            yc.square().sum().backward()
            assert conv.weight.scale.grad is not None and conv.weight.scale.grad.abs().sum() > 0, mode
        finally:
            bnn.set_compute("f32")


def _random_x3_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        Nn = int(rng.choice([8, 16, 24, 80, 88, 160, 168, 200, 264, 400]))
        K = 8 * int(rng.randint(1, 70))
        planes = bool(rng.randint(2)) and Nn > 16
        out.append((int(rng.choice([1, 2, 3, 5, 8])), int(rng.choice([1, 16, 65, 130, 257, 300])), Nn, K, bool(rng.randint(2)),
                    bool(rng.randint(2)), planes))
    return out


@pytest.mark.parametrize("S,M,Nn,K,shared,relu,planes_out", _random_x3_shapes(18, 77))
def test_dense_x3_random_shapes(env, S, M, Nn, K, shared, relu, planes_out):
    """A seeded sweep of ragged shapes over the three-plane dense kernel's tiles and the K-split head kernel (N <= 16)."""
    test_dense_x3_vs_double(env, S, M, Nn, K, shared, relu, planes_out)


def _random_x3_head_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        Nn = int(rng.choice([88, 96, 104, 120, 160, 240, 320, 400, 1200]))          # the 160-column tiles: N % 80 == 0 or N < 128, N > 80
        out.append((int(rng.choice([1, 2, 4, 8])), int(rng.choice([1, 40, 129, 300])), Nn, 8 * int(rng.randint(1, 50)), int(rng.randint(1, 17)),
                    bool(rng.randint(2)), bool(rng.randint(2))))
    return out


@pytest.mark.parametrize("S,M,Nn,K,Nh,shared,relu", _random_x3_head_shapes(10, 99))
def test_fused_head_x3_random_shapes(env, S, M, Nn, K, Nh, shared, relu):
    test_fused_head_x3_vs_double(env, S, M, Nn, K, Nh, shared, relu)
