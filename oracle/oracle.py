"""ctypes/numpy front-end of oracle/bnn_oracle.c -- TEST INFRASTRUCTURE ONLY.

Every wrapper names the C function it calls; the C function carries the
reference file:line it restates.  The shared object is built by
`make -C oracle` (also done by __graft_entry__.build()).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i64 = ctypes.c_int64


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "bnn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_sigma_vec.argtypes = [_f32p, _i64, _f32p]
        L.orc_sample_affine.argtypes = [_f32p, _f32p, _f32p, _i64, _f32p]
        L.orc_linear.argtypes = [_f32p, _f32p, _f32p, _i64, _i64, _i64, _f32p]
        L.orc_conv2d.argtypes = [_f32p, _f32p, _f32p] + [_i64] * 14 + [_f32p]
        L.orc_kl_sum.argtypes = [_f32p, _f32p, _i64, ctypes.c_float, ctypes.c_float]
        L.orc_kl_sum.restype = ctypes.c_double
        L.orc_kl_divergence.argtypes = [_f64p, _i64p, _i64, ctypes.c_double]
        L.orc_kl_divergence.restype = ctypes.c_float
        L.orc_sample_affine_bwd.argtypes = [_f32p, _f32p, _f32p, _i64, _f32p, _f32p]
        L.orc_kl_bwd.argtypes = [_f32p, _f32p, _i64, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, _f32p, _f32p]
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_eps_fill.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                   ctypes.c_uint32, ctypes.c_uint32, _i64, _f32p]
        L.orc_philox4x32_r.argtypes = [_u32p, _u32p, ctypes.c_int, _u32p]
        L.orc_eps_fill_gen.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, _i64, _f32p]
        L.orc_bf16_round.argtypes = [ctypes.c_float]
        L.orc_bf16_round.restype = ctypes.c_float
        _lib = L
    return _lib


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def sigma(rho):
    """orc_sigma_vec -- core.py:25-27."""
    r, rp = _c(rho)
    out = np.empty_like(r)
    lib().orc_sigma_vec(rp, r.size, out.ctypes.data_as(_f32p))
    return out


def sample_affine(mu, rho, eps):
    """orc_sample_affine -- core.py:44-45."""
    m, mp = _c(mu)
    r, rp = _c(rho)
    e, ep = _c(eps)
    assert m.shape == r.shape == e.shape
    out = np.empty_like(m)
    lib().orc_sample_affine(mp, rp, ep, m.size, out.ctypes.data_as(_f32p))
    return out


def linear(x, w, b=None):
    """orc_linear -- dense.py:56-60."""
    x2, xp = _c(x)
    w2, wp = _c(w)
    lead = x2.shape[:-1]
    K = x2.shape[-1]
    M = int(np.prod(lead)) if lead else 1
    N = w2.shape[0]
    assert w2.shape == (N, K)
    bp = None
    if b is not None:
        b2, bp = _c(b)
        assert b2.shape == (N,)
    y = np.empty((M, N), dtype=np.float32)
    lib().orc_linear(xp, wp, bp, M, N, K, y.ctypes.data_as(_f32p))
    return y.reshape(*lead, N)


def conv2d(x, w, b=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    """orc_conv2d -- conv.py:112-119."""
    x4, xp = _c(x)
    w4, wp = _c(w)
    B, C, H, W = x4.shape
    O, Cg, KH, KW = w4.shape
    assert Cg * groups == C and O % groups == 0
    bp = None
    if b is not None:
        b1, bp = _c(b)
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    OH = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    OW = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    y = np.empty((B, O, OH, OW), dtype=np.float32)
    lib().orc_conv2d(xp, wp, bp, B, C, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw, groups,
                     y.ctypes.data_as(_f32p))
    return y


def kl_sum(mu, rho, prior_mu=0.0, prior_sigma=0.1):
    """orc_kl_sum -- loss.py:16-28 (sum; .mean() = sum / numel)."""
    m, mp = _c(mu)
    r, rp = _c(rho)
    return float(lib().orc_kl_sum(mp, rp, m.size, prior_mu, prior_sigma))


def kl_divergence(tensors, n_batches=1.0):
    """orc_kl_divergence -- loss.py:30-38.  tensors: iterable of
    (mu, rho, prior_mu, prior_sigma) in the reference's traversal order."""
    sums, numels = [], []
    for mu, rho, pm, ps in tensors:
        sums.append(kl_sum(mu, rho, pm, ps))
        numels.append(int(np.asarray(mu).size))
    if not sums:
        raise ValueError('KLDivergence was not able to find BayasianModules')
    s = np.asarray(sums, dtype=np.float64)
    n = np.asarray(numels, dtype=np.int64)
    return float(lib().orc_kl_divergence(s.ctypes.data_as(_f64p), n.ctypes.data_as(_i64p),
                                         len(sums), float(n_batches)))


def sample_affine_bwd(gw, rho, eps):
    """orc_sample_affine_bwd -- autograd of core.py:44-45."""
    g, gp = _c(gw)
    r, rp = _c(rho)
    e, ep = _c(eps)
    gmu = np.empty_like(g)
    grho = np.empty_like(g)
    lib().orc_sample_affine_bwd(gp, rp, ep, g.size, gmu.ctypes.data_as(_f32p),
                                grho.ctypes.data_as(_f32p))
    return gmu, grho


def linear_bwd(mu, rho, x, gy, eps, shared_x=False, rounder=None):
    """Gradients autograd derives through F.linear (nn/dense.py:60) and w = mu + sigma * eps
    (nn/core.py:44-45) for S sampled weights, in float64 (examples/MNIST/train.py:63-65 is the caller):
        dW_s = gy_s^T x_s;  g_mu = sum_s dW_s;  g_rho = sum_s dW_s * eps_s * sigmoid(rho);  gx_s = gy_s W_s.
    eps: list of S arrays shaped like mu.  rounder (e.g. bf16_round) is applied to x, gy and W_s first
    (the bf16 compute mode's operands).  -> (g_mu, g_rho, gx (S, M, K))."""
    r = rounder or (lambda a: a)
    mu = np.asarray(mu, np.float32)
    rho = np.asarray(rho, np.float32)
    g_mu = np.zeros(mu.shape, np.float64)
    g_rho = np.zeros(mu.shape, np.float64)
    gx = []
    for s in range(len(eps)):
        xs = r(np.asarray(x if shared_x else x[s], np.float32)).astype(np.float64)
        g = r(np.asarray(gy[s], np.float32)).astype(np.float64)
        w = r(sample_affine(mu, rho, eps[s])).astype(np.float64)
        gw = g.T @ xs
        dmu, drho = sample_affine_bwd(np.ones(mu.shape, np.float32), rho, eps[s])      # dw/dmu, dw/drho
        g_mu += gw * dmu
        g_rho += gw * drho
        gx.append(g @ w)
    return g_mu, g_rho, np.stack(gx)


def prune_score(mu, rho):
    """PruneNormal's ranking score, prune/prune.py:11: Normal(mu, sigma).log_prob(0), sigma as in core.py:25-27 (float64)."""
    sg = sigma(rho).astype(np.float64)
    m = np.asarray(mu, np.float64)
    return -0.5 * (m / sg) ** 2 - np.log(sg) - 0.5 * np.log(2.0 * np.pi)


def kl_bwd(mu, rho, prior_mu, prior_sigma, scale):
    """orc_kl_bwd -- autograd of loss.py:28 times `scale`."""
    m, mp = _c(mu)
    r, rp = _c(rho)
    gmu = np.empty_like(m)
    grho = np.empty_like(m)
    lib().orc_kl_bwd(mp, rp, m.size, prior_mu, prior_sigma, scale,
                     gmu.ctypes.data_as(_f32p), grho.ctypes.data_as(_f32p))
    return gmu, grho


def philox4x32_10(ctr, key):
    """orc_philox4x32_10 -- published Philox4x32-10."""
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data_as(_u32p), k.ctypes.data_as(_u32p),
                            out.ctypes.data_as(_u32p))
    return out


def philox4x32_r(ctr, key, rounds):
    """orc_philox4x32_r -- `rounds` rounds of the Philox4x32 round function (7: Philox4x32-7)."""
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_r(c.ctypes.data_as(_u32p), k.ctypes.data_as(_u32p), int(rounds), out.ctypes.data_as(_u32p))
    return out


def eps_fill(seed, stream, sample, epoch_host, epoch_dev, shape, gen=0):
    """orc_eps_fill_gen -- the build's own Philox/Box-Muller eps stream (RNG contract in include/bnn_hip.h);
    gen: 0 = BNN_GEN_PHILOX10_U24 (the default stream), 1 = BNN_GEN_PHILOX7_U16 (a DrawKey's .gen)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    lib().orc_eps_fill_gen(int(seed) & (2**64 - 1), stream, sample & 0xFFFFFFFF, epoch_host, epoch_dev & 0xFFFFFFFF, int(gen), n,
                           out.ctypes.data_as(_f32p))
    return out.reshape(shape)


def eps_for_key(key, s, epoch_dev, shape):
    """eps of MC sample `s` of the draw a DrawKey (seed, stream, sample0, epoch_host, epoch_dev_delta, gen) names, with the
    device epoch word at `epoch_dev`."""
    return eps_fill(key.seed, key.stream, key.sample0 + s, key.epoch_host, epoch_dev + key.epoch_dev_delta, shape,
                    getattr(key, "gen", 0))


def bf16_round(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def mc_forward(layers, x, eps_per_sample):
    """BayesianNetworkModule.forward, container.py:32-37: the serial MC loop.

    layers: list of ('linear', mu_w, rho_w, mu_b, rho_b) / ('relu',) entries;
    eps_per_sample[s] = list of (eps_w, eps_b) per linear layer.
    Returns the list of S outputs (bare array when S == 1, utils.py:10-11)."""
    outs = []
    for eps_list in eps_per_sample:
        h = np.asarray(x, dtype=np.float32)
        li = 0
        for L in layers:
            if L[0] == 'linear':
                _, mw, rw, mb, rb = L
                ew, eb = eps_list[li]
                li += 1
                w = sample_affine(mw, rw, ew)
                b = sample_affine(mb, rb, eb) if mb is not None else None
                h = linear(h, w, b)
            elif L[0] == 'relu':
                h = np.maximum(h, 0.0)
            else:
                raise ValueError(L[0])
        outs.append(h)
    return outs[0] if len(outs) == 1 else outs
