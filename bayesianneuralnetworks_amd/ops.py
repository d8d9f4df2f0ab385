"""Tensor-level front-end of the C-ABI (include/bnn_hip.h) with autograd.

Forward passes are hand-written HIP (libbnn_hip.so).  Backward: the eps-dependent
elementwise parts are HIP too (bnn_sample_affine_bwd, bnn_kl_backward; eps is regenerated
from the draw key, never stored); the two plain GEMMs of a linear/conv backward go through
torch.bmm / torch's conv backward on the GPU (library GEMMs on already-drawn weights).

Every function here requires CUDA (HIP) tensors and raises BnnHipError otherwise.
"""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import (BnnHipError, Rng, KlTensor, Conv2dShape, ptr, stream_ptr, check, require_cuda_f32,
                   require_cuda_act)
from ._rng import default_generator, DrawKey


def _rng_struct(key, device):
    r = Rng()
    r.seed = key.seed
    r.stream = key.stream
    r.sample0 = key.sample0
    r.epoch_host = key.epoch_host
    r.epoch_dev_delta = key.epoch_dev_delta
    r.epoch_dev = default_generator.epoch_dev(device).data_ptr()
    r.generator = key.gen
    return r


def _compute_code(compute):
    if compute in ("f32", "fp32", _lib.COMPUTE_F32):
        return _lib.COMPUTE_F32
    if compute in ("bf16", _lib.COMPUTE_BF16):
        return _lib.COMPUTE_BF16
    raise ValueError("compute must be 'f32' or 'bf16', got %r" % (compute,))


# --------------------------------------------------------------------------- K1
def sigma(rho):
    """1e-10 + softplus(rho)  (WeightNormal.stddev, core.py:25-27) -- no autograd."""
    rho = rho.detach()
    require_cuda_f32(rho, "rho")
    out = torch.empty_like(rho)
    check(_lib.load().bnn_sigma(ptr(rho), ptr(out), rho.numel(), stream_ptr(rho.device)), "bnn_sigma")
    return out


def eps_philox(shape, key, device):
    """The raw eps stream for `key` -> (nsamples, *shape) fp32."""
    n = 1
    for d in shape:
        n *= d
    out = torch.empty((key.nsamples,) + tuple(shape), dtype=torch.float32, device=device)
    r = _rng_struct(key, out.device)
    check(_lib.load().bnn_eps_philox(ptr(out), n, key.nsamples, n, ctypes.byref(r), stream_ptr(out.device)),
          "bnn_eps_philox")
    return out


def _sample_affine_philox_raw(mu, rho, key, out_dtype=torch.float32):
    require_cuda_f32(mu, "mu")
    require_cuda_f32(rho, "rho")
    n = mu.numel()
    out = torch.empty((key.nsamples,) + tuple(mu.shape), dtype=out_dtype, device=mu.device)
    r = _rng_struct(key, mu.device)
    code = _lib.F32 if out_dtype == torch.float32 else _lib.BF16
    check(_lib.load().bnn_sample_affine_philox(ptr(mu), ptr(rho), ptr(out), n, key.nsamples, n, code,
                                                ctypes.byref(r), stream_ptr(mu.device)),
          "bnn_sample_affine_philox")
    return out


def _sample_affine_bwd_raw(g_w, rho, n, nsamples, eps=None, key=None):
    g_w = g_w.contiguous()
    g_mu = torch.empty_like(rho)
    g_rho = torch.empty_like(rho)
    r = _rng_struct(key, rho.device) if key is not None else None
    check(_lib.load().bnn_sample_affine_bwd(ptr(g_w), n, ptr(rho), ptr(eps), n,
                                             ctypes.byref(r) if r is not None else None,
                                             n, nsamples, ptr(g_mu), ptr(g_rho), 0, stream_ptr(rho.device)),
          "bnn_sample_affine_bwd")
    return g_mu, g_rho


class _SampleAffineEps(torch.autograd.Function):
    """w = mu + sigma(rho) * eps with eps supplied (parity mode)."""

    @staticmethod
    def forward(ctx, mu, rho, eps):
        require_cuda_f32(mu, "mu")
        require_cuda_f32(rho, "rho")
        require_cuda_f32(eps, "eps")
        out = torch.empty_like(mu)
        check(_lib.load().bnn_sample_affine_eps(ptr(mu), ptr(rho), ptr(eps), ptr(out), mu.numel(), _lib.F32,
                                                 stream_ptr(mu.device)), "bnn_sample_affine_eps")
        ctx.save_for_backward(rho, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        rho, eps = ctx.saved_tensors
        g_mu, g_rho = _sample_affine_bwd_raw(g, rho, rho.numel(), 1, eps=eps)
        return g_mu, g_rho, None


class _SampleAffinePhilox(torch.autograd.Function):
    """(S, *shape) draws of w = mu + sigma(rho) * eps(key)."""

    @staticmethod
    def forward(ctx, mu, rho, key):
        out = _sample_affine_philox_raw(mu.detach(), rho.detach(), key)
        ctx.save_for_backward(rho)
        ctx.key = key
        return out

    @staticmethod
    def backward(ctx, g):
        (rho,) = ctx.saved_tensors
        g_mu, g_rho = _sample_affine_bwd_raw(g, rho, rho.numel(), ctx.key.nsamples, key=ctx.key)
        return g_mu, g_rho, None


def sample_affine_eps(mu, rho, eps):
    return _SampleAffineEps.apply(mu.contiguous(), rho.contiguous(), eps.contiguous())


def sample_affine_philox(mu, rho, key):
    return _SampleAffinePhilox.apply(mu.contiguous(), rho.contiguous(), key)


# --------------------------------------------------------------------------- draw-once path (bf16 compute mode)
def _pad64(k):
    return (k + 63) // 64 * 64


def dense_eligible(mu_w):
    """The draw-once path takes a (N, K) posterior whose rows are whole 8-column groups and 16-B aligned."""
    return mu_w.dim() == 2 and mu_w.shape[1] % 8 == 0 and mu_w.data_ptr() % 16 == 0


class Predrawn:
    """The drawn weights of one layer for one forward: w (S, N, Kp) bf16 zero-padded to Kp = roundup(K, 64) -- or, drawn
    for the fp32 parity mode, (3, S, N, Kp): the three bf16 planes of the fp32 draw -- b (S, N) fp32 or None, and the
    DrawKeys they were drawn with.  `ready`: the side stream the draw was launched on
    (the consumer's stream waits for it before the contraction), or None."""
    __slots__ = ("w", "b", "key_w", "key_b", "ready")

    def __init__(self, w, b, key_w, key_b, ready=None):
        self.w, self.b, self.key_w, self.key_b, self.ready = w, b, key_w, key_b, ready

    def wait(self):
        """Make the current stream wait for the draw (once: later consumers of the same side-stream launch are ordered
        behind this one on the same stream)."""
        if self.ready is not None:
            torch.cuda.current_stream(self.w.device).wait_stream(self.ready)
            self.ready = None


_side_streams = {}


def side_stream(device):
    """One extra stream per device for work that runs BESIDE the main stream's launches (draw of the later layers under the
    first layers' contractions: the draw is VALU-bound, the dense GEMM DMA / MFMA-bound, and their workgroups fit a CU
    together)."""
    key = (device.type, device.index)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device)
        _side_streams[key] = st
    return st


def draw_layers(layers, nsamples, kl=None, stream=None, x3=False, split=None):
    """ONE launch (bnn_draw_multi) draws the weights and biases of every (mu_w, rho_w, mu_b, rho_b, key_w, key_b) in
    `layers` for `nsamples` MC samples -> list of Predrawn.  kl (a KlDeferred from kl_normal_begin(carry=True)): the
    launch also carries that KL's first pass.  At most 4 layers (8 tensors) per launch; more are split.
    stream: launch on that side stream (forked from the current one here; consumers join through Predrawn.wait()).
    x3: weights as three bf16 planes of the fp32 draw (BNN_BF16X3; the fp32 parity mode's dense path).
    split (x3 only): an fp32 (M, K) activation whose three-plane split (what split_x3 returns) rides in the FIRST launch as one more
    tensor (kind 3) instead of a launch of its own; the planes come back as `_tls.last_split`."""
    _tls.last_split = None
    out = []
    lib = _lib.load()
    dev = layers[0][0].device
    cur = torch.cuda.current_stream(dev)
    if stream is not None:
        stream.wait_stream(cur)                     # fork: the draw is ordered behind everything already on the main stream
    launch_stream = stream if stream is not None else cur
    sp = _lib.stream_ptr(dev)
    sp.value = launch_stream.cuda_stream
    step = 4 if split is None else 3                # (8 tensors per launch: the split takes one)
    for i0 in range(0, len(layers), step):
        group = layers[i0:i0 + step]
        arr = (_lib.DrawTensor * (2 * len(group) + 1))()
        n = 0
        keep = []
        if split is not None and i0 == 0 and x3:
            require_cuda_f32(split, "x")
            Ms, Ks = split.shape
            lds_ = _pad64(Ks)
            planes = torch.empty((3, 1, Ms, lds_), dtype=torch.bfloat16, device=dev)
            t = arr[n]
            t.mu, t.rho, t.rows, t.cols = split.data_ptr(), split.data_ptr(), Ms, Ks
            t.out, t.ld, t.out_sample_stride, t.out_dtype = planes.data_ptr(), lds_, Ms * lds_, _lib.BF16X3
            t.kind, t.taps = 3, 0
            n += 1
            _tls.last_split = planes
            keep.append(split)
        for spec in group:
            mu_w, rho_w, mu_b, rho_b, key_w, key_b = spec[:6]
            taps = spec[6] if len(spec) > 6 else 0      # KH * KW: a conv weight, written tap-major
            require_cuda_f32(mu_w, "weight.mean")
            require_cuda_f32(rho_w, "weight.scale")
            N, K = mu_w.shape
            kp = _pad64(K)
            w = torch.empty(((3, nsamples, N, kp) if x3 else (nsamples, N, kp)), dtype=torch.bfloat16, device=dev)
            t = arr[n]
            t.mu, t.rho, t.rows, t.cols = mu_w.data_ptr(), rho_w.data_ptr(), N, K
            t.out, t.ld, t.out_sample_stride, t.out_dtype = w.data_ptr(), kp, N * kp, (_lib.BF16X3 if x3 else _lib.BF16)
            t.taps = taps
            t.rng = _rng_struct(key_w, dev)
            n += 1
            b = None
            if mu_b is not None:
                require_cuda_f32(mu_b, "bias.mean")
                require_cuda_f32(rho_b, "bias.scale")
                b = torch.empty((nsamples, N), dtype=torch.float32, device=dev)
                t = arr[n]
                t.mu, t.rho, t.rows, t.cols = mu_b.data_ptr(), rho_b.data_ptr(), 1, N
                t.out, t.ld, t.out_sample_stride, t.out_dtype = b.data_ptr(), N, N, _lib.F32
                t.taps = 0
                t.rng = _rng_struct(key_b, dev)
                n += 1
            keep.append((mu_w, rho_w, mu_b, rho_b))
            out.append(Predrawn(w, b, key_w, key_b, ready=stream))
        carried = False
        if kl is not None and not kl.launched and i0 == 0 and kl.out.device == dev:
            rc = lib.bnn_draw_multi(arr, n, nsamples, kl.arr, kl.T, ptr(kl.ws), sp)
            if rc == 0:
                kl.launched = carried = True
            elif rc != _lib.E_UNSUPPORTED:
                check(rc, "bnn_draw_multi")
        if not carried:
            check(lib.bnn_draw_multi(arr, n, nsamples, None, 0, None, sp), "bnn_draw_multi")
    return out


def rows_pitch(t, K):
    """A (M, K) or (S, M, K) bf16 activation whose rows the dense kernel can read in place -- unit element stride, one row
    pitch (>= K, whole 16-B chunks), samples a whole number of 16-B chunks apart, e.g. the 128-B-aligned rows _dense_raw
    itself writes -> (row pitch, sample stride) in elements, the values to LAUNCH with; None if the view is not of that kind.
    The strides are read off the view, never re-derived from M and K: with one row per sample the row stride of the view says
    nothing (torch may report anything for a size-1 dimension) and the samples of a padded (S, 1, K) view sit stride(0)
    apart, not K."""
    if t.dtype != torch.bfloat16 or t.dim() not in (2, 3) or t.shape[-1] != K or t.stride(-1) != 1 or t.data_ptr() % 16 != 0:
        return None
    M = t.shape[-2]
    ld = t.stride(-2) if M > 1 else K               # a single row: its pitch is never used to step to another row
    if ld < K or ld % 8 != 0:
        return None
    if t.dim() == 2 or t.shape[0] == 1:
        return ld, M * ld
    xs = t.stride(0)
    if xs % 8 != 0 or xs < (M - 1) * ld + K:        # samples overlap, or are not 16-B aligned
        return None
    return ld, xs


def rows_regular(t, K):
    return rows_pitch(t, K) is not None


def _dense_raw(x2, x_sample_stride, M, pre, K, relu, out_dtype, ldx=None, pad_rows=False):
    """y (S, M, N) = act(x . w_s^T + b_s) on Predrawn weights (bnn_dense_forward); x bf16 with row pitch ldx.
    pad_rows (bf16 hidden activations): the rows of y are written 128 B apart-aligned (pitch roundup(N, 64)) and y is
    returned as a view of that buffer -- the next layer's LDS-DMA then reads whole cache lines (layer 2 of the BASELINE
    net: 19.6 -> 17.0 us)."""
    S, N, kp = pre.w.shape
    ldx = K if ldx is None else ldx
    ldy = _pad64(N) if (pad_rows and out_dtype == torch.bfloat16) else N
    ybuf = torch.empty((S, M, ldy), dtype=out_dtype, device=x2.device)
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_Y_BF16 if out_dtype == torch.bfloat16 else 0)
    check(_lib.load().bnn_dense_forward(ptr(x2), x_sample_stride, ldx, ptr(pre.w), N * kp, kp,
                                         ptr(pre.b), N if pre.b is not None else 0, ptr(ybuf), M * ldy, ldy, M, N, K, S, flags,
                                         stream_ptr(x2.device)), "bnn_dense_forward")
    return ybuf if ldy == N else ybuf[:, :, :N]


class HeadPartials:
    """What a hidden layer fused with the classifier head behind it leaves (bnn_dense_forward_head): PARTIAL logits
    (parts, S, M, Nh) fp32 -- the head's contraction cut along the hidden units, partial 0 carrying the head's bias.  Not a
    tensor: the head layer passes it through, `BayesianNetworkModule.predictive_mean` / `ops.mc_mean` reduce it over
    (part, sample) in the step's ONE tail launch, `.logits()` over `part` for callers that want every sample."""
    __slots__ = ("p", "head")

    def __init__(self, p, head=None):
        self.p, self.head = p, head

    @property
    def shape(self):
        return torch.Size((self.p.shape[1] * self.p.shape[2], self.p.shape[3]))

    @property
    def is_cuda(self):
        return True

    @property
    def device(self):
        return self.p.device

    def dim(self):
        return 2

    def logits(self):
        """(S, M, Nh): the sum over the partials (one bnn_mc_sum launch)."""
        parts, S, M, Nh = self.p.shape
        out = torch.empty((S, M, Nh), dtype=torch.float32, device=self.p.device)
        n = S * M * Nh
        check(_lib.load().bnn_mc_sum(ptr(self.p), n, parts, n, 1.0, ptr(out), 0, None, 0, stream_ptr(self.p.device)), "bnn_mc_sum")
        return out


def dense_head_eligible(M, N, pre, pre_head):
    """Hidden layer (drawn weights `pre`, N > 16 outputs) + head (`pre_head`, <= 16 outputs) in one launch?"""
    return (pre is not None and pre_head is not None and pre.w.dim() == 3 and pre_head.w.dim() == 3 and N > 16 and N % 8 == 0 and
            pre_head.w.shape[1] <= 16 and pre_head.w.shape[2] >= N and pre.w.shape[0] == pre_head.w.shape[0] and M > 0)


def _dense_head_raw(x2, x_sample_stride, M, pre, K, relu, pre_head, ldx=None):
    """bnn_dense_forward_head: act(x . w_s^T + b_s) rounded to bf16 and contracted with the head's drawn weights in the same
    launch -> HeadPartials (the hidden activation is never stored)."""
    S, N, kp = pre.w.shape
    _, Nh, kph = pre_head.w.shape
    lib = _lib.load()
    parts = lib.bnn_dense_head_parts(M, N, S)
    P = torch.empty((parts, S, M, Nh), dtype=torch.float32, device=x2.device)
    ldx = K if ldx is None else ldx
    check(lib.bnn_dense_forward_head(ptr(x2), x_sample_stride, ldx, ptr(pre.w), N * kp, kp, ptr(pre.b), N if pre.b is not None else 0,
                                     ptr(pre_head.w), Nh * kph, kph, ptr(pre_head.b), Nh if pre_head.b is not None else 0, Nh,
                                     ptr(P), M, N, K, S, _lib.FLAG_RELU if relu else 0, stream_ptr(x2.device)), "bnn_dense_forward_head")
    return HeadPartials(P)


def dense_head_x3_eligible(M, N, pre, pre_head):
    """The same pair in the fp32 parity mode (three-plane operands): the 160-column tiles only (N % 80 == 0 or N < 128, N > 80)."""
    return (pre is not None and pre_head is not None and pre.w.dim() == 4 and pre_head.w.dim() == 4 and N > 80 and N % 8 == 0 and
            (N % 80 == 0 or N < 128) and pre_head.w.shape[2] <= 16 and pre_head.w.shape[3] >= N and pre.w.shape[1] == pre_head.w.shape[1] and M > 0)


def _dense_head_raw_x3(xp, shared, M, pre, K, relu, pre_head):
    """bnn_dense_forward_x3_head: xp (3, S or 1, M, ldx) planes, pre.w (3, S, N, kp), pre_head.w (3, S, Nh, kph) -> HeadPartials."""
    _, S, N, kp = pre.w.shape
    _, _, Nh, kph = pre_head.w.shape
    ldx = xp.shape[3]
    lib = _lib.load()
    parts = lib.bnn_dense_head_parts(M, N, S)
    P = torch.empty((parts, S, M, Nh), dtype=torch.float32, device=xp.device)
    check(lib.bnn_dense_forward_x3_head(ptr(xp), xp.shape[1] * M * ldx, 0 if shared else M * ldx, ldx,
                                        ptr(pre.w), S * N * kp, N * kp, kp, ptr(pre.b), N if pre.b is not None else 0,
                                        ptr(pre_head.w), S * Nh * kph, Nh * kph, kph, ptr(pre_head.b), Nh if pre_head.b is not None else 0, Nh,
                                        ptr(P), M, N, K, S, _lib.FLAG_RELU if relu else 0, stream_ptr(xp.device)), "bnn_dense_forward_x3_head")
    return HeadPartials(P)


class X3Activation:
    """A hidden activation of the fp32 parity mode as it travels between two dense layers: the fp32 values as three bf16
    planes (3, S, M, ld) (h, m, l: v = h + m + l to 2^-24 |v|), `cols` valid columns.  Not a tensor: only NormalLinear
    consumes it (nn.fuse_activations arranges that); .float() gives the fp32 tensor back."""
    __slots__ = ("planes", "cols")

    def __init__(self, planes, cols):
        self.planes, self.cols = planes, cols

    @property
    def shape(self):
        return torch.Size((self.planes.shape[1] * self.planes.shape[2], self.cols))

    @property
    def is_cuda(self):
        return self.planes.is_cuda

    @property
    def device(self):
        return self.planes.device

    def dim(self):
        return 2

    def float(self):
        p = self.planes[..., :self.cols].float()
        return (p[0] + p[1] + p[2]).reshape(-1, self.cols)


def split_x3(x2):
    """fp32 (M, K) -> (3, 1, M, roundup(K, 64)) bf16 planes (bnn_split_bf16x3)."""
    require_cuda_f32(x2, "x")
    M, K = x2.shape
    ld = _pad64(K)
    out = torch.empty((3, 1, M, ld), dtype=torch.bfloat16, device=x2.device)
    check(_lib.load().bnn_split_bf16x3(ptr(x2), M, K, x2.stride(0) if M > 1 else K, ptr(out), ld, M * ld, stream_ptr(x2.device)),
          "bnn_split_bf16x3")
    return out


def x3_eligible(x2, mu_w, M):
    """fp32 parity mode on the dense kernels: whole 8-column groups, a batch worth the draw-once path, an fp32 (M, K) input
    or an X3Activation; a narrow layer (N <= 16: the K-split head kernel) up to K = 2048."""
    N, K = mu_w.shape
    if not (dense_eligible(mu_w) and M >= 64 and (N > 16 or K <= 2048)):
        return False
    if isinstance(x2, X3Activation):
        return True
    return x2.dtype == torch.float32 and x2.dim() == 2 and x2.stride(-1) == 1 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0


def _dense_raw_x3(xp, shared, M, pre, K, relu, planes_out):
    """y = act(x . w_s^T + b_s) in the fp32 parity mode on three-plane operands (bnn_dense_forward_x3).
    xp: (3, S or 1, M, ldx) planes; pre.w (3, S, N, kp).  planes_out: the result as an X3Activation (for the next dense
    layer) instead of an fp32 (S, M, N) tensor."""
    _, S, N, kp = pre.w.shape
    ldx = xp.shape[3]
    xs = 0 if shared else M * ldx
    if planes_out:
        ldy = _pad64(N)
        y = torch.empty((3, S, M, ldy), dtype=torch.bfloat16, device=xp.device)
        yps = S * M * ldy
    else:
        ldy = N
        y = torch.empty((S, M, N), dtype=torch.float32, device=xp.device)
        yps = 0
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_Y_BF16 if planes_out else 0)
    check(_lib.load().bnn_dense_forward_x3(ptr(xp), xp.shape[1] * M * ldx, xs, ldx, ptr(pre.w), S * N * kp, N * kp, kp,
                                            ptr(pre.b), N if pre.b is not None else 0, ptr(y), yps, M * ldy, ldy, M, N, K, S, flags,
                                            stream_ptr(xp.device)), "bnn_dense_forward_x3")
    return X3Activation(y, N) if planes_out else y


def linear_sampled_x3(x2, shared, M, mu_w, rho_w, mu_b, rho_b, key_w, key_b, relu=False, planes_out=False, predrawn=None, head_pre=None):
    """Inference call of NormalLinear in the fp32 parity mode (no autograd: the caller checked that no gradient is wanted):
    draw once as three bf16 planes (or `predrawn` by the network's draw plan), contract on the dense kernel.
    head_pre (the drawn planes of the classifier head behind this layer): the pair as ONE launch -> HeadPartials."""
    S = key_w.nsamples
    N, K = mu_w.shape
    pre = predrawn
    if pre is None:
        pre = draw_layers([(mu_w, rho_w, mu_b, rho_b, key_w, key_b)], S, kl=_tls.kl_carry, x3=True)[0]
        if _tls.kl_carry is not None and _tls.kl_carry.launched:
            _tls.kl_carry = None
    pre.wait()
    if isinstance(x2, X3Activation):
        xp = x2.planes
        if xp.shape[1] not in (1, S) or xp.shape[2] != M or x2.cols != K:
            raise BnnHipError("linear: three-plane activation of shape %s does not fit (S = %d, M = %d, K = %d)" % (tuple(xp.shape), S, M, K))
        shared = xp.shape[1] == 1
    else:
        xp = split_x3(x2.reshape(-1, K))            # (3, 1, rows, ld); rows = M (shared) or S * M
        if not shared:
            xp = xp.view(3, S, M, xp.shape[3])
    if head_pre is not None:
        head_pre.wait()
        return _dense_head_raw_x3(xp, shared, M, pre, K, relu, head_pre)
    return _dense_raw_x3(xp, shared, M, pre, K, relu, planes_out)


# --------------------------------------------------------------------------- K2 linear
def _linear_sampled_raw(x2, x_sample_stride, M, mu_w, rho_w, mu_b, rho_b, key_w, key_b, compute, relu=False,
                        out_dtype=torch.float32, predrawn=None, pad_rows=False):
    N, K = mu_w.shape
    S = key_w.nsamples
    _lib.ensure_workspace(x2.device)
    if compute == _lib.COMPUTE_BF16 and DRAW_ONCE_BF16 and dense_eligible(mu_w) and M > 0:
        # bf16 compute mode: the layer's weights are drawn ONCE for the S samples (K1, one launch -- or already drawn
        # for this forward by the network's draw plan, `predrawn`) and contracted by the dense MFMA GEMM (csrc/bnn_dense.hip).
        # Same DrawKeys -> the same weights as the fused kernel, bit for bit; the backward re-creates them as always.
        pre = predrawn
        if pre is None:
            pre = draw_layers([(mu_w, rho_w, mu_b, rho_b, key_w, key_b)], S, kl=_tls.kl_carry)[0]
            if _tls.kl_carry is not None and _tls.kl_carry.launched:
                _tls.kl_carry = None
        pre.wait()
        pitch = rows_pitch(x2, K) if x2.dtype == torch.bfloat16 else None
        if pitch is not None:
            ldx, xs = pitch
            return _dense_raw(x2, 0 if x_sample_stride == 0 else xs, M, pre, K, relu, out_dtype, ldx=ldx, pad_rows=pad_rows)
        xb = x2.contiguous().to(torch.bfloat16)        # (the fused kernel rounds its A operand the same way)
        return _dense_raw(xb, x_sample_stride, M, pre, K, relu, out_dtype, pad_rows=pad_rows)
    x2 = x2.contiguous()
    y = torch.empty((S, M, N), dtype=out_dtype, device=x2.device)
    rw = _rng_struct(key_w, x2.device)
    rb = _rng_struct(key_b, x2.device) if mu_b is not None else None
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_X_BF16 if x2.dtype == torch.bfloat16 else 0) | \
            (_lib.FLAG_Y_BF16 if out_dtype == torch.bfloat16 else 0)
    if (M >= DRAW_ONCE_MIN_ROWS and N > 16 and K % 8 == 0 and mu_w.data_ptr() % 16 == 0 and rho_w.data_ptr() % 16 == 0):
        # Large batch: the fused kernel's 256- / 512-row tiles would re-draw every weight M / 256 (M / 512) times, and at
        # that size the draw is what the launch is made of -- so the weights of the S samples are drawn ONCE by K1 (same
        # DrawKey -> the same values, bit for bit) and the same kernel runs on explicit weights (configs[4] layer, 4096 x
        # 4096 at batch 4096: fp32 mode 111 -> 144 TFLOP/s, bf16 mode 310 -> 369).  S * N * K * 4 bytes of scratch; the
        # backward re-creates the draws from the keys as always.
        w = _sample_affine_philox_raw(mu_w.reshape(-1), rho_w.reshape(-1), key_w)
        b = _sample_affine_philox_raw(mu_b, rho_b, key_b) if mu_b is not None else None
        check(_lib.load().bnn_linear_forward(ptr(x2), x_sample_stride, K, ptr(w), N * K, ptr(b), N, ptr(y), M * N, N, M, N, K, S,
                                              compute, flags, stream_ptr(x2.device)), "bnn_linear_forward")
        return y
    h = _tls.kl_carry
    if h is not None and N <= 16 and not h.launched and h.out.device == x2.device:
        # a narrow layer: its launch carries the first pass of the KL begun with kl_normal_begin(carry=True)
        check(_lib.load().bnn_linear_forward_sampled_kl(
            ptr(x2), x_sample_stride, K, ptr(mu_w), ptr(rho_w), ptr(mu_b), ptr(rho_b),
            ptr(y), M * N, N, M, N, K, S, ctypes.byref(rw), ctypes.byref(rb) if rb is not None else None,
            compute, flags, h.arr, h.T, ptr(h.ws), stream_ptr(x2.device)), "bnn_linear_forward_sampled_kl")
        h.launched = True
        _tls.kl_carry = None
        return y
    check(_lib.load().bnn_linear_forward_sampled(
        ptr(x2), x_sample_stride, K, ptr(mu_w), ptr(rho_w), ptr(mu_b), ptr(rho_b),
        ptr(y), M * N, N, M, N, K, S, ctypes.byref(rw), ctypes.byref(rb) if rb is not None else None,
        compute, flags, stream_ptr(x2.device)), "bnn_linear_forward_sampled")
    return y


class _ThreadState(threading.local):
    """Per-thread hand-over state (two threads driving two models must not see each other's):
    kl_carry   -- a KlDeferred whose first pass waits for a narrow layer's launch to carry it (kl_normal_begin(carry=True));
    kl_pending -- KL gradients parked for the layers' weight-gradient launches (FUSE_KL_GRADIENT), keyed by the mean's
                  storage; a backward pass runs its nodes on the thread that called backward() for its device."""

    def __init__(self):
        self.kl_carry = None
        self.kl_pending = {}
        self.last_split = None      # draw_layers(split=...): the three planes of the activation that rode in the last launch


_tls = _ThreadState()
# from this many rows per sample on, a sampled linear layer draws its weights once (K1) instead of in the GEMM
DRAW_ONCE_MIN_ROWS = 2048
# bf16 compute mode: draw once + dense GEMM at every batch size (False: the round-1 fused kernel, kept for A/B runs)
DRAW_ONCE_BF16 = True
DENSE_X3_F32 = os.environ.get("BNN_DENSE_X3", "1") != "0"    # fp32 parity mode of wide inference layers on the dense kernel
# BNN_DRAW_SIDE=1: a network draw plan launches the layers after the first on a side stream, beside the first layer draw + contraction
# (measured on the BASELINE step: SLOWER, 0.0832 vs 0.0749 ms -- the cross-queue dependency costs more than the overlap hides; off)
DRAW_SIDE_STREAM = os.environ.get("BNN_DRAW_SIDE", "0") == "1"


def _bf(t):
    return t.dtype == torch.bfloat16


# --- KL gradient fused into the weight-gradient launch ------------------------------------------------
# KLDivergence's backward does not compute its gradient when the parameter's layer can do it for free: it
# PARKS (upstream, scale, prior) per posterior tensor here, keyed by the mean's storage; the sampled linear /
# conv backward of that layer -- which runs later in the same backward pass -- picks the entry up and the
# weight-gradient kernel adds the closed-form KL gradient in its final store (no bnn_kl_backward pass, no
# autograd accumulation add per parameter).  Entries nobody picked up (parity-mode layers, unused layers, a
# KL node that happened to run after the layers) are flushed at the end of the pass by an engine callback.
# OPT-IN (nn.fuse_kl_gradient(True)): parking returns no gradient from the KL node itself, which is only right when
# the pass accumulates into .grad (loss.backward()); torch.autograd.grad(kl, params) needs the default path.
FUSE_KL_GRADIENT = False


class _KlPending:
    __slots__ = ("up", "scale", "prior", "mu", "rho")

    def __init__(self, up, scale, prior, mu, rho):
        self.up, self.scale, self.prior, self.mu, self.rho = up, scale, prior, mu, rho


def _kl_take(mu):
    pend = _tls.kl_pending
    return pend.pop((mu.device.index, mu.data_ptr()), None) if pend else None


def _kl_fuse_struct(ent_w, ent_b):
    """bnn_kl_fuse_t for a layer whose weight entry (and optionally bias entry) were parked by the same KL node."""
    k = _lib.KlFuse()
    k.upstream = ent_w.up.data_ptr()
    k.mu_w = ent_w.mu.data_ptr()
    k.scale_w, k.prior_mu_w, k.prior_sigma_w = ent_w.scale, ent_w.prior[0], ent_w.prior[1]
    if ent_b is not None:
        k.mu_b = ent_b.mu.data_ptr()
        k.scale_b, k.prior_mu_b, k.prior_sigma_b = ent_b.scale, ent_b.prior[0], ent_b.prior[1]
    else:
        k.mu_b, k.scale_b, k.prior_mu_b, k.prior_sigma_b = None, 0.0, 0.0, 1.0
    return k


def _kl_flush():
    """Engine callback at the end of a backward pass: KL gradients of the tensors no layer picked up."""
    pend = _tls.kl_pending
    if not pend:
        return
    ents = list(pend.values())
    pend.clear()
    lib = _lib.load()
    for e in ents:
        g_mu, g_rho = torch.empty_like(e.mu), torch.empty_like(e.rho)
        arr = _kl_descs([e.mu], [e.rho], [e.prior])
        gm = (ctypes.c_void_p * 1)(g_mu.data_ptr())
        gr = (ctypes.c_void_p * 1)(g_rho.data_ptr())
        # scale = 1 / (n * T * n_batches): hand bnn_kl_backward an n_batches that reproduces it for one tensor
        check(lib.bnn_kl_backward(arr, 1, 1.0 / (e.scale * e.mu.numel()), ptr(e.up), gm, gr, 0, stream_ptr(e.mu.device)),
              "bnn_kl_backward")
        for p_, g_ in ((e.mu, g_mu), (e.rho, g_rho)):
            if p_.grad is None:
                p_.grad = g_
            else:
                p_.grad.add_(g_)


def _relu_backward_raw(g, y):
    """g * (y > 0) (the fused ReLU epilogue's backward)."""
    out = torch.empty_like(g)
    flags = (_lib.FLAG_X_BF16 if _bf(g) else 0) | (_lib.FLAG_Y_BF16 if _bf(y) else 0)
    check(_lib.load().bnn_relu_backward(ptr(g), ptr(y), ptr(out), g.numel(), flags, stream_ptr(g.device)),
          "bnn_relu_backward")
    return out


def _colsum_raw(gy):
    """(S, M, N) -> (S, N) fp32 column sums (the bias gradient of F.linear)."""
    S, M, N = gy.shape
    out = torch.empty((S, N), dtype=torch.float32, device=gy.device)
    check(_lib.load().bnn_colsum(ptr(gy), M * N, N, ptr(out), M, N, S, _lib.FLAG_X_BF16 if _bf(gy) else 0,
                                 stream_ptr(gy.device)), "bnn_colsum")
    return out


def _dgrad_plain_raw(gy, w, x_dtype):
    """gx[s] = gy[s] @ w[s] with explicit fp32 weights (S, N, K)."""
    S, M, N = gy.shape
    K = w.shape[2]
    gx = torch.empty((S, M, K), dtype=x_dtype, device=gy.device)
    flags = (_lib.FLAG_X_BF16 if _bf(gy) else 0) | (_lib.FLAG_Y_BF16 if x_dtype == torch.bfloat16 else 0)
    check(_lib.load().bnn_linear_backward_input(ptr(gy), M * N, N, ptr(w), N * K, ptr(gx), M * K, K, M, N, K, S,
                                                flags, stream_ptr(gy.device)), "bnn_linear_backward_input")
    return gx


DGRAD_ON_DRAWN = True      # bf16 mode: the input gradient contracts on the weights the forward drew (A/B switch)


def _transpose_drawn_raw(w, K):
    """Drawn weights (S, N, ldw) bf16 -> (S, K, roundup(N, 64)) with zeros beyond column N (bnn_transpose_bf16)."""
    S, N, ldw = w.shape
    ldn = _pad64(N)
    out = torch.empty((S, K, ldn), dtype=torch.bfloat16, device=w.device)
    check(_lib.load().bnn_transpose_bf16(ptr(w), N * ldw, ldw, ptr(out), K * ldn, ldn, N, K, S, stream_ptr(w.device)),
          "bnn_transpose_bf16")
    return out


def _dgrad_drawn_raw(gy, w, K):
    """gx[s] = gy[s] @ w_s on the weights the forward drew (bf16 (S, N, ldw)): transpose once, then the dense kernel
    (contraction over n) -- the backward on the draw-once path, no second draw.  gy (S, M, N) bf16 -> gx (S, M, K) bf16."""
    S, M, N = gy.shape
    if gy.data_ptr() % 16 != 0:                      # (a contiguous view at an odd offset: the LDS-DMA wants 16-B aligned rows)
        gy = gy.clone()
    wt = _transpose_drawn_raw(w, K)
    gx = torch.empty((S, M, K), dtype=torch.bfloat16, device=gy.device)
    check(_lib.load().bnn_dense_forward(ptr(gy), M * N, N, ptr(wt), K * wt.shape[2], wt.shape[2], None, 0, ptr(gx), M * K, K,
                                         M, K, N, S, _lib.FLAG_Y_BF16, stream_ptr(gy.device)), "bnn_dense_forward")
    return gx


def _sum_samples(t):
    """(S, ...) fp32 -> sum over S (bnn_mc_sum, scale 1)."""
    out = torch.empty(t.shape[1:], dtype=torch.float32, device=t.device)
    n = out.numel()
    check(_lib.load().bnn_mc_sum(ptr(t), n, t.shape[0], n, 1.0, ptr(out), 0, None, 0, stream_ptr(t.device)),
          "bnn_mc_sum")
    return out


class _SampledLinear(torch.autograd.Function):
    """y[s] = x[s] @ w_s^T + b_s, w_s / b_s drawn in-kernel (NormalLinear.forward, dense.py:56-60).
    Backward (what autograd derives in the reference, train.py:63-65) is HIP as well: the input
    gradient re-draws w_s inside the contraction, the weight gradient folds the draw's backward
    into its epilogue (csrc/bnn_linear_bwd.hip)."""

    @staticmethod
    def forward(ctx, x, mu_w, rho_w, mu_b, rho_b, key_w, key_b, shared_x, compute, relu, out_dtype, predrawn=None, track=True):
        # x: (M, K) shared by all samples, or (S, M, K), fp32 or (bf16 compute mode) bf16;
        # relu: max(., 0) fused in the epilogue; out_dtype: fp32, or bf16 for a hidden activation
        # x may be a row-padded view (rows_regular) on the inference path; whatever the backward saves is contiguous
        # track: grad mode at the call (inside forward it is always off, and needs_input_grad only mirrors requires_grad)
        needs_grad = track and any(ctx.needs_input_grad[:5])
        if needs_grad or not (x.dtype == torch.bfloat16 and rows_regular(x, x.shape[-1])):
            x = x.contiguous()
        require_cuda_act(x, "x", contiguous=False)
        if (x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16) and compute != _lib.COMPUTE_BF16:
            raise BnnHipError("bf16 activations need compute mode 'bf16'")
        for t, n in ((mu_w, "weight.mean"), (rho_w, "weight.scale")):
            require_cuda_f32(t, n)
        if mu_b is not None:
            require_cuda_f32(mu_b, "bias.mean")
            require_cuda_f32(rho_b, "bias.scale")
        M = x.shape[-2]
        K = x.shape[-1]
        if K != mu_w.shape[1]:
            raise BnnHipError("linear: input has %d features, weight expects %d" % (K, mu_w.shape[1]))
        ctx.drawn_w = None
        if (track and ctx.needs_input_grad[0] and not shared_x and compute == _lib.COMPUTE_BF16 and DRAW_ONCE_BF16 and DGRAD_ON_DRAWN
                and dense_eligible(mu_w) and M > 0 and x.dtype == torch.bfloat16 and mu_w.shape[0] > 16 and mu_w.shape[0] % 8 == 0):
            # training, bf16 mode: the input gradient will contract on THESE drawn weights (kept alive by the graph node)
            if predrawn is None:
                predrawn = draw_layers([(mu_w, rho_w, mu_b, rho_b, key_w, key_b)], key_w.nsamples, kl=_tls.kl_carry)[0]
                if _tls.kl_carry is not None and _tls.kl_carry.launched:
                    _tls.kl_carry = None
            if predrawn.w.dim() == 3:
                ctx.drawn_w = predrawn.w
        y = _linear_sampled_raw(x, 0 if shared_x else M * K, M, mu_w, rho_w, mu_b, rho_b, key_w, key_b,
                                compute, relu, out_dtype, predrawn, pad_rows=not needs_grad)
        ctx.save_for_backward(x, mu_w, rho_w, rho_b if mu_b is not None else None, y if relu else None, mu_b)
        ctx.key_w, ctx.key_b, ctx.shared_x, ctx.compute = key_w, key_b, shared_x, compute
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mu_w, rho_w, rho_b, y_relu, mu_b = ctx.saved_tensors
        S, compute = ctx.key_w.nsamples, ctx.compute
        N, K = mu_w.shape
        M = x.shape[-2]
        dev = gy.device
        lib = _lib.load()
        _lib.ensure_workspace(dev)
        gy = gy.contiguous()
        if compute != _lib.COMPUTE_BF16 and gy.dtype != torch.float32:
            gy = gy.float()
        if y_relu is not None:
            gy = _relu_backward_raw(gy, y_relu)                            # fused ReLU
        gflag = _lib.FLAG_X_BF16 if _bf(gy) else 0
        rw = _rng_struct(ctx.key_w, dev)
        gx = g_mu_w = g_rho_w = g_mu_b = g_rho_b = None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if N <= 16 and K % 4 == 0 and M > 0 and need_w and not _bf(gy) and not (ctx.shared_x and ctx.needs_input_grad[0]):
            # narrow layer (classifier head): the whole backward in one pass over the activations
            need_b = rho_b is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
            g_mu_w, g_rho_w = torch.empty_like(mu_w), torch.empty_like(rho_w)
            rb = None
            if need_b:
                g_mu_b, g_rho_b = torch.empty_like(rho_b), torch.empty_like(rho_b)
                rb = _rng_struct(ctx.key_b, dev)
            if ctx.needs_input_grad[0]:
                gx = torch.empty((S, M, K), dtype=x.dtype, device=dev)
            ent_w = _kl_take(mu_w)
            ent_b = _kl_take(mu_b) if (ent_w is not None and need_b and mu_b is not None) else None
            kl = _kl_fuse_struct(ent_w, ent_b) if ent_w is not None else None
            flags = (_lib.FLAG_X_BF16 if _bf(x) else 0) | (_lib.FLAG_Y_BF16 if (gx is not None and _bf(gx)) else 0)
            rc = lib.bnn_linear_backward_narrow_sampled(
                ptr(x), 0 if ctx.shared_x else M * K, K, ptr(gy), M * N, N, ptr(mu_w), ptr(rho_w), ptr(gx), M * K, K,
                ptr(g_mu_w), ptr(g_rho_w), ptr(rho_b) if need_b else None, ptr(g_mu_b), ptr(g_rho_b), M, N, K, S,
                ctypes.byref(rw), ctypes.byref(rb) if rb is not None else None,
                ctypes.byref(kl) if kl is not None else None, flags, 0, stream_ptr(dev))
            if rc == 0:
                return gx, g_mu_w, g_rho_w, g_mu_b, g_rho_b, None, None, None, None, None, None, None, None
            if rc not in (_lib.E_UNSUPPORTED, _lib.E_ALIGN):
                check(rc, "bnn_linear_backward_narrow_sampled")
            # not applicable here (workspace, alignment): the general kernels below; hand the KL entries back
            for e in (ent_w, ent_b):
                if e is not None:
                    _tls.kl_pending[(e.mu.device.index, e.mu.data_ptr())] = e
            gx = g_mu_w = g_rho_w = g_mu_b = g_rho_b = None
        if ctx.needs_input_grad[0]:
            # a shared input sums its gradient over the samples: fp32 partials, then one reduction
            gx_dtype = torch.float32 if ctx.shared_x else x.dtype
            fused = K % 4 == 0 and N % (8 if _bf(gy) else 4) == 0 and \
                (compute == _lib.COMPUTE_BF16 or (not _bf(gy) and gx_dtype == torch.float32))
            if ctx.drawn_w is not None and _bf(gy) and gx_dtype == torch.bfloat16:
                gx = _dgrad_drawn_raw(gy, ctx.drawn_w, K)
            elif fused:
                gx = torch.empty((S, M, K), dtype=gx_dtype, device=dev)
                flags = gflag | (_lib.FLAG_Y_BF16 if gx_dtype == torch.bfloat16 else 0)
                check(lib.bnn_linear_backward_input_sampled(ptr(gy), M * N, N, ptr(mu_w), ptr(rho_w), ptr(gx),
                                                            M * K, K, M, N, K, S, ctypes.byref(rw), compute, flags,
                                                            stream_ptr(dev)), "bnn_linear_backward_input_sampled")
            else:
                w = _sample_affine_philox_raw(mu_w, rho_w, ctx.key_w)      # (S, N, K) fp32, the forward's draw
                gx = _dgrad_plain_raw(gy, w, gx_dtype)
            if ctx.shared_x:
                gx = _sum_samples(gx).to(x.dtype)
        need_b = rho_b is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            g_mu_w = torch.empty_like(mu_w)
            g_rho_w = torch.empty_like(rho_w)
            flags = (_lib.FLAG_X_BF16 if _bf(x) else 0) | (_lib.FLAG_Y_BF16 if _bf(gy) else 0)
            rb = None
            if need_b:                                                     # bias gradient rides in the same launch
                g_mu_b = torch.empty_like(rho_b)
                g_rho_b = torch.empty_like(rho_b)
                rb = _rng_struct(ctx.key_b, dev)
            # ... and so does the KL gradient, when KLDivergence's backward parked it for this layer
            ent_w = _kl_take(mu_w) if M > 0 else None                     # (an empty batch zero-fills; the flush adds KL)
            ent_b = _kl_take(mu_b) if (ent_w is not None and need_b and mu_b is not None) else None
            kl = _kl_fuse_struct(ent_w, ent_b) if ent_w is not None else None
            check(lib.bnn_linear_backward_weight_sampled(ptr(x), 0 if ctx.shared_x else M * K, K, ptr(gy), M * N, N,
                                                         ptr(rho_w), ptr(g_mu_w), ptr(g_rho_w),
                                                         ptr(rho_b) if need_b else None, ptr(g_mu_b), ptr(g_rho_b),
                                                         M, N, K, S, ctypes.byref(rw),
                                                         ctypes.byref(rb) if rb is not None else None,
                                                         ctypes.byref(kl) if kl is not None else None,
                                                         compute, flags, 0, stream_ptr(dev)),
                  "bnn_linear_backward_weight_sampled")
        elif need_b:
            gb = _colsum_raw(gy)                                           # (S, N)
            g_mu_b, g_rho_b = _sample_affine_bwd_raw(gb, rho_b, rho_b.numel(), S, key=ctx.key_b)
        return gx, g_mu_w, g_rho_w, g_mu_b, g_rho_b, None, None, None, None, None, None, None, None


def linear_sampled(x, mu_w, rho_w, mu_b, rho_b, key_w, key_b, shared_x, compute="f32", relu=False,
                   out_dtype=torch.float32, predrawn=None):
    return _SampledLinear.apply(x, mu_w.contiguous(), rho_w.contiguous(),
                                None if mu_b is None else mu_b.contiguous(),
                                None if rho_b is None else rho_b.contiguous(),
                                key_w, key_b, shared_x, _compute_code(compute), bool(relu), out_dtype, predrawn,
                                torch.is_grad_enabled())


class _PlainLinear(torch.autograd.Function):
    """y[s] = x[s] @ w[s]^T + b[s] with given weights (F.linear, dense.py:60)."""

    @staticmethod
    def forward(ctx, x, w, b, shared_x, compute):
        require_cuda_f32(x, "x")
        require_cuda_f32(w, "w")
        S, N, K = w.shape
        M = x.shape[-2]
        if x.shape[-1] != K:
            raise BnnHipError("linear: input has %d features, weight expects %d" % (x.shape[-1], K))
        y = torch.empty((S, M, N), dtype=torch.float32, device=x.device)
        if b is not None:
            require_cuda_f32(b, "b")
        check(_lib.load().bnn_linear_forward(ptr(x), 0 if shared_x else M * K, K, ptr(w), N * K, ptr(b), N,
                                              ptr(y), M * N, N, M, N, K, S, compute, 0, stream_ptr(x.device)),
              "bnn_linear_forward")
        ctx.save_for_backward(x, w)
        ctx.shared_x, ctx.has_b, ctx.compute = shared_x, b is not None, compute
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        S, N, K = w.shape
        M = x.shape[-2]
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _dgrad_plain_raw(gy, w, torch.float32)
            if ctx.shared_x:
                gx = _sum_samples(gx)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            check(_lib.load().bnn_linear_backward_weight(ptr(x), 0 if ctx.shared_x else M * K, K, ptr(gy), M * N, N,
                                                         ptr(gw), N * K, M, N, K, S, ctx.compute, 0, 0,
                                                         stream_ptr(gy.device)), "bnn_linear_backward_weight")
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = _colsum_raw(gy)
        return gx, gw, gb, None, None


def linear_plain(x, w, b, shared_x, compute="f32"):
    return _PlainLinear.apply(x.contiguous(), w.contiguous(), None if b is None else b.contiguous(),
                              shared_x, _compute_code(compute))


# --------------------------------------------------------------------------- K2 conv2d
def _conv_shape(x_shape, w_shape, stride, padding, dilation, groups):
    sh = Conv2dShape()
    sh.B, sh.C, sh.H, sh.W = x_shape
    sh.O, _, sh.KH, sh.KW = w_shape
    sh.stride_h, sh.stride_w = stride
    sh.pad_h, sh.pad_w = padding
    sh.dil_h, sh.dil_w = dilation
    sh.groups = groups
    OH = (sh.H + 2 * sh.pad_h - sh.dil_h * (sh.KH - 1) - 1) // sh.stride_h + 1
    OW = (sh.W + 2 * sh.pad_w - sh.dil_w * (sh.KW - 1) - 1) // sh.stride_w + 1
    return sh, OH, OW


def _conv_workspace(sh, x_samples, compute, device):
    """im2col panel for the fast conv path (None: the generic kernel runs)."""
    nbytes = _lib.load().bnn_conv2d_workspace_bytes(ctypes.byref(sh), x_samples, compute)
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


_CONV_LDS = {64: (78 * 1024, 4), 128: (136 * 1024, 6)}       # O -> (LDS block, ring stages) of k_conv_bf16


def conv_dense_eligible(sh, OH, OW):
    """Shapes bnn_conv2d_dense_forward is built for (include/bnn_hip.h): both BASELINE conv layers."""
    if sh.groups != 1 or not (sh.C == 64 or sh.C % 128 == 0) or sh.O not in (64, 128):
        return False
    P = OH * OW
    if P > 128:
        return False
    img_bytes = sh.H * sh.W * sh.C * 2
    block, st = _CONV_LDS[sh.O]
    return img_bytes + st * sh.O * 128 <= block and sh.O * P * 4 <= block


_CONV_LDS_X3 = {64: (136 * 1024, 4), 128: (136 * 1024, 3)}   # fp32 parity mode: three image planes, the big block for both widths
CONV_X3_F32 = os.environ.get("BNN_CONV_X3", "1") != "0"      # fp32 parity mode of eligible inference convolutions without the im2col panel


def conv_dense_x3_eligible(sh, OH, OW):
    """Shapes bnn_conv2d_dense_forward_x3 takes: as conv_dense_eligible with three bf16 planes of an image resident."""
    if sh.groups != 1 or not (sh.C == 64 or sh.C % 128 == 0) or sh.O not in (64, 128):
        return False
    P = OH * OW
    if P > 128:
        return False
    block, st = _CONV_LDS_X3[sh.O]
    return 3 * sh.H * sh.W * sh.C * 2 + st * sh.O * 128 <= block and sh.O * P * 4 <= block


class _SampledConv2d(torch.autograd.Function):
    """y[s] = conv2d(x[s], w_s, b_s, ...) as an implicit GEMM with in-kernel draws
    (NormalConv2d.forward, conv.py:112-119)."""

    @staticmethod
    def forward(ctx, x, mu_w, rho_w, mu_b, rho_b, key_w, key_b, shared_x, conv_args, compute, track=True):
        # track: grad mode at the call (inside forward it is always off, and needs_input_grad only mirrors requires_grad)
        require_cuda_f32(x, "x")
        require_cuda_f32(mu_w, "weight.mean")
        require_cuda_f32(rho_w, "weight.scale")
        stride, padding, dilation, groups = conv_args
        S = key_w.nsamples
        xs = x.shape[-4:]
        if xs[1] != mu_w.shape[1] * groups:
            raise BnnHipError("conv2d: input has %d channels, weight expects %d" % (xs[1], mu_w.shape[1] * groups))
        sh, OH, OW = _conv_shape(xs, mu_w.shape, stride, padding, dilation, groups)
        if OH < 1 or OW < 1:
            raise BnnHipError("conv2d: kernel larger than padded input")
        y = torch.empty((S, sh.B, sh.O, OH, OW), dtype=torch.float32, device=x.device)
        per = sh.B * sh.C * sh.H * sh.W
        _lib.ensure_workspace(x.device)
        ctx.save_for_backward(x, mu_w, rho_w, rho_b if mu_b is not None else None)
        ctx.key_w, ctx.key_b, ctx.shared_x, ctx.conv_args, ctx.compute = key_w, key_b, shared_x, conv_args, compute
        if compute == _lib.COMPUTE_BF16 and DRAW_ONCE_BF16 and conv_dense_eligible(sh, OH, OW) and mu_w.data_ptr() % 16 == 0:
            # bf16 mode: weights drawn once (tap-major) for the S samples, then the implicit GEMM on them -- images resident
            # in LDS, im2col in the fragment addresses, no panel (csrc/bnn_dense.hip, k_conv_bf16)
            K = mu_w[0].numel()
            pre = draw_layers([(mu_w.reshape(sh.O, K), rho_w.reshape(sh.O, K), mu_b, rho_b, key_w, key_b, sh.KH * sh.KW)], S)[0]
            kp = pre.w.shape[2]
            check(_lib.load().bnn_conv2d_dense_forward(ptr(x), 0 if shared_x else per, ptr(pre.w), sh.O * kp, kp,
                                                        ptr(pre.b), sh.O if pre.b is not None else 0, ptr(y),
                                                        sh.B * sh.O * OH * OW, ctypes.byref(sh), S, 0, stream_ptr(x.device)),
                  "bnn_conv2d_dense_forward")
            return y
        needs_grad = track and any(ctx.needs_input_grad[:5])
        if (compute == _lib.COMPUTE_F32 and CONV_X3_F32 and not needs_grad and conv_dense_x3_eligible(sh, OH, OW) and
                mu_w.data_ptr() % 16 == 0):
            # fp32 parity mode, inference: the same implicit GEMM on three bf16 planes per operand -- weights drawn once as
            # planes (tap-major), images split into planes as they become resident in LDS, six plane pairs per 64-k block; no
            # im2col panel (the backward still takes the panel kernels, so training-time forwards stay on them)
            K = mu_w[0].numel()
            pre = draw_layers([(mu_w.reshape(sh.O, K), rho_w.reshape(sh.O, K), mu_b, rho_b, key_w, key_b, sh.KH * sh.KW)], S, x3=True)[0]
            kp = pre.w.shape[3]
            check(_lib.load().bnn_conv2d_dense_forward_x3(ptr(x), 0 if shared_x else per, ptr(pre.w), S * sh.O * kp, sh.O * kp, kp,
                                                           ptr(pre.b), sh.O if pre.b is not None else 0, ptr(y),
                                                           sh.B * sh.O * OH * OW, ctypes.byref(sh), S, 0, stream_ptr(x.device)),
                  "bnn_conv2d_dense_forward_x3")
            return y
        rw = _rng_struct(key_w, x.device)
        rb = _rng_struct(key_b, x.device) if mu_b is not None else None
        ws, wsb = _conv_workspace(sh, 1 if shared_x else S, compute, x.device)
        check(_lib.load().bnn_conv2d_forward_sampled(
            ptr(x), 0 if shared_x else per, ptr(mu_w), ptr(rho_w), ptr(mu_b), ptr(rho_b), ptr(y),
            sh.B * sh.O * OH * OW, ctypes.byref(sh), S, ctypes.byref(rw),
            ctypes.byref(rb) if rb is not None else None, compute, 0, ptr(ws), wsb, stream_ptr(x.device)),
            "bnn_conv2d_forward_sampled")
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mu_w, rho_w, rho_b = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        S = ctx.key_w.nsamples
        gy = gy.contiguous()
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_b = rho_b is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        sh, OH, OW = _conv_shape(x.shape[-4:], mu_w.shape, stride, padding, dilation, groups)
        K = mu_w[0].numel()
        if groups == 1 and K % 8 == 0 and sh.O % 8 == 0:
            return _SampledConv2d._backward_panel(ctx, gy, sh, OH, OW, need_x, need_w, need_b)
        gx = g_mu_w = g_rho_w = g_mu_b = g_rho_b = None
        if need_x or need_w:
            w = _sample_affine_philox_raw(mu_w, rho_w, ctx.key_w)          # (S, O, Cg, KH, KW)
            gxs, gws = [], []
            for s in range(S):
                xs_ = x if ctx.shared_x else x[s]
                if need_x:
                    gxs.append(torch.nn.grad.conv2d_input(xs_.shape, w[s], gy[s], stride, padding, dilation, groups))
                if need_w:
                    gws.append(torch.nn.grad.conv2d_weight(xs_, w[s].shape, gy[s], stride, padding, dilation, groups))
            if need_x:
                gx = torch.stack(gxs).sum(0) if ctx.shared_x else torch.stack(gxs)
            if need_w:
                gw = torch.stack(gws)
                g_mu_w, g_rho_w = _sample_affine_bwd_raw(gw, rho_w, rho_w.numel(), S, key=ctx.key_w)
        if need_b:
            gb = gy.sum((1, 3, 4))
            g_mu_b, g_rho_b = _sample_affine_bwd_raw(gb, rho_b, rho_b.numel(), S, key=ctx.key_b)
        return gx, g_mu_w, g_rho_w, g_mu_b, g_rho_b, None, None, None, None, None, None

    @staticmethod
    def _backward_panel(ctx, gy, sh, OH, OW, need_x, need_w, need_b):
        """All-HIP backward through the im2col panel (include/bnn_hip.h, 'backward of K2 conv2d'): the conv is
        the linear layer rows = (image, pixel), so the linear backward kernels (re-drawn weights, fused
        draw-backward) do the work; only the layout changes (NCHW -> rows, col2im) are conv-specific."""
        x, mu_w, rho_w, rho_b = ctx.saved_tensors
        S, compute, dev = ctx.key_w.nsamples, ctx.compute, gy.device
        lib = _lib.load()
        _lib.ensure_workspace(dev)
        st = stream_ptr(dev)
        P, O = OH * OW, sh.O
        M, K = sh.B * P, mu_w[0].numel()
        bf = compute == _lib.COMPUTE_BF16
        adt = torch.bfloat16 if bf else torch.float32
        rows = torch.empty((S, M, O), dtype=adt, device=dev)
        check(lib.bnn_nchw_to_rows(ptr(gy), S * sh.B, O, P, ptr(rows), int(bf), st), "bnn_nchw_to_rows")
        rw = _rng_struct(ctx.key_w, dev)
        gx = g_mu_w = g_rho_w = g_mu_b = g_rho_b = None
        aflag = _lib.FLAG_X_BF16 if bf else 0
        if need_w:
            nsx = 1 if ctx.shared_x else S
            panel = torch.empty((nsx, M, K), dtype=adt, device=dev)
            per = sh.B * sh.C * sh.H * sh.W
            check(lib.bnn_conv2d_im2col(ptr(x), 0 if ctx.shared_x else per, ctypes.byref(sh), nsx, ptr(panel), int(bf), st),
                  "bnn_conv2d_im2col")
            g_mu_w = torch.empty_like(mu_w)
            g_rho_w = torch.empty_like(rho_w)
            rb = None
            if need_b:
                g_mu_b = torch.empty_like(rho_b)
                g_rho_b = torch.empty_like(rho_b)
                rb = _rng_struct(ctx.key_b, dev)
            ent_w = _kl_take(mu_w)
            kl = _kl_fuse_struct(ent_w, None) if ent_w is not None else None    # (a conv bias entry is left to the flush)
            check(lib.bnn_linear_backward_weight_sampled(ptr(panel), 0 if ctx.shared_x else M * K, K, ptr(rows), M * O, O,
                                                         ptr(rho_w), ptr(g_mu_w), ptr(g_rho_w),
                                                         ptr(rho_b) if need_b else None, ptr(g_mu_b), ptr(g_rho_b),
                                                         M, O, K, S, ctypes.byref(rw),
                                                         ctypes.byref(rb) if rb is not None else None,
                                                         ctypes.byref(kl) if kl is not None else None,
                                                         compute, aflag | (_lib.FLAG_Y_BF16 if bf else 0), 0, st),
                  "bnn_linear_backward_weight_sampled")
            need_b = False
        if need_x:
            gpanel = torch.empty((S, M, K), dtype=torch.float32, device=dev)
            check(lib.bnn_linear_backward_input_sampled(ptr(rows), M * O, O, ptr(mu_w), ptr(rho_w), ptr(gpanel), M * K, K,
                                                        M, O, K, S, ctypes.byref(rw), compute, aflag, st),
                  "bnn_linear_backward_input_sampled")
            gx = torch.empty((sh.B, sh.C, sh.H, sh.W) if ctx.shared_x else (S, sh.B, sh.C, sh.H, sh.W),
                             dtype=torch.float32, device=dev)
            check(lib.bnn_conv2d_col2im(ptr(gpanel), ctypes.byref(sh), S, int(ctx.shared_x), ptr(gx), st), "bnn_conv2d_col2im")
        if need_b:
            gb = _colsum_raw(rows)                                          # (S, O)
            g_mu_b, g_rho_b = _sample_affine_bwd_raw(gb, rho_b, rho_b.numel(), S, key=ctx.key_b)
        return gx, g_mu_w, g_rho_w, g_mu_b, g_rho_b, None, None, None, None, None, None


def conv2d_flipout_eligible(x, mean, stride, padding, dilation, groups):
    """bf16 compute, inference: the one-launch Flipout conv (bnn_conv2d_flipout_forward) takes this layer."""
    if not (DRAW_ONCE_BF16 and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and groups == 1):
        return False
    O, C, KH, KW = mean.shape
    if not ((C == 64 or C % 128 == 0) and O in (32, 64) and mean.data_ptr() % 16 == 0):
        return False
    sh, OH, OW = _conv_shape(x.shape, mean.shape, stride, padding, dilation, groups)
    if OH < 1 or OW < 1 or OH * OW > 128:
        return False
    block = _CONV_LDS[2 * O][0]
    return sh.H * sh.W * C * 2 + (C // 8) * 16 + 4 * 2 * O * 128 <= block and O * OH * OW * 4 <= block


def conv2d_flipout(x, mean, scale, R, S, stride, padding, dilation):
    """FlipOutNormalConv2d.forward (conv.py:207-221) in two launches: the mean and the stddev written tap-major as bf16
    (bnn_draw_multi, kinds 1 / 2 -- no eps), then ONE implicit GEMM that shares the A fragment between the two
    contractions, with S flipped into its sign bits and R applied in the epilogue (no autograd: inference path)."""
    x = x.contiguous()
    require_cuda_f32(x, "x")
    O, C, KH, KW = mean.shape
    dev = x.device
    lib = _lib.load()
    w2 = flipout_conv_weights(mean, scale)
    kp = w2.shape[1]
    sh, OH, OW = _conv_shape(x.shape, mean.shape, stride, padding, dilation, 1)
    y = torch.empty((sh.B, O, OH, OW), dtype=torch.float32, device=dev)
    # the sign tensors broadcast over the batch like the reference's expand_as (conv.py:207-221): drawn for another batch size
    # (sample=False after the constructor's sample(1)) a (1, C, 1, 1) tensor serves every image
    Sf = S.to(torch.float32).expand(sh.B, C, 1, 1).reshape(sh.B, C).contiguous()
    Rf = R.to(torch.float32).expand(sh.B, O, 1, 1).reshape(sh.B, O).contiguous()
    check(lib.bnn_conv2d_flipout_forward(ptr(x), ptr(w2), kp, ptr(Sf), ptr(Rf), ptr(y), ctypes.byref(sh), 0, stream_ptr(dev)),
          "bnn_conv2d_flipout_forward")
    return y


def conv2d_flipout_x3_fused_eligible(x, mean, stride, padding, dilation):
    """ONE contraction launch for both of Flipout's convolutions in the fp32 parity mode (bnn_conv2d_flipout_forward_x3): the tile's
    columns are [O means | O stddevs], so 2 O = 64 or 128, and three planes of an image + the ring fit the LDS block."""
    O, C, KH, KW = mean.shape
    if not (CONV_X3_F32 and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and O in (32, 64) and (C == 64 or C % 128 == 0)
            and mean.data_ptr() % 16 == 0):
        return False
    sh, OH, OW = _conv_shape(x.shape, mean.shape, stride, padding, dilation, 1)
    if OH < 1 or OW < 1 or OH * OW > 128:
        return False
    block, st = _CONV_LDS_X3[2 * O]
    return 3 * (sh.H * sh.W * C * 2 + (C // 8) * 16) + st * 2 * O * 128 <= block and O * OH * OW * 4 <= block


def conv2d_flipout_x3_fused(x, mean, stddev, R, S, stride, padding, dilation):
    """FlipOutNormalConv2d.forward (conv.py:207-221) in the fp32 parity mode in TWO launches: mean and stddev as three bf16 planes,
    tap-major, stacked [O means | O stddevs] (bnn_draw_multi, kind 1), then ONE implicit GEMM on three-plane operands that shares the
    A fragment between the two contractions -- S in its sign bits, R in the epilogue (no autograd: inference path)."""
    x = x.contiguous()
    O, C, KH, KW = mean.shape
    K = C * KH * KW
    kp = _pad64(K)
    dev = x.device
    lib = _lib.load()
    w2 = torch.empty((3, 2 * O, kp), dtype=torch.bfloat16, device=dev)
    arr = (_lib.DrawTensor * 2)()
    srcs = (mean.detach().contiguous(), stddev.detach().contiguous())
    for i, src in enumerate(srcs):
        require_cuda_f32(src, "weight")
        t = arr[i]
        t.mu, t.rho, t.rows, t.cols = src.data_ptr(), src.data_ptr(), O, K
        t.out, t.ld, t.out_sample_stride, t.out_dtype = w2.data_ptr() + i * O * kp * 2, kp, 2 * O * kp, _lib.BF16X3
        t.kind, t.taps = 1, KH * KW
    check(lib.bnn_draw_multi(arr, 2, 1, None, 0, None, stream_ptr(dev)), "bnn_draw_multi")
    sh, OH, OW = _conv_shape(x.shape, mean.shape, stride, padding, dilation, 1)
    y = torch.empty((sh.B, O, OH, OW), dtype=torch.float32, device=dev)
    Sf = S.to(torch.float32).expand(sh.B, C, 1, 1).reshape(sh.B, C).contiguous()
    Rf = R.to(torch.float32).expand(sh.B, O, 1, 1).reshape(sh.B, O).contiguous()
    check(lib.bnn_conv2d_flipout_forward_x3(ptr(x), ptr(w2), 2 * O * kp, kp, ptr(Sf), ptr(Rf), ptr(y), ctypes.byref(sh), 0, stream_ptr(dev)),
          "bnn_conv2d_flipout_forward_x3")
    return y


def conv2d_flipout_x3(x, mean, stddev, R, S, stride, padding, dilation):
    """FlipOutNormalConv2d.forward (conv.py:207-221) in the fp32 parity mode without the im2col panel: mean and stddev as three
    bf16 planes each (ONE bnn_draw_multi launch, kind 1), two implicit-GEMM contractions (bnn_conv2d_dense_forward_x3), the sign
    tensors by torch (no autograd: inference path; the caller checked conv2d_plain_x3_eligible).  Shapes the one-launch form does not
    take (O = 128)."""
    x = x.contiguous()
    sh, OH, OW = _conv_shape(x.shape, mean.shape, stride, padding, dilation, 1)
    pm, ps = plain_conv_planes([mean.detach().contiguous(), stddev.detach().contiguous()])
    out = _conv_planes_raw(x, pm, None, sh, OH, OW)[0]
    noise = _conv_planes_raw((x * S.expand_as(x)).contiguous(), ps, None, sh, OH, OW)[0]
    return out + noise * R.expand_as(out)


def flipout_conv_weights(mean, scale):
    """[O rows of the mean | O rows of the stddev] as bf16, tap-major, rows zero-padded to a multiple of 64 columns: the
    weight operand of bnn_conv2d_flipout_forward (one bnn_draw_multi launch with kinds 1 / 2 -- no eps)."""
    mean, scale = mean.detach().contiguous(), scale.detach().contiguous()
    require_cuda_f32(mean, "weight.mean")
    require_cuda_f32(scale, "weight.scale")
    O, C, KH, KW = mean.shape
    K = C * KH * KW
    kp = _pad64(K)
    dev = mean.device
    lib = _lib.load()
    w2 = torch.empty((2 * O, kp), dtype=torch.bfloat16, device=dev)
    arr = (_lib.DrawTensor * 2)()
    for i, kind in enumerate((1, 2)):
        t = arr[i]
        t.mu, t.rho, t.rows, t.cols = mean.data_ptr(), scale.data_ptr(), O, K
        t.out, t.ld, t.out_sample_stride, t.out_dtype = w2.data_ptr() + i * O * kp * 2, kp, O * kp, _lib.BF16
        t.kind, t.taps = kind, KH * KW
    check(lib.bnn_draw_multi(arr, 2, 1, None, 0, None, stream_ptr(dev)), "bnn_draw_multi")
    return w2


def conv2d_sampled(x, mu_w, rho_w, mu_b, rho_b, key_w, key_b, shared_x, stride, padding, dilation, groups,
                   compute="f32"):
    return _SampledConv2d.apply(x.contiguous(), mu_w.contiguous(), rho_w.contiguous(),
                                None if mu_b is None else mu_b.contiguous(),
                                None if rho_b is None else rho_b.contiguous(),
                                key_w, key_b, shared_x,
                                (tuple(stride), tuple(padding), tuple(dilation), int(groups)),
                                _compute_code(compute), torch.is_grad_enabled())


class _PlainConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, shared_x, conv_args, compute):
        require_cuda_f32(x, "x")
        require_cuda_f32(w, "w")
        stride, padding, dilation, groups = conv_args
        S = w.shape[0]
        xs = x.shape[-4:]
        if xs[1] != w.shape[2] * groups:
            raise BnnHipError("conv2d: input has %d channels, weight expects %d" % (xs[1], w.shape[2] * groups))
        sh, OH, OW = _conv_shape(xs, w.shape[1:], stride, padding, dilation, groups)
        if OH < 1 or OW < 1:
            raise BnnHipError("conv2d: kernel larger than padded input")
        y = torch.empty((S, sh.B, sh.O, OH, OW), dtype=torch.float32, device=x.device)
        per = sh.B * sh.C * sh.H * sh.W
        wper = w[0].numel()
        _lib.ensure_workspace(x.device)
        ws, wsb = _conv_workspace(sh, 1 if shared_x else S, compute, x.device)
        check(_lib.load().bnn_conv2d_forward(ptr(x), 0 if shared_x else per, ptr(w), wper, ptr(b), sh.O, ptr(y),
                                              sh.B * sh.O * OH * OW, ctypes.byref(sh), S, compute, 0, ptr(ws), wsb,
                                              stream_ptr(x.device)), "bnn_conv2d_forward")
        ctx.save_for_backward(x, w)
        ctx.shared_x, ctx.conv_args, ctx.has_b = shared_x, conv_args, b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        S = w.shape[0]
        gy = gy.contiguous()
        gx = gw = gb = None
        sh, OH, OW = _conv_shape(x.shape[-4:], w.shape[1:], stride, padding, dilation, groups)
        K = w[0, 0].numel()
        if groups == 1 and K % 8 == 0:
            # all-HIP backward through the im2col panel (exact fp32 here: parity mode / Flipout): the conv is
            # F.linear on rows = (image, pixel), see include/bnn_hip.h 'backward of K2 conv2d'
            lib, dev, st = _lib.load(), gy.device, stream_ptr(gy.device)
            _lib.ensure_workspace(dev)
            P, O = OH * OW, sh.O
            M = sh.B * P
            rows = torch.empty((S, M, O), dtype=torch.float32, device=dev)
            check(lib.bnn_nchw_to_rows(ptr(gy), S * sh.B, O, P, ptr(rows), 0, st), "bnn_nchw_to_rows")
            if ctx.needs_input_grad[1]:
                nsx = 1 if ctx.shared_x else S
                panel = torch.empty((nsx, M, K), dtype=torch.float32, device=dev)
                per = sh.B * sh.C * sh.H * sh.W
                check(lib.bnn_conv2d_im2col(ptr(x), 0 if ctx.shared_x else per, ctypes.byref(sh), nsx, ptr(panel), 0, st),
                      "bnn_conv2d_im2col")
                gw = torch.empty_like(w)
                check(lib.bnn_linear_backward_weight(ptr(panel), 0 if ctx.shared_x else M * K, K, ptr(rows), M * O, O,
                                                     ptr(gw), O * K, M, O, K, S, _lib.COMPUTE_F32, 0, 0, st),
                      "bnn_linear_backward_weight")
            if ctx.needs_input_grad[0]:
                gpanel = _dgrad_plain_raw(rows, w.reshape(S, O, K), torch.float32)          # (S, M, K)
                gx = torch.empty((sh.B, sh.C, sh.H, sh.W) if ctx.shared_x else (S, sh.B, sh.C, sh.H, sh.W),
                                 dtype=torch.float32, device=dev)
                check(lib.bnn_conv2d_col2im(ptr(gpanel), ctypes.byref(sh), S, int(ctx.shared_x), ptr(gx), st), "bnn_conv2d_col2im")
            if ctx.has_b and ctx.needs_input_grad[2]:
                gb = _colsum_raw(rows)
            return gx, gw, gb, None, None, None
        gxs, gws = [], []
        for s in range(S):
            xs_ = x if ctx.shared_x else x[s]
            if ctx.needs_input_grad[0]:
                gxs.append(torch.nn.grad.conv2d_input(xs_.shape, w[s], gy[s], stride, padding, dilation, groups))
            if ctx.needs_input_grad[1]:
                gws.append(torch.nn.grad.conv2d_weight(xs_, w[s].shape, gy[s], stride, padding, dilation, groups))
        if gxs:
            gx = torch.stack(gxs).sum(0) if ctx.shared_x else torch.stack(gxs)
        if gws:
            gw = torch.stack(gws)
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = gy.sum((1, 3, 4))
        return gx, gw, gb, None, None, None


def plain_conv_planes(ws):
    """Explicit conv weights, each (O, C, KH, KW) fp32, as the three-plane tap-major operands of bnn_conv2d_dense_forward_x3:
    (3, 1, O, kp) bf16 each -- ONE bnn_draw_multi launch (kind 1: the tensor as it is, no draw)."""
    dev = ws[0].device
    arr = (_lib.DrawTensor * len(ws))()
    outs = []
    for i, w in enumerate(ws):
        require_cuda_f32(w, "w")
        O, C, KH, KW = w.shape
        K = C * KH * KW
        kp = _pad64(K)
        out = torch.empty((3, 1, O, kp), dtype=torch.bfloat16, device=dev)
        t = arr[i]
        t.mu, t.rho, t.rows, t.cols = w.data_ptr(), w.data_ptr(), O, K
        t.out, t.ld, t.out_sample_stride, t.out_dtype = out.data_ptr(), kp, O * kp, _lib.BF16X3
        t.kind, t.taps = 1, KH * KW
        outs.append(out)
    check(_lib.load().bnn_draw_multi(arr, len(ws), 1, None, 0, None, stream_ptr(dev)), "bnn_draw_multi")
    return outs


def _conv_planes_raw(x, planes, b, sh, OH, OW):
    """conv2d of x (B, C, H, W) fp32 with ONE explicit weight given as planes (plain_conv_planes) in the fp32 parity mode:
    the implicit GEMM on three bf16 planes per operand, no im2col panel -> (1, B, O, OH, OW)."""
    kp = planes.shape[3]
    y = torch.empty((1, sh.B, sh.O, OH, OW), dtype=torch.float32, device=x.device)
    check(_lib.load().bnn_conv2d_dense_forward_x3(ptr(x), 0, ptr(planes), sh.O * kp, sh.O * kp, kp, ptr(b), sh.O if b is not None else 0,
                                                   ptr(y), sh.B * sh.O * OH * OW, ctypes.byref(sh), 1, 0, stream_ptr(x.device)),
          "bnn_conv2d_dense_forward_x3")
    return y


def conv2d_plain_x3_eligible(x, w, stride, padding, dilation, groups, compute):
    """fp32 parity mode, no gradient wanted, one explicit weight (S = 1) of a shape the three-plane implicit GEMM takes."""
    if not (CONV_X3_F32 and _compute_code(compute) == _lib.COMPUTE_F32 and x.is_cuda and x.dim() == 4 and w.dim() == 5 and w.shape[0] == 1):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return False
    if x.dtype != torch.float32 or w.dtype != torch.float32 or x.shape[1] != w.shape[2] * groups:
        return False
    sh, OH, OW = _conv_shape(x.shape, w.shape[1:], stride, padding, dilation, groups)
    return OH >= 1 and OW >= 1 and conv_dense_x3_eligible(sh, OH, OW)


def conv2d_plain(x, w, b, shared_x, stride, padding, dilation, groups, compute="f32"):
    bias_grad = torch.is_grad_enabled() and b is not None and b.requires_grad        # (then autograd must see the call)
    if shared_x and not bias_grad and conv2d_plain_x3_eligible(x, w, stride, padding, dilation, groups, compute):
        # inference in the fp32 parity mode: no panel (FlipOutNormalConv2d's two contractions, MC-dropout-free plain layers)
        x, w0 = x.contiguous(), w[0].detach().contiguous()
        if w0.data_ptr() % 16 == 0 and (b is None or b.is_cuda):
            sh, OH, OW = _conv_shape(x.shape, w0.shape, stride, padding, dilation, groups)
            return _conv_planes_raw(x, plain_conv_planes([w0])[0], None if b is None else b.detach().contiguous().float(), sh, OH, OW)
    return _PlainConv2d.apply(x.contiguous(), w.contiguous(), None if b is None else b.contiguous(), shared_x,
                              (tuple(stride), tuple(padding), tuple(dilation), int(groups)),
                              _compute_code(compute))


# --------------------------------------------------------------------------- K3
_kl_ws = {}


def _kl_workspace(device):
    key = (device.type, device.index)
    ws = _kl_ws.get(key)
    if ws is None:
        nbytes = _lib.load().bnn_kl_workspace_bytes(0)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
        _kl_ws[key] = ws
    return ws


def _kl_descs(mus, rhos, priors):
    T = len(mus)
    arr = (KlTensor * T)()
    for i in range(T):
        arr[i].mu = mus[i].data_ptr()
        arr[i].rho = rhos[i].data_ptr()
        arr[i].n = mus[i].numel()
        arr[i].prior_mu = priors[i][0]
        arr[i].prior_sigma = priors[i][1]
    return arr


class _KLNormal(torch.autograd.Function):
    """Returns (out, scalar): out (T + 1) = per-tensor KL SUMS then the KLDivergence scalar (loss.py:16-38); scalar = the
    0-d view out[T] made inside forward, so that a loss built on it back-propagates straight into this node (selecting
    out[T] outside costs the backward a zero-fill and a copy launch)."""

    @staticmethod
    def forward(ctx, n_batches, priors, out, *params):
        T = len(params) // 2
        mus = [p.detach() for p in params[:T]]
        rhos = [p.detach() for p in params[T:]]
        for m, r in zip(mus, rhos):
            require_cuda_f32(m, "mean")
            require_cuda_f32(r, "scale")
        dev = mus[0].device
        if out is None:
            out = torch.empty(T + 1, dtype=torch.float32, device=dev)
        else:
            require_cuda_f32(out, "out")
            if out.numel() != T + 1:
                raise BnnHipError("kl_normal: out must hold %d floats" % (T + 1))
            ctx.mark_dirty(out)
        arr = _kl_descs(mus, rhos, priors)
        check(_lib.load().bnn_kl_forward(arr, T, float(n_batches), ptr(out), ptr(_kl_workspace(dev)),
                                          stream_ptr(dev)), "bnn_kl_forward")
        ctx.save_for_backward(*params)
        ctx.n_batches, ctx.priors, ctx.T = float(n_batches), priors, T
        ctx.set_materialize_grads(False)
        return out, out[T]

    @staticmethod
    def backward(ctx, g_out, g_scalar):
        # Only the scalar is differentiable here -- as the second output, or as out[T] of the first (the per-tensor sums
        # are reported for diagnostics / the sharded all-reduce and carry no gradient).
        params = ctx.saved_tensors
        T = ctx.T
        mus, rhos = params[:T], params[T:]
        if g_out is None and g_scalar is None:
            return (None, None, None) + (None,) * (2 * T)
        up = g_scalar.reshape(1) if g_scalar is not None else None
        if g_out is not None:
            up2 = g_out[T:T + 1].contiguous()
            up = up2 if up is None else up + up2
        up = up.contiguous()
        if FUSE_KL_GRADIENT and all(m.is_leaf and r.is_leaf for m, r in zip(mus, rhos)) and \
                not any((m.device.index, m.data_ptr()) in _tls.kl_pending for m in mus):   # (two KL terms on one tensor: no parking)
            # park the gradient for the layers' weight-gradient launches (see _ThreadState above)
            for m, r, pr in zip(mus, rhos, ctx.priors):
                _tls.kl_pending[(m.device.index, m.data_ptr())] = _KlPending(up, 1.0 / (m.numel() * T * ctx.n_batches), pr, m, r)
            torch.autograd.Variable._execution_engine.queue_callback(_kl_flush)
            return (None, None, None) + (None,) * (2 * T)
        g_mu = [torch.empty_like(m) for m in mus]
        g_rho = [torch.empty_like(r) for r in rhos]
        arr = _kl_descs(mus, rhos, ctx.priors)
        gm = (ctypes.c_void_p * T)(*[t.data_ptr() for t in g_mu])
        gr = (ctypes.c_void_p * T)(*[t.data_ptr() for t in g_rho])
        check(_lib.load().bnn_kl_backward(arr, T, ctx.n_batches, ptr(up), gm, gr, 0, stream_ptr(up.device)),
              "bnn_kl_backward")
        return (None, None, None) + tuple(g_mu) + tuple(g_rho)


def kl_normal(mus, rhos, priors, n_batches=1.0, out=None):
    """priors: list of (prior_mu, prior_sigma) floats.  -> tensor (T + 1): per-tensor KL sums,
    then the KLDivergence scalar.  `out` (optional, T + 1 floats) receives the result in place."""
    mus = [m.contiguous() for m in mus]
    rhos = [r.contiguous() for r in rhos]
    return _KLNormal.apply(n_batches, tuple(priors), out, *mus, *rhos)[0]


def kl_normal_scalar(mus, rhos, priors, n_batches=1.0):
    """The KLDivergence scalar of kl_normal as a 0-d tensor (differentiable; what KLDivergence.forward returns)."""
    mus = [m.contiguous() for m in mus]
    rhos = [r.contiguous() for r in rhos]
    return _KLNormal.apply(n_batches, tuple(priors), None, *mus, *rhos)[1]


class KlDeferred:
    """A KL whose first pass has been launched (kl_normal_begin); mc_mean(..., kl=this) runs the second pass inside the
    MC reduction's launch and fills `out` (T + 1 floats: per-tensor sums, then the KLDivergence scalar)."""
    __slots__ = ("arr", "T", "n_batches", "out", "ws", "keep", "done", "launched")


def kl_normal_begin(mus, rhos, priors, n_batches=1.0, out=None, carry=False):
    """First half of kl_normal for an inference step that ends in mc_mean (no autograd): launches the partial sums
    only.  The result lands in the returned handle's `.out` when mc_mean(..., kl=handle) has run; values are
    bit-identical to kl_normal's.
    carry=True: nothing is launched here -- the next narrow sampled linear layer (N <= 16, a classifier head, whose
    launch leaves most CUs idle) carries the partial sums in ITS launch (bnn_linear_forward_sampled_kl); if no such
    layer runs before mc_mean(kl=handle), mc_mean launches them itself."""
    T = len(mus)
    mus = [m.detach().contiguous() for m in mus]
    rhos = [r.detach().contiguous() for r in rhos]
    for m, r in zip(mus, rhos):
        require_cuda_f32(m, "mean")
        require_cuda_f32(r, "scale")
    dev = mus[0].device
    if out is None:
        out = torch.empty(T + 1, dtype=torch.float32, device=dev)
    else:
        require_cuda_f32(out, "out")
        if out.numel() != T + 1:
            raise BnnHipError("kl_normal_begin: out must hold %d floats" % (T + 1))
    h = KlDeferred()
    h.arr, h.T, h.n_batches, h.out, h.ws = _kl_descs(mus, rhos, priors), T, float(n_batches), out, _kl_workspace(dev)
    h.keep, h.done, h.launched = (mus, rhos), False, False
    if carry:
        _tls.kl_carry = h
        return h
    check(_lib.load().bnn_kl_forward_partial(h.arr, T, ptr(h.ws), stream_ptr(dev)), "bnn_kl_forward_partial")
    h.launched = True
    return h


# --------------------------------------------------------------------------- MC reduction
def mc_mean(y, out=None, scale=None, advance=None, kl=None):
    """scale * sum over the leading MC axis (default scale 1/S = torch.stack(preds).mean(0),
    examples/MNIST/uncertainty.py:50).  `out` (optional, y[0].numel() floats) is written in place.
    `advance` (optional, a device epoch cell of _rng.EpsGenerator.epoch_dev) is bumped by one in the
    same launch: the fresh-noise step of a captured MC forward without a launch of its own.
    `kl` (optional, a KlDeferred from kl_normal_begin): that KL's second pass runs in this launch too.
    y may be a HeadPartials (a hidden layer fused with its classifier head): the same launch then adds the partial logits over
    (part, sample) -- the default scale stays 1 / S."""
    if isinstance(y, HeadPartials):
        parts, S, M, Nh = y.p.shape
        y = y.p.view(parts * S, M, Nh)              # addend v = part * S + s, M * Nh floats apart
        nadd, out_shape = parts * S, (M, Nh)
    else:
        S = nadd = y.shape[0]
        out_shape = y.shape[1:]
    require_cuda_f32(y, "y")
    n = y[0].numel()
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float32, device=y.device)
    else:
        require_cuda_f32(out, "out")
        if out.numel() != n:
            raise BnnHipError("mc_mean: out must hold %d floats" % n)
    sc = (1.0 / S) if scale is None else float(scale)
    adv = ptr(advance) if advance is not None else None
    if kl is not None:
        # second pass of a KL begun by kl_normal_begin, as one extra workgroup of this launch
        if kl.done:
            raise BnnHipError("mc_mean: this KlDeferred has already been finished")
        if not kl.launched:                         # no narrow layer took it along
            if _tls.kl_carry is kl:
                _tls.kl_carry = None
            check(_lib.load().bnn_kl_forward_partial(kl.arr, kl.T, ptr(kl.ws), stream_ptr(y.device)), "bnn_kl_forward_partial")
            kl.launched = True
        check(_lib.load().bnn_mc_sum_kl(ptr(y), n, nadd, n, sc, ptr(out), 0, adv, 1, kl.arr, kl.T, kl.n_batches,
                                        ptr(kl.out), ptr(kl.ws), stream_ptr(y.device)), "bnn_mc_sum_kl")
        kl.done = True
        return out
    check(_lib.load().bnn_mc_sum(ptr(y), n, nadd, n, sc, ptr(out), 0, adv, 1, stream_ptr(y.device)), "bnn_mc_sum")
    return out


# --------------------------------------------------------------------------- training-loop callers
class _SoftmaxXent(torch.autograd.Function):
    """CrossEntropyLoss()(logits, target), reduction 'mean' (examples/MNIST/train.py:39,59-61): the loss
    and d loss / d logits in one HIP pass."""

    @staticmethod
    def forward(ctx, logits, target):
        require_cuda_f32(logits, "logits")
        if target.dtype != torch.int64 or not target.is_cuda:
            raise BnnHipError("cross_entropy: target must be a CUDA int64 tensor")
        R, C = logits.shape
        if target.numel() != R:
            raise BnnHipError("cross_entropy: %d targets for %d rows" % (target.numel(), R))
        lib = _lib.load()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        g = torch.empty_like(logits) if logits.requires_grad else None
        ws = torch.empty(lib.bnn_xent_workspace_bytes(R) // 8, dtype=torch.float64, device=logits.device)
        check(lib.bnn_softmax_xent(ptr(logits), ptr(target), R, C, ptr(loss), ptr(g), ptr(ws),
                                   stream_ptr(logits.device)), "bnn_softmax_xent")
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, up):
        (g,) = ctx.saved_tensors
        return (g * up if g is not None else None), None


def cross_entropy(logits, target):
    """Mean cross-entropy of (rows, classes) fp32 logits against int64 class indices."""
    return _SoftmaxXent.apply(logits.contiguous(), target.contiguous())


def prune_score(mu, rho):
    """log N(0; mu, sigma(rho)) element-wise (PruneNormal's ranking score, prune/prune.py:11)."""
    require_cuda_f32(mu, "mean")
    require_cuda_f32(rho, "scale")
    mu_c, rho_c = mu.detach().contiguous(), rho.detach().contiguous()
    out = torch.empty_like(mu_c)
    check(_lib.load().bnn_prune_score(ptr(mu_c), ptr(rho_c), ptr(out), mu_c.numel(), stream_ptr(mu_c.device)),
          "bnn_prune_score")
    return out
