"""Dense Bayesian layers (pytorch_bayesian/nn/dense.py).

NormalLinear is the hot path.  On a CUDA/HIP input its forward is draw once + dense GEMM: the posterior is drawn for
all S MC samples of the call by bnn_draw_multi (ONE launch for every NormalLinear of a network when a
BayesianNetworkModule's draw plan runs first), then contracted by bnn_dense_forward (bf16 mode: bf16 operands, fp32
accumulate) or bnn_dense_forward_x3 (fp32 parity mode at inference: three bf16 planes per operand).  Training-time fp32
forwards, narrow fp32 layers and A/B runs take the round-1 fused kernel (bnn_linear_forward_sampled: the draw inside the
B-operand loader of the GEMM).  The other classes (Flipout, multivariate, evidential, MC-dropout)
are outside the HIP scope and run as PyTorch-ROCm ops with the
reference's semantics so that its examples keep working.
"""
import math

import torch
from torch.nn import init
from torch.distributions import Normal
from torch.distributions.multivariate_normal import MultivariateNormal

from .. import _mc, ops
from .._rng import default_generator, generator_for
from . import _settings
from .container import BayesianModule
from .core import WeightNormal, WeightMultivariateNormal


class BayesianLinear(BayesianModule):
    """dense.py:9-24: allocates weight (out, in) and bias (out) posteriors of type `weight`."""

    def __init__(self, in_features, out_features, bias, weight, prior, bias_prior=None):
        super().__init__(in_features, out_features, prior, bias_prior)
        self.weight = weight(out_features, in_features)
        if bias:
            self.bias = weight(out_features)
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        pass


def _init_normal_posterior(layer):
    """dense.py:34-42 / conv.py:53-61: kaiming-uniform means, rho ~ N(-2, 0.15)."""
    init.kaiming_uniform_(layer.weight.mean, a=math.sqrt(5))
    init.normal_(layer.weight.scale, -2.0, 0.15)
    if layer.bias is not None:
        fan_in, _ = init._calculate_fan_in_and_fan_out(layer.weight.mean)
        bound = 1 / math.sqrt(fan_in)
        init.uniform_(layer.bias.mean, -bound, bound)
        init.normal_(layer.bias.scale, -2.0, 0.15)


class _NormalSampling:
    """sample()/.sampled shared by NormalLinear and NormalConvNd (dense.py:46-54, conv.py:65-73)."""

    compute = None      # None -> module-wide default (nn.set_compute)
    activation = None   # 'relu': max(., 0) fused into the kernel epilogue (nn.fuse_activations)
    out_dtype = None    # torch.bfloat16: emit a bf16 hidden activation (bf16 compute mode only)

    def sample(self, nsamples=1, sample0=0):
        # weight first, then bias (dense.py:47-51); one epoch for the layer's draw
        epoch = default_generator.next_epoch() if self.weight.mean.is_cuda else None
        gen = generator_for(self._compute_mode())       # the eps stream is part of the key: whoever re-creates the draw reads it there
        self.weight.sample(nsamples, sample0, epoch, gen)
        if self.bias is not None:
            self.bias.sample(nsamples, sample0, epoch, gen)

    def _fresh_keys(self, nsamples, sample0):
        """The DrawKeys sample(nsamples, sample0) would record, without recording them (the network's draw plan draws on
        them ahead of the layer's call; _adopt_keys records them when the layer uses that draw)."""
        from .._rng import DrawKey
        epoch = default_generator.next_epoch()
        gen = generator_for(self._compute_mode())
        kw = DrawKey(default_generator.seed, self.weight._stream, sample0, nsamples, epoch, gen=gen)
        kb = DrawKey(default_generator.seed, self.bias._stream, sample0, nsamples, epoch, gen=gen) if self.bias is not None else None
        return kw, kb

    def _adopt_keys(self, kw, kb):
        self.weight._key, self.weight._explicit = kw, None
        if self.bias is not None:
            self.bias._key, self.bias._explicit = kb, None

    @property
    def sampled(self):
        return (self.weight.sampled, self.bias.sampled if self.bias is not None else None)

    @sampled.setter
    def sampled(self, value):
        w, b = value
        self.weight.sampled = w
        if self.bias is not None and b is not None:
            self.bias.sampled = b

    def _compute_mode(self):
        return self.compute or _settings.get_compute()

    def _trainable(self):
        """Does any posterior tensor of the layer want a gradient?  (The inference-only paths have no autograd node: a frozen mean
        with a trainable scale or bias must not take them.)"""
        ps = [self.weight.mean, self.weight.scale]
        if self.bias is not None:
            ps += [self.bias.mean, self.bias.scale]
        return any(p.requires_grad for p in ps)

    def _mc_plan(self, x, sample):
        """-> (S, sample0, shared_x, rows_per_sample) for the current MC context."""
        ctx = _mc.current()
        if ctx is None or ctx.samples == 1:
            S, s0 = 1, (ctx.sample0 if ctx else 0)
            shared, per = True, x.shape[0]
        else:
            S, s0 = ctx.samples, ctx.sample0
            if x.shape[0] == ctx.base_batch:
                shared, per = True, x.shape[0]
            elif x.shape[0] == ctx.base_batch * S:
                shared, per = False, ctx.base_batch
            else:
                raise RuntimeError("mc_batched: layer input has %d rows, expected %d or %d"
                                   % (x.shape[0], ctx.base_batch, ctx.base_batch * S))
        if sample:
            self.sample(S, s0)
        return S, s0, shared, per

    def _x3_call(self, x, rows):
        """fp32 parity mode on the dense kernel?  Inference only (the backward kernels take the fused path's saved tensors),
        a 2-D fp32 input or an X3Activation, a wide layer."""
        if not ops.DENSE_X3_F32:
            return False
        if torch.is_grad_enabled() and (self._trainable() or (torch.is_tensor(x) and x.requires_grad)):
            return False
        if torch.is_tensor(x) and (x.dim() != 2 or x.dtype != torch.float32):
            return False
        x2 = x if isinstance(x, ops.X3Activation) else x
        return ops.x3_eligible(x2, self.weight.mean, rows)

    def _try_fused_head(self, x2, shared, per, S, mode, predrawn, sample_head):
        """This hidden layer + the classifier head behind it in ONE launch (bnn_dense_forward_head)?  Only when the caller has
        said it will reduce the outputs itself (predictive_mean -> McContext.lazy_head), in the bf16 mode, at inference, with
        both layers' weights drawn by the network's plan for THIS forward -- otherwise None (the plain path runs)."""
        ctx = _mc.current()
        head = self.__dict__.get("_fuse_head")
        if head is None or ctx is None or not ctx.lazy_head or mode != "bf16" or predrawn is None or torch.is_grad_enabled():
            return None
        hd = getattr(head, "_predrawn", None)
        if hd is None or hd[0] is not ctx or self.out_dtype != torch.bfloat16 or head._compute_mode() != "bf16" or head.activation is not None:
            return None
        K = x2.shape[-1]
        if not ops.dense_head_eligible(per, self.weight.mean.shape[0], predrawn, hd[1]) or predrawn.w.dim() != 3:
            return None
        if x2.dtype == torch.bfloat16:
            pitch = ops.rows_pitch(x2, K)
            xb, ldx, xs = (x2, pitch[0], pitch[1]) if pitch is not None else (x2.contiguous(), K, per * K)
        else:
            xb, ldx, xs = x2.contiguous().to(torch.bfloat16), K, per * K
        head._predrawn = None                                   # the head's drawn weights are consumed here ...
        head._adopt_keys(hd[1].key_w, hd[1].key_b)              # ... and this is its sample() for this forward
        predrawn.wait()
        hd[1].wait()
        hp = ops._dense_head_raw(xb, 0 if shared else xs, per, predrawn, K, self.activation == 'relu', hd[1], ldx=ldx)
        hp.head = head
        return hp

    def _keys(self, S):
        kw = self.weight.draw_key
        kb = self.bias.draw_key if self.bias is not None else None
        if kw is None or (self.bias is not None and kb is None):
            return None
        if kw.nsamples != S:
            if S == 1:
                return kw.last_sample(), (kb.last_sample() if kb is not None else None)
            raise RuntimeError("sample=False: the recorded draw has %d MC samples, this call needs %d"
                               % (kw.nsamples, S))
        return kw, kb


class NormalLinear(_NormalSampling, BayesianLinear):
    """dense.py:27-60."""

    def __init__(self, in_features, out_features, bias=True, prior=Normal(0, .1)):
        super().__init__(in_features, out_features, bias, WeightNormal, prior)

    def reset_parameters(self):
        _init_normal_posterior(self)
        self.sample()                                                   # dense.py:44

    def forward(self, x, sample=True):
        if not x.is_cuda:
            # CPU-resident module: the reference's own op sequence (dense.py:56-60)
            if sample:
                self.sample()
            y = torch.nn.functional.linear(x, *self.sampled)
            return torch.relu(y) if self.activation == 'relu' else y
        if isinstance(x, ops.HeadPartials):
            # the hidden layer in front of this head contracted with this layer's drawn weights in its own launch
            # (ops._dense_head_raw): nothing left to do here but pass the partial logits on
            if x.head is not self:
                raise RuntimeError("a fused hidden layer's partial logits reached a layer that is not its head")
            return x
        if x.dim() == 1:
            return self.forward(x.unsqueeze(0), sample).squeeze(0)
        # weights already drawn for this forward by the network's draw plan (container._draw_plan)?  Consumed on use.
        predrawn = None
        pd = getattr(self, "_predrawn", None)
        if pd is not None:
            self._predrawn = None
            if sample and pd[0] is _mc.current():
                predrawn, sample = pd[1], False
                self._adopt_keys(predrawn.key_w, predrawn.key_b)        # this call's sample(): the plan drew on these keys
        S, _, shared, per = self._mc_plan(x, sample)
        keys = self._keys(S)
        mode = self._compute_mode()
        if keys is not None and mode != "bf16" and self._x3_call(x, per):
            # fp32 parity mode, inference: the draw-once dense path on three bf16 planes per operand (ops.linear_sampled_x3);
            # a hidden layer that feeds another dense layer hands its result on in that format (out_x3, nn.fuse_activations)
            K = x.shape[-1]
            ctx = _mc.current()
            xpl = ctx.x_planes if ctx is not None else None
            if xpl is not None and torch.is_tensor(x) and shared and x.data_ptr() == xpl[0].data_ptr() and x.shape == xpl[0].shape:
                x = ops.X3Activation(xpl[1], K)         # the input's planes were split by the draw plan's launch
            x2 = x if isinstance(x, ops.X3Activation) else (x.reshape(-1, K) if shared else x.reshape(S, -1, K))
            pre3 = predrawn if (predrawn is not None and predrawn.w.dim() == 4) else None
            # the classifier head behind this layer in the same launch (predictive_mean, nn.fuse_activations(fuse_head=True))?
            head_pre, head = None, self.__dict__.get("_fuse_head")
            if head is not None and ctx is not None and ctx.lazy_head and pre3 is not None and head.activation is None:
                hd = getattr(head, "_predrawn", None)
                if hd is not None and hd[0] is ctx and head._compute_mode() != "bf16" and \
                        ops.dense_head_x3_eligible(per, self.weight.mean.shape[0], pre3, hd[1]):
                    head._predrawn = None
                    head._adopt_keys(hd[1].key_w, hd[1].key_b)
                    head_pre = hd[1]
            y = ops.linear_sampled_x3(x2, shared, per, self.weight.mean.detach(), self.weight.scale.detach(),
                                      self.bias.mean.detach() if self.bias is not None else None,
                                      self.bias.scale.detach() if self.bias is not None else None,
                                      keys[0], keys[1], relu=self.activation == 'relu',
                                      planes_out=bool(getattr(self, "out_x3", False)),
                                      predrawn=pre3, head_pre=head_pre)
            if isinstance(y, ops.HeadPartials):
                y.head = head
                return y
            if isinstance(y, ops.X3Activation):
                return y
            return y.reshape(S * per, y.shape[-1])
        if isinstance(x, ops.X3Activation):
            x = x.float()
        lead = x.shape[1:-1]
        K = x.shape[-1]
        x2 = x.reshape(-1, K) if shared else x.reshape(S, -1, K)
        if keys is not None:
            odt = torch.bfloat16 if (self.out_dtype == torch.bfloat16 and mode == "bf16") else torch.float32
            hp = self._try_fused_head(x2, shared, per, S, mode, predrawn, sample_head=True)
            if hp is not None:
                return hp
            y = ops.linear_sampled(x2, self.weight.mean, self.weight.scale,
                                   self.bias.mean if self.bias is not None else None,
                                   self.bias.scale if self.bias is not None else None,
                                   keys[0], keys[1], shared, mode, relu=self.activation == 'relu',
                                   out_dtype=odt, predrawn=predrawn if (mode == "bf16" and predrawn is not None and predrawn.w.dim() == 3) else None)
        else:
            # weights were set explicitly (parity mode / user-assigned .sampled)
            w, b = self.sampled
            y = ops.linear_plain(x2.float(), w.unsqueeze(0).expand(S, -1, -1), None if b is None else
                                 b.unsqueeze(0).expand(S, -1), shared, mode)
            if self.activation == 'relu':
                y = torch.relu(y)
        return y.reshape(S * per, *lead, y.shape[-1])


class FlipoutNormalLinear(NormalLinear):
    """dense.py:63-83: y = x mu^T + ((x * S) sigma^T) * R with random sign vectors; no bias."""

    def __init__(self, in_features, out_features, prior=Normal(0, .1)):
        super().__init__(in_features, out_features, False, prior)

    def sample(self, *unused):
        dev = self.weight.device
        self.R = (torch.rand(self.weight.size(0), device=dev) - .5).sign()
        self.S = (torch.rand(self.weight.size(1), device=dev) - .5).sign()

    @property
    def sampled(self):
        return (self.R, self.S)

    def forward(self, x, sample=True):
        if sample:
            self.sample()
        if not x.is_cuda:
            perturbation = torch.matmul(x * self.S, self.weight.stddev.t()) * self.R
            # the reference hands the perturbation to F.linear as its bias argument (dense.py:83)
            return torch.nn.functional.linear(x, self.weight.mean, perturbation)
        # device: x mu^T + ((x * S) sigma^T) * R == x (mu + sigma * (R (x) S))^T -- the sampled-weight
        # affine with eps = outer(R, S) (HIP K1, eps supplied) and ONE HIP contraction instead of two;
        # both have HIP backwards, so d/d mean, d/d scale, d/dx come out of the same kernels.
        eps = torch.outer(self.R, self.S)
        w = ops.sample_affine_eps(self.weight.mean, self.weight.scale, eps)
        K = x.shape[-1]
        y = ops.linear_plain(x.reshape(-1, K).float(), w.unsqueeze(0), None, True, self._compute_mode())
        return y.reshape(*x.shape[:-1], y.shape[-1])


class MultivariateNormalLinear(BayesianLinear):
    """dense.py:86-138 (PyTorch ops)."""

    def __init__(self, in_features, out_features, bias=True, weight_prior=None, bias_prior=None):
        if not weight_prior:
            weight_prior = MultivariateNormal(torch.zeros(out_features, in_features),
                                              torch.eye(in_features).repeat(out_features, 1, 1))
        if bias and not bias_prior:
            bias_prior = MultivariateNormal(torch.zeros(out_features), torch.eye(out_features))
        super().__init__(in_features, out_features, bias, WeightMultivariateNormal,
                         weight_prior, bias_prior)

    @staticmethod
    def _mask_upper(scale):
        with torch.no_grad():
            upper = torch.triu(torch.ones_like(scale), 1).to(torch.bool)
            scale[upper] = -100

    def reset_parameters(self):
        _init_normal_posterior(self)
        self._mask_upper(self.weight.scale)
        if self.bias is not None:
            self._mask_upper(self.bias.scale)
        self.sample()

    def sample(self):
        self.weight.sample()
        if self.bias is not None:
            self.bias.sample()
            self.sampled = (self.weight.sampled, self.bias.sampled)
        else:
            self.sampled = (self.weight.sampled, None)

    def forward(self, x, sample=True):
        if sample:
            self.sample()
        return torch.nn.functional.linear(x, *self.sampled)


class NormalInverseGaussianLinear(BayesianModule):
    """dense.py:141-162: evidential head, deterministic Linear(in, 4*out) + softplus splits."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, None)
        self.linear = torch.nn.Linear(in_features, 4 * out_features, bias)

    def forward(self, x, sample=False):
        sp = torch.nn.functional.softplus
        gamma, upsilon, alpha, beta = torch.split(self.linear(x), self.out_channels, dim=-1)
        upsilon = 1e-10 + sp(upsilon)
        alpha = 1 + 1e-10 + sp(alpha)
        beta = 1e-10 + sp(beta)
        if not sample:
            return (gamma, upsilon, alpha, beta)
        return Normal(gamma.clone(), torch.sqrt(beta / (upsilon * (alpha - 1))))


class MCDropoutLinear(BayesianModule):
    """dense.py:165-179: dropout stays active while `sample` is true."""

    def __init__(self, in_features, out_features, bias=True, drop_prob=0.5):
        super().__init__(in_features, out_features, None)
        self.drop_prob = drop_prob
        self.linear = torch.nn.Linear(in_features, out_features, bias)

    def forward(self, x, sample=True):
        return torch.nn.functional.dropout(self.linear(x), self.drop_prob, sample, False)
