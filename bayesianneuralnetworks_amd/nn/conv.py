"""Convolutional Bayesian layers (pytorch_bayesian/nn/conv.py).

NormalConv2d is the hot path: an implicit-GEMM MFMA kernel whose B-operand loader draws the
filter bank (bnn_conv2d_forward_sampled).  NormalConv1d/3d have no configuration in
BASELINE.json: they take their weights from the fused HIP sampler (K1) and contract with the
PyTorch-ROCm convNd op.  FlipOut / MC-dropout variants are PyTorch ops.
"""
import torch
from torch.distributions import Normal

from .. import ops
from . import _settings
from ..utils import _single, _pair, _triple
from .container import BayesianModule
from .core import WeightNormal
from .dense import _NormalSampling, _init_normal_posterior


class BayesianConvNd(BayesianModule):
    """conv.py:9-40."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation,
                 transposed, groups, bias, weight, prior, bias_prior=None):
        super().__init__(in_channels, out_channels, prior, bias_prior)
        # conv.py:15-18
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.groups = groups
        if transposed:
            self.weight = weight(in_channels, out_channels // groups, *kernel_size)
        else:
            self.weight = weight(out_channels, in_channels // groups, *kernel_size)
        if bias:
            self.bias = weight(out_channels)
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        pass


class NormalConvNd(_NormalSampling, BayesianConvNd):
    """conv.py:43-73."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation,
                 transposed, groups, bias, prior):
        super().__init__(in_channels, out_channels, _single(kernel_size), stride, padding, dilation,
                         transposed, groups, bias, WeightNormal, prior)

    def reset_parameters(self):
        _init_normal_posterior(self)
        self.sample()

    def _torch_conv(self, op, x, sample):
        if sample:
            self.sample()
        return op(x, *self.sampled, self.stride, self.padding, self.dilation, self.groups)


class NormalConv1d(NormalConvNd):
    """conv.py:76-96."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, prior=Normal(0, .1)):
        super().__init__(in_channels, out_channels, _single(kernel_size), _single(stride),
                         _single(padding), _single(dilation), False, groups, bias, prior)

    def forward(self, x, sample=True):
        if not x.is_cuda or x.dim() not in (2, 3):
            return self._torch_conv(torch.nn.functional.conv1d, x, sample)
        # device: a 1-d convolution is the 2-d one on images of height 1 -- the same HIP kernels (implicit GEMM where the shape
        # allows it), the same draws (the flat element order of (O, C, k) and (O, C, 1, k) is the same)
        x3 = x.unsqueeze(0) if x.dim() == 2 else x
        y = _device_conv2d(self, x3.unsqueeze(2), lambda t: t.unsqueeze(2), (1,) + tuple(self.stride), (0,) + tuple(self.padding),
                           (1,) + tuple(self.dilation), sample).squeeze(2)
        return y.squeeze(0) if x.dim() == 2 else y


def _device_conv2d(layer, x, view, stride, padding, dilation, sample):
    """NormalConv2d.forward on the device (conv.py:112-119) for a layer whose weight tensors `view` turns into (O, C, KH, KW)."""
    S, _, shared, per = layer._mc_plan(x, sample)
    x5 = x if shared else x.reshape(S, per, *x.shape[1:])
    keys = layer._keys(S)
    mode = layer._compute_mode()
    if keys is not None:
        y = ops.conv2d_sampled(x5, view(layer.weight.mean), view(layer.weight.scale),
                               layer.bias.mean if layer.bias is not None else None,
                               layer.bias.scale if layer.bias is not None else None,
                               keys[0], keys[1], shared, stride, padding, dilation, layer.groups, mode)
    else:
        w, b = layer.sampled
        w = view(w)
        y = ops.conv2d_plain(x5, w.unsqueeze(0).expand(S, *w.shape),
                             None if b is None else b.unsqueeze(0).expand(S, -1), shared,
                             stride, padding, dilation, layer.groups, mode)
    return y.reshape(S * per, *y.shape[2:])


class NormalConv2d(NormalConvNd):
    """conv.py:99-119."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, prior=Normal(0, .1)):
        super().__init__(in_channels, out_channels, _pair(kernel_size), _pair(stride),
                         _pair(padding), _pair(dilation), False, groups, bias, prior)

    def forward(self, x, sample=True):
        if not x.is_cuda:
            # CPU-resident module: the reference's own op sequence (conv.py:112-119)
            return self._torch_conv(torch.nn.functional.conv2d, x, sample)
        if x.dim() == 3:
            return self.forward(x.unsqueeze(0), sample).squeeze(0)
        return _device_conv2d(self, x, lambda t: t, self.stride, self.padding, self.dilation, sample)


class NormalConv3d(NormalConvNd):
    """conv.py:122-142."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, prior=Normal(0, .1)):
        super().__init__(in_channels, out_channels, _triple(kernel_size), _triple(stride),
                         _triple(padding), _triple(dilation), False, groups, bias, prior)

    def forward(self, x, sample=True):
        return self._torch_conv(torch.nn.functional.conv3d, x, sample)


class FlipOutNormalConvNd(NormalConvNd):
    """conv.py:145-161: per-example sign tensors R (B, out, 1..) and S (B, in/groups.., 1..); no bias."""

    _op = None
    _ones = ()

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation,
                 transposed, groups, prior):
        super().__init__(in_channels, out_channels, _single(kernel_size), stride, padding, dilation,
                         transposed, groups, False, prior)

    def sample(self, batch_size=1, additional_dims=()):
        dev = self.weight.device
        self.R = (torch.rand(batch_size, self.weight.size(0), *additional_dims, device=dev) - .5).sign()
        self.S = (torch.rand(batch_size, self.weight.size(1), *additional_dims, device=dev) - .5).sign()

    @property
    def sampled(self):
        return (self.R, self.S)

    def forward(self, x, sample=True):
        # conv.py:182-196 (1d), 213-227 (2d), 244-258 (3d)
        if sample:
            self.sample(x.size(0), self._ones)
        if x.is_cuda and x.dim() == 4 and type(self)._op is torch.nn.functional.conv2d:
            # device, 2-d: both contractions are the HIP implicit GEMM (the sign tensors are per EXAMPLE,
            # conv.py:154-161, so they cannot be folded into the weight)
            comp = _settings.get_compute()
            geo = (self.stride, self.padding, self.dilation, self.groups)
            needs_grad = torch.is_grad_enabled() and (x.requires_grad or self._trainable())
            if comp == "bf16" and not needs_grad and ops.conv2d_flipout_eligible(x, self.weight.mean, *geo):
                # one launch for both contractions: shared A tile, S in the fragment's sign bits, R in the epilogue
                return ops.conv2d_flipout(x, self.weight.mean, self.weight.scale, self.R, self.S, *geo[:3])
            if comp == "f32" and not needs_grad and self.groups == 1 and ops.conv2d_flipout_x3_fused_eligible(x, self.weight.mean, *geo[:3]):
                # fp32 parity mode, inference: ONE contraction launch on three-plane operands (S in the sign bits, R in the epilogue)
                return ops.conv2d_flipout_x3_fused(x, self.weight.mean, self.weight.stddev, self.R, self.S, *geo[:3])
            if (comp == "f32" and not needs_grad and self.weight.mean.data_ptr() % 16 == 0 and
                    ops.conv2d_plain_x3_eligible(x, self.weight.mean.detach().unsqueeze(0), *geo, comp)):
                # fp32 parity mode, inference: both contractions as implicit GEMMs on three-plane operands, no im2col panel
                return ops.conv2d_flipout_x3(x, self.weight.mean, self.weight.stddev, self.R, self.S, *geo[:3])
            out = ops.conv2d_plain(x, self.weight.mean.unsqueeze(0), None, True, *geo, comp)[0]
            noise = ops.conv2d_plain(x * self.S.expand_as(x), self.weight.stddev.unsqueeze(0), None, True, *geo, comp)[0]
            return out + noise * self.R.expand_as(out)
        conv = type(self)._op
        out = conv(x, self.weight.mean, self.bias, self.stride, self.padding, self.dilation, self.groups)
        noise = conv(x * self.S.expand_as(x), self.weight.stddev, self.bias, self.stride, self.padding,
                     self.dilation, self.groups)
        out += noise * self.R.expand_as(out)
        return out


def _flipout(ntuple, op, ones):
    class _FlipOut(FlipOutNormalConvNd):
        _op = staticmethod(op)
        _ones = ones

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                     groups=1, prior=Normal(0, .1)):
            super().__init__(in_channels, out_channels, ntuple(kernel_size), ntuple(stride),
                             ntuple(padding), ntuple(dilation), False, groups, prior)
    return _FlipOut


FlipOutNormalConv1d = _flipout(_single, torch.nn.functional.conv1d, (1,))
FlipOutNormalConv1d.__name__ = FlipOutNormalConv1d.__qualname__ = 'FlipOutNormalConv1d'
FlipOutNormalConv2d = _flipout(_pair, torch.nn.functional.conv2d, (1, 1))
FlipOutNormalConv2d.__name__ = FlipOutNormalConv2d.__qualname__ = 'FlipOutNormalConv2d'
FlipOutNormalConv3d = _flipout(_triple, torch.nn.functional.conv3d, (1, 1, 1))
FlipOutNormalConv3d.__name__ = FlipOutNormalConv3d.__qualname__ = 'FlipOutNormalConv3d'


class MCDropoutConvNd(BayesianModule):
    """conv.py:254-262."""

    def __init__(self, in_channels, out_channels, drop_prob):
        super().__init__(in_channels, out_channels, None)
        self.drop_prob = drop_prob


def _mcdropout(name, conv_cls):
    class _MCDropout(MCDropoutConvNd):
        # conv.py:265-326: a stock ConvNd followed by F.dropout that stays on while `sample`;
        # .weight / .bias alias the inner conv's Parameters.
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                     groups=1, bias=True, drop_prob=0.5):
            super().__init__(in_channels, out_channels, drop_prob)
            self.conv = conv_cls(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                 groups, bias)
            self.weight = self.conv.weight
            self.bias = self.conv.bias

        def forward(self, x, sample=True):
            return torch.nn.functional.dropout(self.conv(x), self.drop_prob, sample, False)
    _MCDropout.__name__ = _MCDropout.__qualname__ = name
    return _MCDropout


MCDropoutConv1d = _mcdropout('MCDropoutConv1d', torch.nn.Conv1d)
MCDropoutConv2d = _mcdropout('MCDropoutConv2d', torch.nn.Conv2d)
MCDropoutConv3d = _mcdropout('MCDropoutConv3d', torch.nn.Conv3d)
