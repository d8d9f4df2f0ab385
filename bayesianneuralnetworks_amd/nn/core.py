"""Weight posteriors (pytorch_bayesian/nn/core.py).

WeightNormal keeps the reference's surface (Parameters .mean/.scale, stddev, variance,
dist, shape, device, requires_grad, size(), sample(), .sampled).  On a CUDA/HIP device a
draw is *addressed* (a DrawKey of the counter-based eps stream) instead of stored: the
fused GEMM/conv kernels regenerate it in their B-operand loader, and `.sampled`
materialises it on demand with the standalone K1 kernel -- bit-identical, because both
run the same device function on the same key.
"""
import torch
from torch.nn import Module
from torch.nn.parameter import Parameter
from torch.distributions import Normal, MultivariateNormal

from .. import ops
from .._rng import default_generator, new_stream_id, DrawKey


class WeightNormal(Module):
    """Mean-field Gaussian posterior, core.py:7-45."""

    def __init__(self, *channels):
        super().__init__()
        self.mean = Parameter(torch.zeros(*channels))
        self.scale = Parameter(torch.zeros(*channels))
        self._stream = new_stream_id()
        self._key = None
        self._explicit = None
        self.sample()                      # core.py:15: `.sampled` exists from construction

    # ---- reference properties (core.py:17-42)
    @property
    def device(self):
        return self.mean.device

    @property
    def requires_grad(self):
        return self.mean.requires_grad

    @property
    def stddev(self):
        return 1e-10 + torch.nn.functional.softplus(self.scale)     # core.py:27

    @property
    def variance(self):
        return self.stddev.pow(2)

    @property
    def dist(self):
        return Normal(self.mean, self.stddev)

    @property
    def shape(self):
        return self.size()

    def size(self, *dims):
        return self.mean.size(*dims)

    # ---- draws
    def sample(self, nsamples=1, sample0=0, epoch=None, gen=0):
        """core.py:44-45.  CUDA/HIP: record the draw key (nothing is computed until a kernel
        needs the weights).  CPU tensors: the reference's own expression on torch's global
        generator -- host-side semantics for construction and CPU-resident modules."""
        if self.mean.is_cuda:
            if epoch is None:
                epoch = default_generator.next_epoch()
            self._key = DrawKey(default_generator.seed, self._stream, sample0, nsamples, epoch, gen=gen)    # gen: _rng.GEN_*
            self._explicit = None
        else:
            self._key = None
            self._explicit = self.mean + self.stddev * torch.randn_like(self.mean)

    @property
    def draw_key(self):
        return self._key

    @property
    def sampled(self):
        if self._explicit is not None:
            return self._explicit
        if self._key is None:
            self.sample()
            return self.sampled
        # the last MC sample of the recorded draw, like the serial loop would leave behind
        return ops.sample_affine_philox(self.mean, self.scale, self._key.last_sample())[0]

    @sampled.setter
    def sampled(self, value):
        self._explicit = value
        self._key = None

    def sampled_all(self):
        """(S, *shape): every MC sample of the recorded draw."""
        if self._key is None:
            return self.sampled.unsqueeze(0)
        return ops.sample_affine_philox(self.mean, self.scale, self._key)

    def sample_with_eps(self, eps):
        """Parity mode: w = mean + stddev * eps with caller-supplied eps (K1, eps given)."""
        if self.mean.is_cuda:
            self.sampled = ops.sample_affine_eps(self.mean, self.scale, eps)
        else:
            self.sampled = self.mean + self.stddev * eps

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        if self._explicit is not None:
            self._explicit = fn(self._explicit)
        if self.mean.is_cuda and self._explicit is not None and not self._explicit.is_cuda:
            self._explicit = None
        return self


class WeightMultivariateNormal(Module):
    """Per-row full-covariance posterior, core.py:48-92 (PyTorch ops; outside the HIP path).
    Keeps the reference's quirks: uniform noise (core.py:91) and the element-wise sqrt of
    the lower-triangular factor (core.py:69)."""

    def __init__(self, *channels):
        super().__init__()
        self.mean = Parameter(torch.zeros(*channels))
        eye = torch.eye(channels[-1])
        self.scale = Parameter(eye.repeat(*channels[:-1], 1, 1))
        self.sample()

    @property
    def device(self):
        return self.mean.device

    @property
    def requires_grad(self):
        return self.mean.requires_grad

    @property
    def variance(self):
        lower = torch.tril(torch.nn.functional.softplus(self.scale))
        return lower + 1e-10 * torch.eye(self.size(-1), device=self.device)

    @property
    def stddev(self):
        return self.variance.sqrt()

    @property
    def dist(self):
        return MultivariateNormal(self.mean, scale_tril=self.variance)

    @property
    def shape(self):
        return self.size()

    def size(self, *dims):
        return self.mean.size(*dims)

    def sample(self):
        noise = torch.rand_like(self.mean).unsqueeze(-1)
        self.sampled = self.mean + torch.matmul(self.stddev, noise).squeeze(-1)

    def sample_with_noise(self, noise):
        """Parity mode: the reference's expression (core.py:89-92) on caller-supplied noise (its own torch.rand_like draw)."""
        self.sampled = self.mean + torch.matmul(self.stddev, noise.unsqueeze(-1)).squeeze(-1)
