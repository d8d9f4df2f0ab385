"""Losses (pytorch_bayesian/nn/loss.py).

KLDivergence is the hot path (K3): every mean-field Gaussian tensor of the model that lives
on the GPU and has a scalar Normal prior goes through ONE multi-tensor HIP launch
(bnn_kl_forward); other posteriors (multivariate, tensor-valued priors, CPU tensors) use
torch.distributions and are averaged in.
"""
import math
import warnings

import torch
from torch.nn import Module
from torch.distributions import MultivariateNormal, Normal
from torch.distributions.kl import kl_divergence

from .. import ops
from ..utils import apply_wb
from .core import WeightNormal
from .dense import MultivariateNormalLinear


def _scalar_normal(prior):
    if not isinstance(prior, Normal):
        return None
    if prior.loc.numel() != 1 or prior.scale.numel() != 1:
        return None
    return float(prior.loc), float(prior.scale)


class KLDivergence(Module):
    """loss.py:11-38: mean over all weight/bias tensors of mean(KL(q || prior)), / n_batches."""

    def __init__(self, number_of_batches=1):
        super().__init__()
        self.n_batches = number_of_batches

    @staticmethod
    def _prior_of(module, type):
        prior = module.weight_prior if type == 'w' else module.bias_prior
        return prior

    def compute_kl(self, param, module, type):
        """loss.py:16-28 for one tensor -> its mean KL (0-d tensor)."""
        prior = self._prior_of(module, type)
        sn = _scalar_normal(prior)
        if isinstance(param, WeightNormal) and param.mean.is_cuda and sn is not None:
            out = ops.kl_normal([param.mean], [param.scale], [sn], 1.0)
            return out[1]
        if isinstance(module, MultivariateNormalLinear):
            prior = MultivariateNormal(prior.mean.to(param.device),
                                       scale_tril=prior.scale_tril.to(param.device))
        return kl_divergence(param.dist, prior).mean()

    def forward(self, model):
        entries = model.traverse(
            lambda m: apply_wb(m, lambda p, module, type: (p, module, type),
                               pass_module=True, pass_type=True))
        if entries is None:
            # loss.py:34-36
            raise ValueError('KLDivergence was not able to find BayasianModules')
        fused, other = [], []
        for param, module, type in entries:
            sn = _scalar_normal(self._prior_of(module, type))
            if isinstance(param, WeightNormal) and param.mean.is_cuda and sn is not None:
                fused.append((param, sn))
            else:
                other.append(self.compute_kl(param, module, type))
        if fused and not other:
            return ops.kl_normal_scalar([p.mean for p, _ in fused], [p.scale for p, _ in fused],
                                        [sn for _, sn in fused], self.n_batches)
        means = list(other)
        if fused:
            # mixed model: per-tensor means from the fused launch's scalar with n_batches = 1
            for p, sn in fused:
                means.append(ops.kl_normal([p.mean], [p.scale], [sn], 1.0)[1])
        return torch.stack(means).mean() / self.n_batches


class Entropy(Module):
    """loss.py:41-51."""

    def __init__(self, dim=0):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        if (x == 0).all():
            warnings.warn('Entropy received a tensor containing all zeros', RuntimeWarning)
        return (-x * torch.log(x + 1e-10)).sum(dim=self.dim).mean()


class NormalInverseGaussianLoss(Module):
    """loss.py:54-69: evidential-regression NLL + reg_lambda * |y - gamma| (2 upsilon + alpha)."""

    def __init__(self, reg_lambda=1e-2):
        super().__init__()
        self.reg_lambda = reg_lambda

    def nll(self, y, gamma, upsilon, alpha, beta):
        omega = 2 * beta * (1 + upsilon)
        return (0.5 * torch.log(math.pi / upsilon)
                - alpha * torch.log(omega)
                + (alpha + 0.5) * torch.log(upsilon * (y - gamma) ** 2 + omega)
                + torch.lgamma(alpha) - torch.lgamma(alpha + 0.5))

    def forward(self, gamma, upsilon, alpha, beta, y):
        penalty = torch.mean(torch.abs(y - gamma) * (2 * upsilon + alpha))
        return self.nll(y, gamma, upsilon, alpha, beta).mean() + self.reg_lambda * penalty


class NormalInverseGaussianUncertainty(Module):
    """loss.py:72-79."""

    def forward(self, upsilon, alpha, beta):
        aleatoric = beta / (alpha - 1)
        return (aleatoric, aleatoric / upsilon)
