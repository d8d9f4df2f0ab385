"""torch-CPU port of the reference hot path -- TEST / BASELINE INFRASTRUCTURE ONLY.

The identical op sequence the reference executes, on CPU tensors, with no Module
machinery: bench.py times it as `cpu_baseline` (kind "port") on the GPU box's host cores,
and tests/test_oracle_golden.py pins it bit-for-bit to the fixtures the real reference
produced (same torch seed -> same mt19937 draws -> same numbers).

Citations are relative to /root/reference/.
"""
import torch
from torch.distributions import Normal
from torch.distributions.kl import kl_divergence


def stddev(scale):
    # pytorch_bayesian/nn/core.py:25-27
    return 1e-10 + torch.nn.functional.softplus(scale)


def sample(mean, scale):
    # pytorch_bayesian/nn/core.py:44-45
    return mean + stddev(scale) * torch.randn_like(mean)


def normal_linear(x, mu_w, rho_w, mu_b, rho_b):
    # pytorch_bayesian/nn/dense.py:46-60: weight draw, then bias draw, then F.linear
    w = sample(mu_w, rho_w)
    b = sample(mu_b, rho_b) if mu_b is not None else None
    return torch.nn.functional.linear(x, w, b)


def normal_conv2d(x, mu_w, rho_w, mu_b, rho_b, stride, padding, dilation, groups):
    # pytorch_bayesian/nn/conv.py:65-73, 112-119
    w = sample(mu_w, rho_w)
    b = sample(mu_b, rho_b) if mu_b is not None else None
    return torch.nn.functional.conv2d(x, w, b, stride, padding, dilation, groups)


def mlp_forward(x, posteriors, samples):
    """BayesianNetworkModule.forward (container.py:32-37) over a Sequential of
    NormalLinear / ReLU: the serial MC loop."""
    outs = []
    for _ in range(samples):
        h = x
        for j, (mu_w, rho_w, mu_b, rho_b) in enumerate(posteriors):
            h = normal_linear(h, mu_w, rho_w, mu_b, rho_b)
            if j < len(posteriors) - 1:
                h = torch.relu(h)
        outs.append(h)
    return outs[0] if len(outs) == 1 else outs


def kl_divergence_loss(posteriors, prior=None, n_batches=1):
    # pytorch_bayesian/nn/loss.py:16-38
    prior = prior or Normal(0, .1)
    parts = []
    for mu_w, rho_w, mu_b, rho_b in posteriors:
        parts.append(kl_divergence(Normal(mu_w, stddev(rho_w)), prior).mean())
        if mu_b is not None:
            parts.append(kl_divergence(Normal(mu_b, stddev(rho_b)), prior).mean())
    return torch.stack(parts).mean() / n_batches
