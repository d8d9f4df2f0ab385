"""Seeded synthetic posteriors / inputs shared by make_golden.py (which feeds them
to the real reference) and by the tests (which feed the same numbers to the
oracle and to the HIP path).  Uses only torch's CPU generator, which ships with
PyTorch-ROCm on the GPU box, so large tensors need not be stored as fixtures.

Distributions follow the reference's init (pytorch_bayesian/nn/dense.py:34-42,
conv.py:53-61): mu ~ U(+-1/sqrt(fan_in)), rho ~ N(-2.0, 0.15).
"""
import torch


def posterior(gen, w_shape, bias=True):
    """Draw (mu_w, rho_w, mu_b, rho_b) in that order from `gen`."""
    fan_in = 1
    for d in w_shape[1:]:
        fan_in *= d
    bound = 1.0 / fan_in ** 0.5
    mu_w = (torch.rand(w_shape, generator=gen) * 2 - 1) * bound
    rho_w = torch.randn(w_shape, generator=gen) * 0.15 - 2.0
    mu_b = rho_b = None
    if bias:
        mu_b = (torch.rand(w_shape[0], generator=gen) * 2 - 1) * bound
        rho_b = torch.randn(w_shape[0], generator=gen) * 0.15 - 2.0
    return mu_w, rho_w, mu_b, rho_b


def eps_like_reference(seed, shapes, samples=1):
    """The eps stream the reference consumes after torch.manual_seed(seed):
    per MC sample, per layer, randn(w.shape) then randn(b.shape)
    (core.py:45, dense.py:46-54, container.py:36-37).
    shapes: list of (w_shape, b_shape or None).  Returns [sample][layer] -> (ew, eb)."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    out = []
    for _ in range(samples):
        per = []
        for ws, bs in shapes:
            ew = torch.randn(ws)
            eb = torch.randn(bs) if bs is not None else None
            per.append((ew, eb))
        out.append(per)
    torch.random.set_rng_state(state)
    return out


def mlp_posteriors(dims=(784, 1200, 1200, 10), seed=0):
    gen = torch.Generator().manual_seed(seed)
    return [posterior(gen, (o, i)) for i, o in zip(dims[:-1], dims[1:])]


def mlp_input(batch=512, in_features=784, seed=1):
    return torch.randn(batch, in_features, generator=torch.Generator().manual_seed(seed))
