"""Multi-GPU execution of the Monte-Carlo axis.

MC samples are i.i.d. given the replicated posterior (pytorch_bayesian/nn/container.py:36-37 has
no cross-iteration dependence), so rank r of G runs samples [r*S/G, (r+1)*S/G) -- the eps stream
is addressed by the GLOBAL sample id, so the union over ranks is the same set of draws for any G.
The KL term is eps-independent: every rank reduces a 1/G slice of every posterior tensor.
ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) carries the packed buffer
    [ KL partial sum per tensor (T)  ||  sum over local samples of the predictions (numel) ].
At the BASELINE shape that is 6 + 5120 floats = 20.5 KB: latency-bound, one call.
"""
import torch
import torch.distributed as dist


def shard_samples(samples, rank, world):
    """Contiguous block of the S global sample ids owned by `rank` -> (first, count)."""
    if samples % world != 0:
        raise ValueError("samples (%d) must be divisible by the world size (%d)" % (samples, world))
    per = samples // world
    return rank * per, per


def shard_range(n, rank, world):
    """1/G slice [lo, hi) of a flat tensor of n elements (16-B aligned cuts keep vector loads)."""
    per = ((n + world - 1) // world + 3) // 4 * 4
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def pack(kl_sums, pred_sum):
    return torch.cat([kl_sums.reshape(-1), pred_sum.reshape(-1)])


def allreduce_packed(packed, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


def unpack(packed, numels, samples, pred_shape, n_batches=1.0):
    """-> (KLDivergence scalar as loss.py:28,38 define it, predictive mean over all S samples)."""
    T = len(numels)
    sums = packed[:T]
    n = torch.as_tensor(numels, dtype=packed.dtype, device=packed.device)
    kl = (sums / n).mean() / n_batches
    pred = (packed[T:] / samples).reshape(pred_shape)
    return kl, pred


def forward_sharded(net, x, samples, kl_tensors, n_batches=1.0, group=None):
    """One sharded stochastic forward.

    net: BayesianNetworkModule; kl_tensors: list of (WeightNormal, (prior_mu, prior_sigma)) in
    traversal order.  Returns (kl, predictive_mean, local_outputs (S/G, B, ...))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    s0, cnt = shard_samples(samples, rank, world)
    ys = net.forward_stacked(x, cnt, sample0=s0)
    numels = [p.mean.numel() for p, _ in kl_tensors]
    if x.is_cuda:
        from . import ops
        mus, rhos, priors = [], [], []
        for p, pr in kl_tensors:
            lo, hi = shard_range(p.mean.numel(), rank, world)
            if hi > lo:
                mus.append(p.mean.reshape(-1)[lo:hi])
                rhos.append(p.scale.reshape(-1)[lo:hi])
                priors.append(pr)
        sums = torch.zeros(len(kl_tensors), device=x.device)
        if mus:
            part = ops.kl_normal(mus, rhos, priors, 1.0)[:len(mus)]
            idx = [i for i, (p, _) in enumerate(kl_tensors) if shard_range(p.mean.numel(), rank, world)[1] >
                   shard_range(p.mean.numel(), rank, world)[0]]
            sums[idx] = part
        pred_sum = ops.mc_mean(ys.contiguous(), scale=1.0)
    else:
        from torch.distributions import Normal
        from torch.distributions.kl import kl_divergence
        sums = []
        for p, pr in kl_tensors:
            lo, hi = shard_range(p.mean.numel(), rank, world)
            q = Normal(p.mean.reshape(-1)[lo:hi], p.stddev.reshape(-1)[lo:hi])
            sums.append(kl_divergence(q, Normal(pr[0], pr[1])).sum() if hi > lo else torch.zeros(()))
        sums = torch.stack(sums)
        pred_sum = ys.sum(0)
    packed = allreduce_packed(pack(sums, pred_sum), group)
    kl, pred = unpack(packed, numels, samples, pred_sum.shape, n_batches)
    return kl, pred, ys
