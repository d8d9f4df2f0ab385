"""Multi-GPU execution of the Monte-Carlo axis.

MC samples are i.i.d. given the replicated posterior (pytorch_bayesian/nn/container.py:36-37 has
no cross-iteration dependence), so rank r of G runs samples [r*S/G, (r+1)*S/G) -- the eps stream
is addressed by the GLOBAL sample id, so the union over ranks is the same set of draws for any G.
The KL term is eps-independent: every rank reduces a 1/G slice of every posterior tensor.
ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) carries the packed buffer
    [ KL partial sum per tensor (T)  ||  sum over local samples of the predictions (numel) ].
At the BASELINE shape that is 6 + 5120 floats = 20.5 KB: latency-bound, one call.

Training adds the gradient exchange: GradAllReducer keeps the gradients of all
posterior parameters in a few flat fp32 buckets (the .grad tensors are views into them, so the HIP
backward kernels' results are accumulated straight into the bucket), and all-reduces a bucket as
soon as the backward pass has produced its last gradient -- the exchange of the later layers runs
under the backward of the earlier ones.  xGMI is point-to-point (a ring is bound by one ~153 GB/s
link), so buckets are large and few: 8 MiB by default = 2 buckets for the 19.2 MB of the MLP.
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


class GradAllReducer:
    """Bucketed all-reduce of parameter gradients, overlapped with the backward pass.

        red = GradAllReducer(model.parameters())
        loss.backward()            # buckets are launched from post-accumulate hooks
        red.finish()               # waits; p.grad now holds the mean over ranks
        optimizer.step()

    Buckets are filled in REVERSE parameter order (the order backward produces gradients in)."""

    def __init__(self, params, bucket_bytes=8 << 20, group=None, average=True):
        self.group = group
        self.average = average
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        self.buckets = []          # [flat, [params], pending, handle]
        cur, size = [], 0
        for p in reversed(params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b[1]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, plist):
        dev = plist[0].device
        n = sum(((p.numel() + 3) // 4) * 4 for p in plist)          # 16-B aligned views
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in plist:
            if p.dtype != torch.float32:
                raise TypeError("GradAllReducer expects fp32 parameters")
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += ((p.numel() + 3) // 4) * 4
        self.buckets.append([flat, plist, len(plist), None])

    def _make_hook(self, bi):
        def hook(_p):
            b = self.buckets[bi]
            b[2] -= 1
            if b[2] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.world > 1:
            if self.average:
                b[0].mul_(1.0 / self.world)
            b[3] = dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        """Zero the buckets in place (keeps the views; use instead of optimizer.zero_grad(set_to_none=True))."""
        for b in self.buckets:
            b[0].zero_()

    def finish(self):
        """Wait for every bucket; launches the ones whose parameters got no gradient this step."""
        for b in self.buckets:
            if b[2] != 0 and b[3] is None:
                self._launch(b)
            if b[3] is not None:
                b[3].wait()
                b[3] = None
            b[2] = len(b[1])

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
