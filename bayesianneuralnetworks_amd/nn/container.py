"""BayesianModule / BayesianNetworkModule with the reference's surface
(pytorch_bayesian/nn/container.py:6-37) plus the MI355X execution options:

  * mc_batched  -- run all `samples` MC draws of every Bayesian layer in one grid per layer
                   instead of the serial Python loop of container.py:36-37;
  * forward_sharded -- shard the MC axis over the ranks of a torch.distributed (RCCL)
                   process group and all-reduce [KL sums || sum of predictions] once.
"""
import torch
from torch.nn import Module

from .. import _mc
from ..utils import _item_or_list, traverse


class BayesianModule(Module):
    """container.py:6-14: holds the priors and channel counts."""

    def __init__(self, in_channels, out_channels, prior, bias_prior=None):
        super().__init__()
        self.weight_prior = prior
        self.bias_prior = bias_prior if bias_prior else prior   # container.py:12
        self.in_channels = in_channels
        self.out_channels = out_channels

    def kl_divergence(self, number_of_batches=1):
        """Convenience BASELINE.json's north_star names (the reference has only the module form, loss.py:11-38):
        `KLDivergence(number_of_batches)` over this layer's own weight / bias posteriors."""
        from .loss import KLDivergence
        return KLDivergence(number_of_batches)(_Single(self))


class _Single:
    """A one-layer 'model' for KLDivergence.forward: traverse(fn) visits just that layer."""

    def __init__(self, layer):
        self.layer = layer

    def traverse(self, fn, *args, **kwargs):
        return traverse(self.layer, fn, *args, **kwargs)


class BayesianNetworkModule(Module):
    """container.py:17-37.  Subclasses implement `_forward`."""

    def __init__(self, in_channels, out_channels, samples=10):
        super().__init__()
        self.samples = samples
        self.in_channels = in_channels
        self.out_channels = out_channels
        # Opt-in: valid when every layer after the first Bayesian one treats batch rows
        # independently (no train-mode BatchNorm downstream of a Bayesian layer).
        self.mc_batched = False

    def _forward(self, x, *args, **kwargs):
        raise NotImplementedError('self._forward() not implemented')

    def traverse(self, fn, *args, **kwargs):
        return traverse(self, fn, *args, **kwargs)

    def kl_divergence(self, number_of_batches=1):
        """`KLDivergence(number_of_batches)(self)` (loss.py:30-38) -- the `.kl_divergence()` BASELINE.json's north_star names."""
        from .loss import KLDivergence
        return KLDivergence(number_of_batches)(self)

    def forward(self, x, samples=None, *args, **kwargs):
        if samples is None:
            samples = self.samples
        if self.mc_batched and samples > 1 and isinstance(x, torch.Tensor) and x.is_cuda:
            return _item_or_list(self._forward_batched(x, samples, 0, *args, **kwargs))
        # container.py:36-37: the serial MC loop
        return _item_or_list([self._forward(x, *args, **kwargs) for _ in range(samples)])

    # ------------------------------------------------------------------ MI355X paths
    def _forward_batched(self, x, samples, sample0, *args, **kwargs):
        """One pass, all samples per layer launch.  Returns the list of per-sample outputs
        (views of one (S*B, ...) tensor)."""
        return list(self._forward_batched_stacked(x, samples, sample0, *args, **kwargs).unbind(0))

    def predictive_mean(self, x, samples=None, sample0=0, out=None, scale=None, advance=None, kl=None, *args, **kwargs):
        """scale (default 1 / samples) * sum over the MC samples of `_forward(x)` -- torch.stack(preds).mean(0) of
        examples/MNIST/uncertainty.py:50 -- as the tail of ONE batched pass: the caller wants the mean, not the samples, so a
        hidden layer may run fused with the classifier head behind it (nn.fuse_activations; bnn_dense_forward_head) and the
        reduction adds that pair's partial logits in the same launch.  out / advance / kl: as ops.mc_mean."""
        from .. import ops
        if samples is None:
            samples = self.samples
        if not (self.mc_batched and isinstance(x, torch.Tensor) and x.is_cuda):
            ys = self.forward_stacked(x, samples, sample0, *args, **kwargs)
            m = ys.sum(0) * ((1.0 / samples) if scale is None else scale)
            if out is not None:
                out.copy_(m.reshape(out.shape))
                return out
            return m
        y = self._forward_batched_stacked(x, samples, sample0, *args, _lazy_head=True, **kwargs)
        return ops.mc_mean(y, out=out, scale=(1.0 / samples) if scale is None else scale, advance=advance, kl=kl)

    def _forward_batched_stacked(self, x, samples, sample0, *args, _lazy_head=False, **kwargs):
        B = x.shape[0]
        with _mc.McContext(samples, B, sample0) as ctx:
            ctx.lazy_head = _lazy_head
            drawn = self._draw_plan(ctx, x)
            try:
                y = self._forward(x, *args, **kwargs)
            finally:
                for m in drawn:
                    left = getattr(m, "_predrawn", None)
                    if left is not None:
                        left[1].wait()          # a layer `_forward` never reached: join its side-stream draw anyway
                    m._predrawn = None
        from .. import ops
        if isinstance(y, ops.HeadPartials):
            return y if _lazy_head else y.logits()
        if y.shape[0] == B * samples:
            return y.view(samples, B, *y.shape[1:])
        if y.shape[0] == B:
            # no Bayesian layer saw the batch: every draw is the same deterministic output
            return y.unsqueeze(0).expand(samples, *y.shape)
        raise RuntimeError("mc_batched: _forward returned %d rows for batch %d x %d samples"
                           % (y.shape[0], B, samples))

    def _draw_plan(self, ctx, x=None):
        """bf16 compute mode: the posteriors of EVERY NormalLinear of the network are drawn for this forward's S samples
        in ONE launch (ops.draw_layers -> bnn_draw_multi) before `_forward` runs; each layer then finds its drawn
        weights (consumed on first use: a layer called twice draws again, like the reference's per-call sample(),
        dense.py:56-58).  The plan itself leaves the layers' state alone: the keys travel in the Predrawn entry and a layer
        records them -- exactly as its own sample() would -- when it uses the entry.  Returns the layers it drew for."""
        from .. import ops
        from . import _settings
        from .dense import NormalLinear
        if not ops.DRAW_ONCE_BF16:
            return []
        todo, todo3 = [], []
        infer = not torch.is_grad_enabled()
        for m in self.modules():
            if type(m) is not NormalLinear or not m.weight.mean.is_cuda or not ops.dense_eligible(m.weight.mean):
                continue
            if (m.compute or _settings.get_compute()) == "bf16":
                todo.append(m)
            elif ops.DENSE_X3_F32 and (infer or not m._trainable()) and ctx.base_batch >= 64 and \
                    (m.weight.mean.shape[0] > 16 or m.weight.mean.shape[1] <= 2048):
                todo3.append(m)         # fp32 parity mode, inference: the same plan with three-plane draws (ops.linear_sampled_x3)
        if not todo and not todo3:
            return []

        def specs_of(mods):
            # The plan draws on FRESH keys but does not touch the layers: a layer adopts its keys only when it consumes the
            # drawn weights with sample=True (NormalLinear.forward); one called with sample=False, or never reached by
            # `_forward`, keeps its recorded draw or its user-assigned `.sampled` (dense.py:56-58: `if sample: self.sample()`).
            specs = []
            for m in mods:
                kw, kb = m._fresh_keys(ctx.samples, ctx.sample0)
                specs.append((m.weight.mean.detach(), m.weight.scale.detach(),
                              m.bias.mean.detach() if m.bias is not None else None,
                              m.bias.scale.detach() if m.bias is not None else None, kw, kb))
            return specs

        kl = ops._tls.kl_carry
        pre = []
        if todo:
            specs = specs_of(todo)
            if ops.DRAW_SIDE_STREAM and len(specs) > 1:
                # the first layer's weights on the main stream; the rest (and the KL's first pass) on a side stream, where the
                # VALU-bound draw runs beside the first layer's draw and DMA / MFMA-bound contraction
                dev = specs[0][0].device
                pre = ops.draw_layers(specs[:1], ctx.samples)
                pre += ops.draw_layers(specs[1:], ctx.samples, kl=kl, stream=ops.side_stream(dev))
            else:
                pre = ops.draw_layers(specs, ctx.samples, kl=kl)
        if todo3:
            # fp32 parity mode: the network input's three bf16 planes (what the first dense layer would launch bnn_split_bf16x3
            # for) ride in the same launch; the layer that is called with this very tensor takes them (ctx.x_planes)
            split = None
            if x is not None and x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda and x.stride(1) == 1 and x.is_contiguous() and \
                    x.shape[1] % 8 == 0 and x.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and x.requires_grad):
                split = x
            pre += ops.draw_layers(specs_of(todo3), ctx.samples, kl=kl if (kl is not None and not kl.launched) else None, x3=True, split=split)
            if split is not None and ops._tls.last_split is not None:
                ctx.x_planes = (x, ops._tls.last_split)
                ops._tls.last_split = None
        todo = todo + todo3
        if kl is not None and kl.launched:
            ops._tls.kl_carry = None
        for m, p_ in zip(todo, pre):
            m._predrawn = (ctx, p_)
        return todo

    def forward_stacked(self, x, samples=None, sample0=0, *args, **kwargs):
        """(S, B, ...) tensor of all MC outputs (batched path when enabled)."""
        if samples is None:
            samples = self.samples
        if self.mc_batched and x.is_cuda:
            return self._forward_batched_stacked(x, samples, sample0, *args, **kwargs)
        out = [self._forward(x, *args, **kwargs) for _ in range(samples)]
        return torch.stack(out)
