"""Signal-to-noise pruning (pytorch_bayesian/prune/prune.py:5-22).  On the device the score log N(0; mu, sigma) is
a HIP kernel (bnn_prune_score); the top-k selection and the masked assignment are torch ops on the device."""
import torch

from ..utils import apply_wb


class PruneNormal:

    def __call__(self, module, percentage=0.5):
        self.prune(module, percentage)

    def prune_param(self, param, percentage):
        """prune.py:10-17: the `percentage` entries whose posterior puts the most density on 0
        get mean = 0, scale = -30.  (log_prob is given a tensor: current torch rejects the
        reference's Python-int argument.)"""
        if param.mean.is_cuda:
            from .. import ops
            log_prob = ops.prune_score(param.mean, param.scale)
        else:
            zero = torch.zeros((), device=param.mean.device, dtype=param.mean.dtype)
            log_prob = param.dist.log_prob(zero)
        flat = log_prob.flatten()
        k = int(percentage * flat.size(0))
        _, idx = torch.topk(flat, k)
        mask = torch.zeros_like(flat).scatter(0, idx, 1).bool().view(log_prob.shape)
        param.mean[mask] = 0
        param.scale[mask] = -30

    def prune(self, module, percentage=0.5):
        with torch.no_grad():
            module.traverse(lambda m: apply_wb(m, self.prune_param, percentage))
