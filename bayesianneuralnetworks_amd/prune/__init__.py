"""Pruning of Gaussian posteriors (exports of pytorch_bayesian/prune/__init__.py:1-5).

`PruneNormal()(model, fraction)` zeroes the `fraction` of every posterior tensor whose density at 0 is
highest; on the device the score is the HIP kernel bnn_prune_score (see prune.py)."""
from . import prune as _impl

PruneNormal = _impl.PruneNormal
__all__ = ('PruneNormal',)
