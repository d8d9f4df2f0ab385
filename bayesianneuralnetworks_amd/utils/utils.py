"""Plumbing helpers with the behaviour of pytorch_bayesian/utils/utils.py (reference
file:line cited per function).  Pure Python, no compute."""
from collections.abc import Iterable

from torch.nn import ModuleDict, ModuleList, Sequential


def _item_or_list(n):
    """utils.py:10-11 -- a one-element sequence collapses to its element."""
    if len(n) == 1:
        return n[0]
    return n


def _make_ntuple(count):
    def to_tuple(value):
        # utils.py:14-20 -- iterables pass through untouched, scalars are repeated
        if isinstance(value, Iterable):
            return value
        return (value,) * count
    return to_tuple


_single = _make_ntuple(1)   # utils.py:22
_pair = _make_ntuple(2)     # utils.py:23
_triple = _make_ntuple(3)   # utils.py:24


def apply_wb(module, fn, *args, pass_module=False, pass_type=False, **kwargs):
    """utils.py:30-47 -- call fn on module.weight then module.bias (skipping None),
    optionally passing module= and type='w'|'b'; collect non-None results; None if empty."""
    extra = dict(kwargs)
    if pass_module:
        extra['module'] = module
    collected = []
    for tag, param in (('w', module.weight), ('b', module.bias)):
        if param is None:
            continue
        if pass_type:
            extra['type'] = tag
        out = fn(param, *args, **extra)
        if out is not None:
            collected.append(out)
    return collected or None


def traverse(module, fn, *args, **kwargs):
    """utils.py:50-67 -- depth-first over Sequential / ModuleList / ModuleDict /
    BayesianNetworkModule only; fn runs on BayesianModule leaves and must return a list.
    Anything else (and empty results) yields None."""
    from ..nn.container import BayesianModule, BayesianNetworkModule
    if isinstance(module, (ModuleList, ModuleDict, Sequential, BayesianNetworkModule)):
        found = []
        for child in module.children():
            sub = traverse(child, fn, *args, **kwargs)
            if sub is not None:
                found += sub
        return found or None
    if isinstance(module, BayesianModule):
        out = fn(module, *args, **kwargs)
        if isinstance(out, list) and out:
            return out
    return None
