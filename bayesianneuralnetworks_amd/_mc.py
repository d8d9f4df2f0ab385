"""Monte-Carlo batching context.

BayesianNetworkModule.forward (pytorch_bayesian/nn/container.py:32-37) runs `_forward`
`samples` times in a Python loop.  When a network opts in (mc_batched=True) the loop is
replaced by ONE pass in which every Bayesian layer launches all S samples in one grid:
rows [s*B, (s+1)*B) of an activation belong to MC sample s.  A Bayesian layer that still
sees the un-replicated batch (B rows) reads it with sample stride 0, so the deterministic
prefix of a network (e.g. the stock Conv2d stack of examples/MNIST/model.py:21-27) runs
once instead of S times.
"""
import threading

_state = threading.local()


class McContext:
    def __init__(self, samples, base_batch, sample0=0):
        self.samples = samples
        self.base_batch = base_batch
        self.sample0 = sample0
        # set by BayesianNetworkModule.predictive_mean: the caller reduces the outputs over the MC axis itself, so a hidden layer
        # may fuse the classifier head behind it and hand on partial logits (ops.HeadPartials) instead of a tensor
        self.lazy_head = False
        # (x, planes): the network input and its three bf16 planes, split by the draw plan's launch (fp32 parity mode)
        self.x_planes = None

    def __enter__(self):
        self.prev = getattr(_state, "ctx", None)
        _state.ctx = self
        return self

    def __exit__(self, *exc):
        _state.ctx = self.prev
        return False


def current():
    return getattr(_state, "ctx", None)
