"""MI355X-native variational-layer engine with the pytorch_bayesian Module API.

    from bayesianneuralnetworks_amd.nn import NormalLinear, NormalConv2d, KLDivergence, ...

CUDA/HIP tensors run on hand-written gfx950 kernels (libbnn_hip.so, C-ABI in
include/bnn_hip.h) with no fallback.  See DESIGN.md.
"""
from . import nn, prune, utils, optim
from ._rng import manual_seed
from .nn._settings import set_compute, get_compute

__version__ = '0.0.4+mi355x.r1'

__all__ = ['nn', 'prune', 'utils', 'optim', '__version__', 'manual_seed', 'set_compute', 'get_compute']
