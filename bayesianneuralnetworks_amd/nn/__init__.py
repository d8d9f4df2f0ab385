from .container import BayesianModule, BayesianNetworkModule
from .core import WeightNormal, WeightMultivariateNormal
from .dense import (BayesianLinear, NormalLinear, MultivariateNormalLinear, FlipoutNormalLinear,
                    NormalInverseGaussianLinear, MCDropoutLinear)
from .conv import (BayesianConvNd, NormalConvNd, NormalConv1d, NormalConv2d, NormalConv3d,
                   FlipOutNormalConvNd, FlipOutNormalConv1d, FlipOutNormalConv2d, FlipOutNormalConv3d,
                   MCDropoutConvNd, MCDropoutConv1d, MCDropoutConv2d, MCDropoutConv3d)
from .loss import KLDivergence, Entropy, NormalInverseGaussianLoss, NormalInverseGaussianUncertainty
from ._settings import set_compute, get_compute, fuse_activations, fuse_kl_gradient

# the names pytorch_bayesian/nn/__init__.py:7-35 exports
__all__ = [
    'BayesianModule', 'BayesianNetworkModule', 'WeightNormal', 'WeightMultivariateNormal',
    'BayesianLinear', 'NormalLinear', 'MultivariateNormalLinear', 'FlipoutNormalLinear',
    'NormalInverseGaussianLinear', 'MCDropoutLinear', 'BayesianConvNd', 'NormalConvNd',
    'NormalConv1d', 'NormalConv2d', 'NormalConv3d', 'FlipOutNormalConvNd', 'FlipOutNormalConv1d',
    'FlipOutNormalConv2d', 'FlipOutNormalConv3d', 'MCDropoutConvNd', 'MCDropoutConv1d',
    'MCDropoutConv2d', 'MCDropoutConv3d', 'KLDivergence', 'Entropy', 'NormalInverseGaussianLoss',
    'NormalInverseGaussianUncertainty',
]

# BASELINE.json's north_star uses these two names for the layers above
BayesianConv2d = NormalConv2d
