"""Drop-in alias: `import pytorch_bayesian` resolves to bayesianneuralnetworks_amd, so the
reference's examples (examples/*/model.py, train.py, prune.py: `from pytorch_bayesian.nn
import NormalLinear, ...`) run unchanged on the MI355X engine."""
import sys

import bayesianneuralnetworks_amd as _impl
from bayesianneuralnetworks_amd import nn, prune, utils, __version__  # noqa: F401

for _name in ("nn", "prune", "utils"):
    sys.modules[__name__ + "." + _name] = getattr(_impl, _name)

__all__ = ['nn', 'prune', 'utils', '__version__']
