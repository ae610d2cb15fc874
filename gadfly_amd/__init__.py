"""
gadfly_amd -- MI355X-native implementation of gadfly's GP hot path.

Public names follow /root/reference/gadfly/__init__.py:3-6 (``core`` and ``gp`` re-exported).
Importing the package needs neither a GPU nor the built HIP library; the first compute
call does, and fails loudly without them (no CPU fallback).
"""
from .core import (  # noqa: F401
    Hyperparameters, StellarOscillatorKernel, SolarOscillatorKernel,
    ShotNoiseKernel, Filter,
)
from . import scale  # noqa: F401
from .gp import GaussianProcess, ConditionalDistribution, LinAlgError  # noqa: F401
from .batch import BatchedLogLikelihood, log_likelihood_batch  # noqa: F401
from . import terms  # noqa: F401
from .psd import PowerSpectrum, bin_power_spectrum  # noqa: F401
from .interp import interpolate_missing_data, stitch_quarters  # noqa: F401

__version__ = "0.1.0"
