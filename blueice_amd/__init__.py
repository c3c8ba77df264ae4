"""blueice_amd: the binned-likelihood hot path of blueice, MI355X-native.

Top-level names follow the reference's `blueice/__init__.py` (likelihood, model, source, exceptions).
"""
from .exceptions import *  # noqa: F401,F403
from .model import Model  # noqa: F401
from .source import Source, HistogramPdfSource, DensityEstimatingSource, MonteCarloSource  # noqa: F401
from .likelihood import (LogLikelihoodBase, BinnedLogLikelihood, UnbinnedLogLikelihood,  # noqa: F401
                         LogLikelihoodSum, LogLikelihoodReParam, LogAncillaryLikelihood)

__version__ = '0.1.0'
