"""Exception types of the likelihood API (same names as the reference's blueice/exceptions.py:1-32,
including its historical spelling `NoOpimizationNecessary`, so user `except` clauses keep working)."""


__all__ = ['BlueIceException', 'NoOpimizationNecessary', 'OptimizationFailed', 'NotPreparedException',
           'NoShapeParameters', 'InvalidParameter', 'InvalidParameterSpecification', 'PDFNotComputedException',
           'DeviceError']


class BlueIceException(Exception):
    """Root of all errors raised on purpose by this package."""


class NoOpimizationNecessary(BlueIceException):
    """make_objective was asked to fit with every parameter fixed."""


class OptimizationFailed(BlueIceException):
    """The minimizer (and its Nelder-Mead retry) did not converge."""


class NotPreparedException(BlueIceException):
    """prepare() / set_data() has to be called before this operation."""


class NoShapeParameters(BlueIceException):
    """A morpher was created without any shape parameter."""


class InvalidParameter(BlueIceException):
    """The likelihood function has no parameter of that name."""


class InvalidParameterSpecification(BlueIceException):
    """add_shape_parameter / add_rate_parameter was called inconsistently."""


class PDFNotComputedException(BlueIceException):
    """A source's pdf was used before compute_pdf ran."""


class DeviceError(BlueIceException):
    """libblueice_hip reported a failure (missing library, no GPU, HIP error)."""
