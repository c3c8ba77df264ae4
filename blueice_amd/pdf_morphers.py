"""Morphers: interpolate model-derived tensors between anchor models -- on the device.

Extension point mirrored from the reference (blueice/pdf_morphers.py:15-39,196):

    MORPHERS[name](config, shape_parameters)
        .get_anchor_points(bounds, n_models=None) -> list of z tuples
        .make_interpolator(f, extra_dims, anchor_models) -> callable(zs) -> ndarray(extra_dims)

`GridInterpolator` keeps the anchor tensor resident in HBM and evaluates the multilinear
interpolation of scipy's RegularGridInterpolator (the reference's backend, pdf_morphers.py:67-70)
with a HIP kernel, bit-identically.  The fused likelihood does not go through these callables at
all: BinnedLogLikelihood streams the anchor models into one DeviceContext (`stream_to_device`) and
evaluates morph + Poisson reduction in a single kernel without materialising the morphed tensor.

The reference's experimental RadialInterpolator (pdf_morphers.py:83-148) is out of scope.
"""
import numpy as np

from .device import DeviceContext
from .exceptions import NoShapeParameters
from .utils import arrays_to_grid

__all__ = ['Morpher', 'GridInterpolator', 'DeviceInterpolator', 'MORPHERS']


class Morpher:
    def __init__(self, config, shape_parameters):
        """shape_parameters: OrderedDict name -> (anchors dict z -> setting, log_prior, base_value)."""
        self.config = config
        self.shape_parameters = shape_parameters
        if not len(shape_parameters):
            raise NoShapeParameters("Attempt to initialize a morpher without shape parameters")

    def get_anchor_points(self, bounds, n_models=None):
        raise NotImplementedError

    def make_interpolator(self, f, extra_dims, anchor_models):
        raise NotImplementedError


class DeviceInterpolator:
    """callable(zs) -> ndarray(extra_dims), backed by an anchor tensor in HBM."""

    def __init__(self, anchor_z_arrays, anchor_scores, extra_dims, device=None):
        self.extra_dims = tuple(int(x) for x in extra_dims)
        self.grid = [np.asarray(g, dtype=float) for g in anchor_z_arrays]
        grid_shape = tuple(len(g) for g in self.grid)
        n = int(np.prod(self.extra_dims, dtype=np.int64)) if self.extra_dims else 1
        self.ctx = DeviceContext(device)
        self.ctx.upload_model(self.grid, np.asarray(anchor_scores, dtype=float).reshape(grid_shape + (1, n)),
                              np.zeros(grid_shape + (1,)))

    def __call__(self, zs):
        out = self.ctx.interpolate('ps', np.asarray(zs, dtype=float))
        if not self.extra_dims:
            return out.reshape(())[()]
        return out.reshape(self.extra_dims)


class GridInterpolator(Morpher):
    """Full Cartesian grid of anchors, multilinear interpolation in between."""

    def __init__(self, config, shape_parameters):
        super().__init__(config, shape_parameters)
        self.anchor_z_arrays = [np.array(sorted(anchors.keys())) for anchors, _, _ in shape_parameters.values()]
        self.anchor_z_grid = arrays_to_grid(self.anchor_z_arrays)
        self.grid_shape = self.anchor_z_grid.shape[:-1]

    def anchor_items(self):
        """(linear index, multi index, z tuple) of every anchor, C order."""
        for lin, multi in enumerate(np.ndindex(*self.grid_shape)):
            yield lin, multi, tuple(self.anchor_z_grid[multi])

    def get_anchor_points(self, bounds=None, n_models=None):
        return [zs for _, _, zs in self.anchor_items()]

    def make_interpolator(self, f, extra_dims, anchor_models):
        extra_dims = list(extra_dims)
        scores = np.zeros(tuple(self.grid_shape) + tuple(extra_dims))
        for _, multi, zs in self.anchor_items():
            scores[multi] = f(anchor_models[zs])
        return DeviceInterpolator(self.anchor_z_arrays, scores, extra_dims, device=self.config.get('device'))

    def stream_to_device(self, ctx, anchor_models, n_sources, n_bins, bb_source=-1, rows_of=None):
        """Fill `ctx` with (template rows, expected events[, MC counts of the BB source]) of every anchor
        model, one anchor at a time -- the dense host tensor of pdf_morphers.py:59 never exists.
        rows_of(model) -> (rows [S, n_bins], MC counts [S, n_bins] or None); default: the PMF grids."""
        rows_of = rows_of or (lambda m: m.pmf_grids())
        ctx.begin_model(self.anchor_z_arrays, n_sources, n_bins, bb_source=bb_source)
        for lin, _, zs in self.anchor_items():
            m = anchor_models[zs]
            rows, n_mc = rows_of(m)
            ctx.set_anchor(lin, rows, m.expected_events(), n_mc[bb_source] if bb_source >= 0 else None)
        ctx.end_model()


MORPHERS = {cls.__name__: cls for cls in (GridInterpolator,)}
