"""Small host-side helpers shared by the Model / Source / likelihood layers."""
from copy import deepcopy

import numpy as np

__all__ = ['combine_dicts', 'arrays_to_grid', 'events_to_analysis_dimensions', 'is_numeric']


def combine_dicts(*dicts, exclude=(), deep_copy=False):
    """Merge dicts left to right (later wins), dropping the keys in `exclude`.
    Same contract as the reference's utils.combine_dicts (blueice/utils.py:27-40)."""
    merged = {}
    for d in dicts:
        merged.update(deepcopy(d) if deep_copy else d)
    for k in exclude:
        merged.pop(k, None)
    return merged


def arrays_to_grid(arrs):
    """n one-dimensional arrays -> array [len_0, .., len_{n-1}, n] of grid coordinates
    ('ij' indexing; reference: blueice/utils.py:150-153)."""
    return np.stack(np.meshgrid(*arrs, indexing='ij'), axis=-1)


def events_to_analysis_dimensions(events, analysis_space):
    """Columns of the record array `events` named by the analysis space, in its order."""
    return [events[name] for name, _ in analysis_space]


def is_numeric(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool) or isinstance(x, (np.integer, np.floating))
