"""The reference's own tests that had no counterpart by name elsewhere in this suite, restated against this package
(reference: tests/test_model.py::test_rates, tests/test_source.py::test_mcsource, tests/test_utils.py::
test_arrays_to_grid, tests/test_binned_likelihood.py::test_twobin_mc, tests/test_likelihood.py::test_no_shape_params /
test_shape_params, tests/test_inference.py::test_limit).  The host-side ones run without a GPU; everything that calls
a likelihood is a GPU test (there is no CPU path)."""
import numpy as np
import pytest
from scipy import stats


def test_expected_events_follow_the_source_settings():
    from blueice_amd import Model
    from blueice_amd.test_helpers import conf_for_test
    m = Model(conf_for_test(n_sources=1))
    np.testing.assert_array_equal(m.expected_events(), [1000])
    for source in m.sources:
        source.config['livetime_days'] = 2
    np.testing.assert_array_equal(m.expected_events(), [2000])
    for source in m.sources:
        source.config['livetime_days'] = 1
    m.sources[0].fraction_in_range = 0.5
    np.testing.assert_array_equal(m.expected_events(), [500])
    m.sources[0].fraction_in_range = 1
    m.config['some_multiplier'] = 2                   # after the fact: no effect, the source read it when it was built
    np.testing.assert_array_equal(m.expected_events(), [1000])
    conf = conf_for_test(n_sources=2)
    conf['some_multiplier'] = 2
    m = Model(conf)
    np.testing.assert_array_equal(m.expected_events(), [2000, 2000])
    assert m.get_source(1) is m.sources[1] and m.get_source('s1') is m.sources[1]
    assert m.get_source_i(1) == 1 and m.get_source_i('s1') == 1
    conf = conf_for_test(n_sources=1)
    conf['strlen_multiplier'] = 'hi'                  # a non-numeric setting: the test source multiplies by its length
    np.testing.assert_array_equal(Model(conf).expected_events(), [2000])


def test_monte_carlo_source_density():
    from blueice_amd import Model
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(0)
    conf = conf_for_test(mc=True)
    s = Model(conf).sources[0]
    bins = conf['analysis_space'][0][1]
    assert s.events_per_day == 1000
    assert s.fraction_in_range > 0.9999
    assert abs(s.pdf([0]) - stats.norm.pdf(0)) < 0.01
    # linear interpolation between bin centres ... which are the edges' midpoints: halfway between two edges the
    # density is the mean of the densities at the edges
    assert (s.pdf([bins[0]]) + s.pdf([bins[1]])) / 2 == s.pdf([(bins[0] + bins[1]) / 2])


def test_arrays_to_grid():
    from blueice_amd.utils import arrays_to_grid
    np.testing.assert_array_equal(arrays_to_grid([np.array([0, 1]), np.array([0, 1])]),
                                  [[[0, 0], [0, 1]], [[1, 0], [1, 1]]])
    np.testing.assert_array_equal(arrays_to_grid([np.array([1, 2]), np.array([3, 4])]),
                                  [[[1, 3], [1, 4]], [[2, 3], [2, 4]]])


@pytest.mark.gpu
def test_two_bins_of_a_monte_carlo_template():
    from blueice_amd import BinnedLogLikelihood
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(1)
    lf = BinnedLogLikelihood(conf_for_test(mc=True, analysis_space=[['x', [-40, 0, 40]]]))
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.ones(100, dtype=[('x', float), ('source', int)]))          # 100 events at x = 1
    want = stats.poisson(500).logpmf(100) + stats.poisson(500).logpmf(0)
    assert abs(lf() - want) <= 1e-2 * abs(want)


@pytest.mark.gpu
def test_unbinned_likelihood_without_shape_parameters():
    from blueice_amd import UnbinnedLogLikelihood
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(2)
    for mc in (False, True):                          # (a Monte Carlo source computes its pdf on first use)
        lf = UnbinnedLogLikelihood(conf_for_test(mc=mc))
        d = lf.base_model.simulate()
        lf.prepare()
        lf.set_data(d)
        assert np.isfinite(lf())


@pytest.mark.gpu
def test_unbinned_shape_parameter_with_non_numeric_settings():
    from blueice_amd import UnbinnedLogLikelihood
    from blueice_amd.exceptions import InvalidParameterSpecification
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(3)
    lf = UnbinnedLogLikelihood(conf_for_test(n_sources=1))
    lf.add_rate_parameter('s0')
    with pytest.raises(InvalidParameterSpecification):
        lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'})
    lf.add_shape_parameter('strlen_multiplier', {1: 'q', 2: 'hi', 3: 'wha'}, base_value=1)
    d = lf.base_model.simulate()
    lf.prepare()
    lf.set_data(d)
    assert len(lf.anchor_models) == 3
    with pytest.raises(ValueError):
        lf(strlen_multiplier='hi')                    # the raw setting is not a parameter value
    lf(strlen_multiplier=1.5)                         # its representative number is
    assert lf() == lf(strlen_multiplier=1)            # the base value
    assert lf(strlen_multiplier=1.5) < lf()           # interpolated between the representatives


@pytest.mark.gpu
def test_every_kind_of_interval_runs():
    from blueice_amd import UnbinnedLogLikelihood
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(4)
    lf = UnbinnedLogLikelihood(conf_for_test(n_sources=2))
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(lf.base_model.simulate())
    up = lf.one_parameter_interval(target='s0_rate_multiplier', kind='upper', bound=40)
    lo = lf.one_parameter_interval(target='s0_rate_multiplier', kind='lower', bound=0.1)
    a, b = lf.one_parameter_interval(target='s0_rate_multiplier', kind='central', bound=(0.1, 20))
    assert 0.1 < lo < 1.3 and 0.7 < up < 40 and lo < up and a < b and a < up and lo < b
