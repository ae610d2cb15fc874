"""CPU tests of the restatement of interpolate_missing_data (oracle/interp_ref.py; reference
/root/reference/gadfly/interp.py:6-60) on hand-built series."""
import numpy as np

from oracle import interp_ref


def test_fills_single_and_multiple_gaps():
    t = np.array([0.0, 1.0, 2.0, 4.0, 5.0, 9.0, 10.0])
    f = np.array([0.0, 10.0, 20.0, 40.0, 50.0, 90.0, 100.0])
    tt, ff = interp_ref.interpolate_missing_data(t, f)
    np.testing.assert_array_equal(tt, np.arange(11.0))
    np.testing.assert_allclose(ff, 10.0 * np.arange(11.0), rtol=1e-15)


def test_complete_series_is_returned_unchanged():
    t = 3.0 + 0.25 * np.arange(50)
    f = np.sin(t)
    tt, ff = interp_ref.interpolate_missing_data(t, f)
    np.testing.assert_array_equal(tt, t)
    np.testing.assert_array_equal(ff, f)


def test_given_cadence_numbers_and_jittered_times():
    rng = np.random.default_rng(3)
    cad = np.sort(rng.choice(np.arange(1000, 1400), 300, replace=False))
    t = 54000.0 + cad * 0.0204 + rng.uniform(-1e-4, 1e-4, cad.size)
    f = np.cos(0.01 * cad)
    tt, ff = interp_ref.interpolate_missing_data(t, f, cadences=cad)
    assert len(tt) == cad[-1] - cad[0] + 1 and np.all(np.diff(tt) > 0)
    # the original points are still there, the new ones lie on the chords
    keep = np.isin(tt, t)
    np.testing.assert_array_equal(tt[keep], t)
    np.testing.assert_array_equal(ff[keep], f)
    np.testing.assert_allclose(ff[~keep], np.interp(tt[~keep], t, f), rtol=0, atol=0)
    # without the cadence numbers the same grid is found from the times
    tt2, ff2 = interp_ref.interpolate_missing_data(t, f)
    assert len(tt2) == len(tt)


def test_grid_drift_reorders_by_time():
    """With cadence numbers given, dt is a median and the grid t0 + m dt can drift past a
    neighbouring time stamp: the reference merges BY TIME (interp.py:56-59), so the missing
    cadence lands before the point that precedes it in cadence order, with the flux of the
    interval that holds its grid time."""
    cad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
    t = cad * 1.0
    t[7] = 8.2                       # a late time stamp; cadence 8 is missing, grid time 8.0 < t[7]
    f = np.arange(10.0) ** 2
    tt, ff = interp_ref.interpolate_missing_data(t, f, cadences=cad)
    assert len(tt) == 11 and np.all(np.diff(tt) > 0)
    k = int(np.flatnonzero(tt == 8.0)[0])
    assert tt[k + 1] == 8.2          # the missing cadence comes BEFORE the late point
    assert ff[k] == np.interp(8.0, t, f) and t[6] < 8.0 < t[7]


def test_product_function_validates_before_touching_the_device():
    """gadfly_amd.interpolate_missing_data rejects malformed input on the host (no GPU needed)."""
    import pytest
    import gadfly_amd
    with pytest.raises(ValueError):
        gadfly_amd.interpolate_missing_data(np.array([0.0, 2.0, 1.0]), np.zeros(3))
    with pytest.raises(ValueError):
        gadfly_amd.interpolate_missing_data(np.arange(4.0), np.zeros(3))
    with pytest.raises(ValueError):
        gadfly_amd.interpolate_missing_data(np.arange(4.0), np.zeros(4), cadences=np.arange(3))
