"""GPU parity of gadfly_amd.interpolate_missing_data (gf_interp_plan / gf_interp_fill) against the
numpy restatement of /root/reference/gadfly/interp.py:6-60: bit-identical times and fluxes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _series(n_full, frac_missing, seed, jitter=0.0, runs=True):
    rng = np.random.default_rng(seed)
    keep = rng.uniform(size=n_full) > frac_missing
    if runs and n_full > 1000:                  # a few long gaps (quarter boundaries)
        for s in rng.integers(0, n_full - 500, 4):
            keep[s:s + int(rng.integers(2, 400))] = False
    keep[[0, -1]] = True
    cad = np.flatnonzero(keep) + 7000
    t = 2454833.0 + cad * (58.85 / 86400.0) + jitter * rng.uniform(-1, 1, cad.size) / 86400.0
    f = 1e4 + 50 * np.sin(cad * 0.01) + rng.normal(size=cad.size)
    return t, f, cad


@pytest.mark.parametrize("n,frac,jitter", [(5000, 0.02, 0.0), (5000, 0.3, 2.0), (100000, 0.05, 0.5),
                                           (2049, 0.0, 0.0), (300, 0.5, 0.0)])
@pytest.mark.parametrize("with_cadences", [False, True])
def test_matches_numpy_bit_for_bit(hip, n, frac, jitter, with_cadences):
    import gadfly_amd
    from oracle import interp_ref
    t, f, cad = _series(n, frac, seed=n + int(100 * frac), jitter=jitter, runs=frac > 0)
    kw = {"cadences": cad} if with_cadences else {}
    tt, ff = gadfly_amd.interpolate_missing_data(t, f, **kw)
    rt, rf = interp_ref.interpolate_missing_data(t, f, **kw)
    assert tt.shape == rt.shape
    np.testing.assert_array_equal(tt, rt)
    np.testing.assert_array_equal(ff, rf)
    if frac == 0.0:
        assert len(tt) == len(t)


def _case_names():
    from tests.interp_cases import CASES
    return list(CASES)


@pytest.mark.parametrize("name", _case_names())
def test_matches_reference_fixtures(hip, name):
    """The HIP kernels against the outputs of the REFERENCE's own interpolate_missing_data
    (tests/golden/reference/interp_reference.npz, made by tests/golden/reference/make_interp_golden.py from
    /root/reference/gadfly/interp.py): bit-identical times and fluxes."""
    import gadfly_amd
    from tests.test_interp_reference import check_against_fixture
    check_against_fixture(name, gadfly_amd.interpolate_missing_data)


def test_full_size_properties_and_device_output(hip):
    """N = 2e6 cadences with 10 % missing: the filled grid is complete and in time order, the input
    points are untouched, new points lie on the chords; the output can stay on the device and feed the
    power spectrum directly."""
    import gadfly_amd
    t, f, cad = _series(2_000_000, 0.1, seed=1)
    td, fd = gadfly_amd.interpolate_missing_data(t, f, cadences=cad, return_device=True)
    assert td.is_cuda and len(td) == cad[-1] - cad[0] + 1
    tt, ff = td.cpu().numpy(), fd.cpu().numpy()
    dt = np.median(np.diff(t) / np.diff(cad))
    # (the grid t0 + m dt drifts against the time stamps -- dt is a median -- by less than half a cadence here)
    assert np.all(np.diff(tt) > 0) and np.all(np.abs(np.diff(tt) - dt) < 0.5 * dt)
    pos = cad - cad[0]
    np.testing.assert_array_equal(tt[pos], t)
    np.testing.assert_array_equal(ff[pos], f)
    new = np.ones(len(tt), bool); new[pos] = False
    np.testing.assert_array_equal(ff[new], np.interp(tt[new], t, f))
    ps = gadfly_amd.PowerSpectrum.from_flux(fd - fd.mean(), dt * 86400e-6)
    assert ps.power.shape == (len(tt) // 2,) and np.all(np.isfinite(ps.power))


def test_grid_drift_reorders_by_time(hip):
    """The hand-built drift case of tests/test_interp_oracle.py on the device."""
    import gadfly_amd
    from oracle import interp_ref
    cad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
    t = cad * 1.0
    t[7] = 8.2
    f = np.arange(10.0) ** 2
    tt, ff = gadfly_amd.interpolate_missing_data(t, f, cadences=cad)
    rt, rf = interp_ref.interpolate_missing_data(t, f, cadences=cad)
    np.testing.assert_array_equal(tt, rt)
    np.testing.assert_array_equal(ff, rf)


def test_rejects_bad_input(hip):
    import gadfly_amd
    with pytest.raises(ValueError):
        gadfly_amd.interpolate_missing_data(np.array([0.0, 2.0, 1.0]), np.zeros(3))
    with pytest.raises(ValueError):
        gadfly_amd.interpolate_missing_data(np.arange(4.0), np.zeros(3))
