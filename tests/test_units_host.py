"""
Host-side unit handling of ``GaussianProcess`` (reference
/root/reference/gadfly/gp.py:22-59 light-curve branch, :61-86, :88-126, :128-165, compute's
``diag`` quirk :196-201).  No GPU: the factorisation is stubbed out, everything before it is the
product's own host code.  astropy is not installed in the image, so the Quantity branches run on
a ~40-line stand-in (``Q`` / ``Unit`` below: an ndarray subclass carrying ``.unit``, with ``.to``,
``.value``, ``is_equivalent`` and unit algebra for * and /) patched in as ``gadfly_amd.units.u``.
"""
import types

import numpy as np
import pytest

import gadfly_amd
from gadfly_amd import units as gunits
from gadfly_amd.synth import solar_like_hyperparameters


from tests.fake_units import Unit, Q, ONE, fake_astropy_units  # noqa: E402,F401  (the astropy stand-in)


@pytest.fixture
def units(monkeypatch):
    u = fake_astropy_units()
    monkeypatch.setattr(gunits, "u", u)
    monkeypatch.setattr(gunits, "HAS_ASTROPY", True)
    return u


class HostOnlyGP(gadfly_amd.GaussianProcess):
    """The drop-in class with the device factorisation stubbed out: what is left is the host logic
    under test (unit stripping, diagonal assembly, light-curve bookkeeping)."""

    def _do_compute(self, quiet):
        self.computed_quiet = quiet


@pytest.fixture
def kernel():
    return gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(6), texp=60.0)


# ---- the light_curve= branch (gp.py:43-57) -------------------------------------------------------
def test_light_curve_with_plain_arrays(kernel):
    """Duck-typed light curve with ndarray fields: times pass through (already 1/uHz), the flux
    median is cached, flux_err becomes yerr (squared onto the diagonal)."""
    n = 50
    lc = types.SimpleNamespace(time=np.arange(n) * 60e-6,
                               flux=1e4 + np.arange(n, dtype=float),
                               flux_err=np.full(n, 3.0))
    gp = HostOnlyGP(kernel, light_curve=lc)
    assert gp._original_flux_median == np.median(lc.flux)
    assert gp._t is not None and np.array_equal(gp._t, lc.time)
    np.testing.assert_array_equal(gp._diag, 9.0 * np.ones(n))
    assert gp._size == n and np.all(gp.mean_value == 0.0)
    # masked fluxes: the reference takes nanmedian of `.unmasked` (gp.py:49-52)
    raw = np.where(np.arange(n) % 7 == 0, np.nan, lc.flux)
    masked_flux = types.SimpleNamespace(unmasked=raw)
    lc2 = types.SimpleNamespace(time=lc.time, flux=masked_flux, flux_err=lc.flux_err)
    gp2 = HostOnlyGP(kernel, light_curve=lc2)
    assert gp2._original_flux_median == np.nanmedian(raw)
    # an explicit t is overridden by the light curve's times, as in the reference (gp.py:47)
    gp3 = HostOnlyGP(kernel, t=np.arange(n) * 1.0, light_curve=lc)
    assert np.array_equal(gp3._t, lc.time)


def test_light_curve_in_electrons_per_second(kernel, units):
    """lightkurve-style light curve: time in days, flux and flux_err in e-/s.  Times -> 1/uHz,
    errors -> 1e6 * err / median [ppm]; fluxes -> 1e6 (f / median - 1) (gp.py:115-124)."""
    u = units
    n = 40
    rng = np.random.default_rng(0)
    raw = 5e4 + 30.0 * rng.normal(size=n)
    lc = types.SimpleNamespace(time=Q(np.arange(n) / 1440.0, u.day),
                               flux=Q(raw, u.electron / u.s),
                               flux_err=Q(np.full(n, 25.0), u.electron / u.s))
    gp = HostOnlyGP(kernel, light_curve=lc)
    med = np.median(raw)
    assert float(gp._original_flux_median.value) == med
    assert gp._original_flux_median.unit.is_equivalent(u.electron / u.s)
    np.testing.assert_allclose(gp._t, np.arange(n) * 60e-6, rtol=1e-13)      # days -> 1e6 s
    np.testing.assert_allclose(gp._diag, (1e6 * 25.0 / med) ** 2, rtol=1e-14)
    y = gp._flux_to_ppm(lc.flux)
    assert isinstance(y, np.ndarray) and not hasattr(y, "unit")
    np.testing.assert_allclose(y, 1e6 * (raw / med - 1.0), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(gp._flux_to_ppm(lc.flux_err, is_error=True), 1e6 * 25.0 / med)
    # and back: power 1 is a flux, power 2 a variance with the reference's (sic) median * unit
    back = gp._ppm_to_flux(y)
    assert back.unit.is_equivalent(u.electron / u.s)
    np.testing.assert_allclose(back.value, raw, rtol=1e-12)
    var = gp._ppm_to_flux(np.array([4.0, 9.0]), power=2)
    np.testing.assert_allclose(var.value, 1e-6 * np.array([4.0, 9.0]) * med, rtol=1e-14)
    assert var.unit.is_equivalent((u.electron / u.s) ** 2)


def test_flux_and_time_quantities(kernel, units):
    u = units
    gp = HostOnlyGP(kernel)
    # relative fluxes in ppm (or any dimensionless unit) are converted by scale (gp.py:126)
    np.testing.assert_allclose(gp._flux_to_ppm(Q([1e-3, 2e-6], ONE)), [1e3, 2.0])
    np.testing.assert_allclose(gp._flux_to_ppm(Q([5.0, 7.0], u.cds.ppm)), [5.0, 7.0])
    # times: any time unit -> 1/uHz; a frequency unit can be chosen (gp.py:61-86)
    np.testing.assert_allclose(gp._time_to_freq(Q([0.0, 1.0], u.day)), [0.0, 0.0864])
    np.testing.assert_allclose(gp._time_to_freq(Q([60.0], u.s)), [60e-6])
    np.testing.assert_allclose(gp._time_to_freq(Q([2.0], u.min), freq_unit=Unit(1.0, s=-1)), [120.0])
    # without a cached median the result is a ppm Quantity regardless of the power (gp.py:163-165)
    for power in (1, 2):
        out = gp._ppm_to_flux(np.array([1.0, 2.0]), power=power)
        assert out.unit is u.cds.ppm and np.array_equal(out.value, [1.0, 2.0])


def test_ppm_to_flux_with_cached_median_minimal_unit_object(kernel):
    """No astropy at all: a float carrying ``.unit`` is enough for both powers (gp.py:146-161)."""
    class Median(float):
        unit = 3.0                              # stands for the median's unit in the (sic) product

    gp = HostOnlyGP(kernel)
    gp._original_flux_median = Median(2.0e4)
    v = np.array([0.0, 1e6, -5e5])
    np.testing.assert_array_equal(gp._ppm_to_flux(v), (1e-6 * v + 1) * 2.0e4)
    np.testing.assert_array_equal(gp._ppm_to_flux(v, power=2), (1e-6 * v) * 2.0e4 * 3.0)


def test_return_quantity_without_astropy_raises_clearly(kernel):
    gp = HostOnlyGP(kernel)
    if not gunits.HAS_ASTROPY:
        with pytest.raises(ImportError, match="astropy"):
            gp._ppm_to_flux(np.zeros(3))


# ---- compute(): diagonal assembly and the reference's diag quirk (gp.py:196-201) -------------------
def test_compute_diag_quirk(kernel, units):
    u = units
    gp = HostOnlyGP(kernel)
    t = np.arange(6) * 60e-6
    gp.compute(t, yerr=2.0)
    np.testing.assert_array_equal(gp._diag, 4.0 * np.ones(6))
    gp.compute(t, diag=np.arange(6.0), quiet=True)
    np.testing.assert_array_equal(gp._diag, np.arange(6.0))
    assert gp.computed_quiet is True
    gp.compute(t)
    np.testing.assert_array_equal(gp._diag, np.zeros(6))
    # yerr with a unit is converted as an error ...
    gp.compute(t, yerr=Q(np.full(6, 3e-6), ONE))
    np.testing.assert_allclose(gp._diag, 9.0)
    # ... but diag is converted ONLY when *yerr* carries a unit (gp.py:200 tests yerr, not diag):
    # a Quantity diag with yerr=None is used as the bare numbers it holds
    gp.compute(t, diag=Q(np.full(6, 2e-6), ONE))
    np.testing.assert_allclose(gp._diag, 2e-6)
    # both given -> celerite2's ValueError, after the quirk converted diag without the error flag
    with pytest.raises(ValueError, match="only one"):
        gp.compute(t, yerr=Q(np.full(6, 3e-6), ONE), diag=Q(np.full(6, 2e-6), ONE))
    # Quantity times through compute
    gp.compute(Q(np.arange(6) / 1440.0, u.day))
    np.testing.assert_allclose(gp._t, t, rtol=1e-13)
