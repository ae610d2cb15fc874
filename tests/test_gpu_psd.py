"""
GPU parity of the device power-spectrum path (gadfly_amd.psd: hipFFT + gf_psd_power + gf_psd_bin)
against the CPU restatement of /root/reference/gadfly/psd.py:186-300, :566-587 (oracle/psd_ref.py).
Floating point: power and binned statistics within 1e-10 relative of the numpy/scipy values (the
FFTs differ in rounding by ~1e-15 of the largest coefficient; observed ~1e-13).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _series(n, seed, r=None):
    rng = np.random.default_rng(seed)
    t = np.arange(n) * 60e-6
    shape = (n,) if r is None else (r, n)
    return (300 * np.sin(2 * np.pi * 3000.0 * t) + 50 * rng.normal(size=shape)
            + np.cumsum(rng.normal(size=shape), axis=-1))


def _relmax(a, b):
    return float(np.nanmax(np.abs(a - b)) / np.nanmax(np.abs(b)))


@pytest.mark.parametrize("n", [4096, 10007, 100000, 99999])
@pytest.mark.parametrize("zero", [False, True])
def test_power_matches_numpy(hip, n, zero):
    import gadfly_amd
    from oracle import psd_ref
    flux = _series(n, n)
    ps = gadfly_amd.PowerSpectrum.from_flux(flux, 60e-6, include_zero_freq=zero)
    freq, power, norm = psd_ref.fft_power(flux, 60e-6, include_zero_freq=zero)
    np.testing.assert_array_equal(ps.frequency, freq)
    assert ps.norm == norm and ps.power.shape == power.shape
    # relative to the spectrum's scale: single coefficients far below the peak carry the FFT's
    # absolute rounding error
    assert _relmax(ps.power, power) < 1e-13
    np.testing.assert_allclose(ps.power, power, rtol=1e-6, atol=1e-13 * power.max())
    np.testing.assert_allclose(ps.light_curve_rms, (power * norm) ** 0.5, rtol=1e-6,
                               atol=1e-9 * power.max() ** 0.5)
    np.testing.assert_array_equal(ps.omega, 2 * np.pi * freq)


@pytest.mark.parametrize("n,bins,log,constant", [
    (100000, 15, True, 1), (100000, None, True, 1), (20000, 7, False, 3), (5001, 40, True, 1),
    (3000, 1, True, 1), (1 << 14, 200, True, 2),
])
def test_binning_matches_reference_formulation(hip, n, bins, log, constant):
    """Device binning of a HOST power spectrum against the scipy.binned_statistic formulation."""
    import gadfly_amd
    from oracle import psd_ref
    flux = _series(n, 3 * n)
    freq, power, _ = psd_ref.fft_power(flux, 60e-6)
    kw = {} if constant == 1 else {"constant": constant}
    got = gadfly_amd.PowerSpectrum(freq, power, name="x").bin(bins, log=log, **kw)
    c, s, e = psd_ref.bin_power_lookup(freq, power, bins=bins, log=log, constant=constant)
    assert got.name == "x (binned)"
    np.testing.assert_allclose(got.frequency, c, rtol=1e-15)
    np.testing.assert_array_equal(np.isnan(got.power), np.isnan(s))
    np.testing.assert_allclose(got.power, s, rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(got.error, e, rtol=RTOL, equal_nan=True)


def test_explicit_edges(hip):
    import gadfly_amd
    from oracle import psd_ref
    flux = _series(8000, 5)
    freq, power, _ = psd_ref.fft_power(flux, 60e-6)
    edges = np.array([0.5, 1.0, 2.0, 2.0 + 1e-9, 3.5])
    got = gadfly_amd.bin_power_spectrum(gadfly_amd.PowerSpectrum(freq, power), edges)
    c, s, e = psd_ref.bin_power_lookup(freq, power, bins=edges)
    np.testing.assert_allclose(got.power, s, rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(got.error, e, rtol=RTOL, equal_nan=True)
    with pytest.raises(ValueError):
        gadfly_amd.PowerSpectrum(freq[::-1], power).bin(5)


def test_batch_stays_on_device(hip):
    """(R, N) draws -> power -> bins without leaving the GPU; every row equals the single-series path."""
    import torch
    import gadfly_amd
    from oracle import psd_ref
    flux = _series(50000, 9, r=5)
    dev = torch.as_tensor(flux, device="cuda")
    ps = gadfly_amd.PowerSpectrum.from_flux(dev, 60e-6)
    assert ps.power.shape == (5, 25000) and ps._power_dev.is_cuda
    b = ps.bin(20)
    assert b.power.shape == (5, 20) and b.error.shape == (5, 20)
    freq, power, _ = psd_ref.fft_power(flux, 60e-6)
    assert _relmax(ps.power, power) < 1e-13
    c, s, e = psd_ref.bin_power_ranges(freq, power, bins=20)
    np.testing.assert_allclose(b.power, s, rtol=RTOL)
    np.testing.assert_allclose(b.error, e, rtol=RTOL)
    cut = ps.cutout(3.0, 1000.0)
    keep = (freq >= 3.0) & (freq <= 1000.0)
    np.testing.assert_array_equal(cut.frequency, freq[keep])
    assert cut.power.shape == (5, keep.sum()) and cut.name == "Power spectrum (cutout)"


def test_sample_device_equals_sample(hip):
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters
    kernel = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(12), texp=60.0)
    t = np.arange(20000) * 60e-6
    gp = gadfly_amd.GaussianProcess(kernel, t=t, mean=3.0)
    for size in (None, 4):
        np.random.seed(3)
        host = gp.sample(size=size)
        np.random.seed(3)
        dev = gp.sample_device(size=size)
        assert dev.is_cuda and tuple(dev.shape) == host.shape
        np.testing.assert_allclose(dev.cpu().numpy(), host, rtol=0, atol=1e-9 * np.abs(host).max())


def test_round_trip_on_device(hip, n_trials=4, nbins=15):
    """The reference's hot-path test (gadfly/tests/test_core.py:19-51) with everything after the
    normal draws on the GPU: sample -> FFT power -> log bins, against kernel.get_psd within 5 sigma."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters
    np.random.seed(42)
    kernel = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(30), texp=60.0)
    t = np.arange(100_000) * 60e-6
    gp = gadfly_amd.GaussianProcess(kernel, t=t)
    draws = gp.sample_device(size=n_trials)
    ps = gadfly_amd.PowerSpectrum.from_flux(draws, 60e-6, name="draws").bin(nbins)
    model = kernel.get_psd(2 * np.pi * ps.frequency)
    ok = (ps.frequency < 1e3) & (ps.frequency > 3)
    for r in range(n_trials):
        dev = np.abs((model[ok] - ps.power[r][ok]) / np.nanmax(ps.error[r]))
        assert np.nanmax(dev) < 5


def test_sample_device_with_device_rng(hip):
    """``sample_device(rng="device")``: normal vectors drawn on the GPU (an opt-in extension: not the
    reference's random stream).  Reproducible for a seed, unit-variance inputs (the draws' sample variance
    matches the kernel's variance), and the default path still reproduces numpy's legacy stream."""
    import torch
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(6), texp=60.0)
    N = 20000
    gp = gadfly_amd.GaussianProcess(k, t=uniform_times(N, 60.0))
    a = gp.sample_device(size=16, rng="device", seed=7)
    b = gp.sample_device(size=16, rng="device", seed=7)
    c = gp.sample_device(size=16, rng="device", seed=8)
    assert a.is_cuda and a.shape == (16, N) and torch.equal(a, b) and not torch.equal(a, c)
    # variance ACROSS the 16 realisations at a fixed time (averaged over the times) = the kernel's variance
    # (a draw's own sample variance over a finite window is smaller: the granulation terms are red)
    var = float(a.var(dim=0).mean())
    k0 = float(k.get_value(np.zeros(1))[0])
    assert 0.8 * k0 < var < 1.2 * k0, (var, k0)
    one = gp.sample_device(rng="device", seed=1)
    assert one.shape == (N,) and abs(float(one.mean())) < 1e-9 * (1 + float(one.abs().max()))
    np.random.seed(3)
    ref = gp.sample(size=2)
    np.random.seed(3)
    dev = gp.sample_device(size=2).cpu().numpy()
    assert np.max(np.abs(dev - ref)) <= 1e-12 * np.max(np.abs(ref))
    with pytest.raises(ValueError):
        gp.sample_device(rng="philox")
