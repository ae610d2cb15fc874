"""
GPU analogue of the reference's only hot-path test, ``test_ps_lc_round_trip``
(/root/reference/gadfly/tests/test_core.py:19-51): draws from the solar kernel must have a
binned FFT power spectrum that matches ``kernel.get_psd`` within 5 sigma between 3 and 1000 uHz.
The PSD estimate restates /root/reference/gadfly/psd.py:566-587 (FFT norm d / sqrt(2 pi) / N) and
psd.py:185-228 (trapezoid bin means, error = std / sqrt(n) * mean_x / span) in numpy.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _binned_psd(flux_ppm, d, nbins):
    N = len(flux_ppm)
    freq = np.fft.rfftfreq(N, d)[1:]                       # uHz when d is in 1/uHz
    fft = np.fft.rfft(flux_ppm)[1:]
    power = np.real(fft * np.conj(fft)) * d / (2 * np.pi) ** 0.5 / N
    x = np.log10(freq)
    edges = np.linspace(x.min(), x.max(), nbins + 1)
    idx = np.clip(np.digitize(x, edges) - 1, 0, nbins - 1)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    centers, mean, err = [], [], []
    for b in range(nbins):
        sel = np.flatnonzero(idx == b)
        centers.append(10 ** (0.5 * (edges[b] + edges[b + 1])))
        if len(sel) < 2 or x[sel[-1]] <= x[sel[0]]:
            mean.append(np.nan); err.append(np.nan)
            continue
        span = x[sel[-1]] - x[sel[0]]
        mean.append(trapz(power[sel], x[sel]) / span)
        err.append(np.std(power[sel]) / len(sel) ** 0.5 * np.mean(x[sel]) / span)
    return np.array(centers), np.array(mean), np.array(err)


def test_ps_lc_round_trip(hip, n_trials=10, nbins=15):
    import gadfly_amd
    np.random.seed(42)
    kernel = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
    assert len(kernel) == 172                              # general-width kernels (W > 63)
    t_days = np.linspace(0, 100, int(1e5))
    t = t_days * 86400e-6                                  # 1/uHz
    gp = gadfly_amd.GaussianProcess(kernel, t=t)
    d = np.median(np.diff(t))
    worst = 0.0
    for _ in range(n_trials):
        flux = gp.sample()
        f, p, e = _binned_psd(flux, d, nbins)
        model = kernel.get_psd(2 * np.pi * f)
        ok = (f < 1e3) & (f > 3)
        dev = np.abs((model[ok] - p[ok]) / np.nanmax(e))
        worst = max(worst, float(np.nanmax(dev)))
        assert np.nanmax(dev) < 5
    assert worst > 0


def test_round_trip_fast_path_kernel(hip):
    """Same statistic through the fused / time-parallel kernels (W = 60 synthetic solar-like)."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters
    np.random.seed(7)
    kernel = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(30), texp=60.0)
    t = np.arange(100_000) * 60e-6
    gp = gadfly_amd.GaussianProcess(kernel, t=t)
    assert gp._fast is not None
    d = 60e-6
    for _ in range(4):
        f, p, e = _binned_psd(gp.sample(), d, 15)
        model = kernel.get_psd(2 * np.pi * f)
        ok = (f < 1e3) & (f > 3)
        assert np.nanmax(np.abs((model[ok] - p[ok]) / np.nanmax(e))) < 5
