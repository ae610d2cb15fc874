"""
TEST INFRASTRUCTURE -- not part of the product path (only tests/ may import this).

CPU restatement (numpy / scipy) of the reference's power-spectrum estimate, the step after
``sample`` in its only hot-path test (/root/reference/gadfly/tests/test_core.py:29-34):

* ``fft_power``       -- ``PowerSpectrum._fft`` (/root/reference/gadfly/psd.py:566-587) and the
  zero-frequency drop of ``from_light_curve`` (psd.py:559-561), unit-free: times in 1/uHz,
  fluxes in ppm, frequencies in uHz, power in ppm^2/uHz;
* ``bin_power_lookup`` -- ``bin_power_spectrum`` (psd.py:229-300) the way the reference runs it:
  ``scipy.stats.binned_statistic`` with per-bin callables that find the bin's end points by
  VALUE in the full arrays (``spectral_binning`` psd.py:186-199, ``spectral_binning_err``
  psd.py:202-226);
* ``bin_power_ranges`` -- the same statistics over index ranges of the ascending axis (what the
  device kernel implements); ``bin_starts`` applies binned_statistic's edge rules (half-open
  bins, the right-most edge closed with scipy's rounding test).

Parity unpinned at the reference level: the reference holds no value-level fixture for its power
spectra (only the 5-sigma statistical round trip); the two formulations here pin each other.
"""
import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


def fft_power(flux_ppm, d, include_zero_freq=False):
    """(frequency [uHz], power [ppm^2/uHz], norm) of an evenly sampled series, ``d`` in 1/uHz."""
    flux_ppm = np.asarray(flux_ppm, dtype=np.float64)
    n = flux_ppm.shape[-1]
    freq = np.fft.rfftfreq(n, d)
    spec = np.fft.rfft(flux_ppm, axis=-1)
    norm = d / (2 * np.pi) ** 0.5 / n
    power = np.real(spec * np.conj(spec)) * norm
    if not include_zero_freq:
        freq, power = freq[1:], power[..., 1:]
    return freq, power, norm


def _axis(freq, log):
    return np.log10(freq) if log else np.asarray(freq, dtype=np.float64)


def _default_bins(axis, bins):
    return len(axis) // 10000 if bins is None else bins


def bin_power_lookup(freq, power, bins=None, log=True, constant=1):
    """Reference formulation (value look-ups inside scipy.stats.binned_statistic callables)."""
    from scipy.stats import binned_statistic
    axis = _axis(freq, log)
    bins = _default_bins(axis, bins)

    def ends(y):
        lo = np.argwhere(power == y[0])[0, 0]
        hi = np.argwhere(power == y[-1])[0, 0]
        return lo, hi

    def stat(y):
        if len(y) == 0:
            return np.nan
        lo, hi = ends(y)
        if hi > lo and axis[hi] - axis[lo] > 0:
            return _trapz(y, axis[lo:hi + 1]) / (axis[hi] - axis[lo])
        return y[0]

    def stat_err(y):
        if len(y) == 0:
            return np.nan
        lo, hi = ends(y)
        if hi > lo and axis[hi] - axis[lo] > 0:
            return (np.nanstd(y) / len(y) ** 0.5
                    * np.nanmean(axis[lo:hi + 1]) / (axis[hi] - axis[lo]) / constant)
        return y[0]

    bs = binned_statistic(axis, power, statistic=stat, bins=bins)
    bs_err = binned_statistic(axis, power, statistic=stat_err, bins=bins)
    mid = 0.5 * (bs.bin_edges[1:] + bs.bin_edges[:-1])
    return (10 ** mid if log else mid), bs.statistic, bs_err.statistic


def bin_starts(axis, bins):
    """(edges, start): bin b holds axis[start[b]:start[b+1]] (axis ascending).

    scipy.stats.binned_statistic: an integer ``bins`` means ``linspace(min, max, bins + 1)``;
    bins are half-open [edge_b, edge_b+1) except that points equal to the last edge (after
    rounding to ``int(-log10(min bin width)) + 6`` decimals) belong to the last bin.
    """
    axis = np.asarray(axis, dtype=np.float64)
    if np.ndim(bins) == 0:
        lo, hi = axis.min(), axis.max()
        if lo == hi:
            lo, hi = lo - 0.5, hi + 0.5
        edges = np.linspace(lo, hi, int(bins) + 1)
    else:
        edges = np.asarray(bins, dtype=np.float64)
    start = np.searchsorted(axis, edges, side="left").astype(np.int64)
    decimal = int(-np.log10(np.diff(edges).min())) + 6
    on_edge = (axis >= edges[-1]) & (np.around(axis, decimal) == np.around(edges[-1], decimal))
    start[-1] += int(np.count_nonzero(on_edge))
    return edges, start


def bin_power_ranges(freq, power, bins=None, log=True, constant=1):
    """Index-range formulation (the device kernel's); ``power`` may be (M,) or (R, M)."""
    axis = _axis(freq, log)
    bins = _default_bins(axis, bins)
    edges, start = bin_starts(axis, bins)
    power = np.asarray(power, dtype=np.float64)
    p2 = np.atleast_2d(power)
    nb = len(edges) - 1
    stat = np.full((p2.shape[0], nb), np.nan)
    err = np.full((p2.shape[0], nb), np.nan)
    for b in range(nb):
        s, e = start[b], start[b + 1]
        if e <= s:
            continue
        x = axis[s:e]
        span = x[-1] - x[0]
        for r in range(p2.shape[0]):
            y = p2[r, s:e]
            if e - s == 1 or not span > 0:
                stat[r, b] = err[r, b] = y[0]
                continue
            stat[r, b] = _trapz(y, x) / span
            err[r, b] = np.std(y) / len(y) ** 0.5 * np.mean(x) / span / constant
    mid = 0.5 * (edges[1:] + edges[:-1])
    centers = 10 ** mid if log else mid
    if power.ndim == 1:
        return centers, stat[0], err[0]
    return centers, stat, err
