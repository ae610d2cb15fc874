"""
TEST INFRASTRUCTURE -- not part of the product path (only tests/ may import this).

CPU restatement (numpy) of ``interpolate_missing_data`` (/root/reference/gadfly/interp.py:6-60):
fill the missing cadences of an otherwise evenly sampled series by linear interpolation, the step
the reference runs before every FFT power spectrum (psd.py:495, :531).

Parity unpinned at the reference level: the reference has no test or fixture for this function;
the restatement follows its published steps (median spacing, rounded cadence indices, set
difference against the full index range, ``np.interp``, merge in time order) and is checked against
hand-built series in tests/test_interp_oracle.py.
"""
import numpy as np


def interpolate_missing_data(times, fluxes, cadences=None):
    times = np.asarray(times, dtype=np.float64)
    fluxes = np.asarray(fluxes, dtype=np.float64)
    t0 = times[0]
    if cadences is None:
        dt = np.median(np.diff(times))                              # interp.py:40
        index = np.rint((times - t0) / dt)                          # interp.py:43
    else:
        cadences = np.asarray(cadences)
        dt = np.median(np.diff(times) / np.diff(cadences))          # interp.py:36
        index = cadences - cadences[0]                              # interp.py:37
    every = np.arange(index.min(), index.max())                     # interp.py:46-47 (last one is present)
    missing = np.setdiff1d(every, index)                            # interp.py:48
    t_new = t0 + missing * dt                                       # interp.py:50
    f_new = np.interp(t_new, times, fluxes)                         # interp.py:53
    t_all = np.concatenate([times, t_new])
    f_all = np.concatenate([fluxes, f_new])
    order = np.argsort(t_all, kind="stable")                        # interp.py:56-59
    return t_all[order], f_all[order]


def stitch_quarters(quarters, detrend_poly_order=3, in_ppm=False):
    """Restatement of the multi-quarter preparation of /root/reference/gadfly/psd.py:483-531 (the FFT branch with
    ``detrend=True``): per quarter `interpolate_missing_data`, polynomial normalisation to ppm, stitch in the
    order given, `interpolate_missing_data` again over the stitched series.  lightkurve's ``remove_nans`` /
    ``remove_outliers`` / ``stitch`` are not restated (inputs are clean arrays; stitch(lambda x: x) is a
    concatenation).  Returns (t, flux_ppm, median spacing)."""
    ts, fs = [], []
    for t, f in quarters:
        t, f = interpolate_missing_data(np.asarray(t, float), np.asarray(f, float))
        if not in_ppm:
            fit = np.polyval(np.polyfit(t - t.mean(), f, detrend_poly_order), t - t.mean())
            normed_flux = f / fit
            median_flux = np.median(normed_flux)
            f = 1e6 * np.array(normed_flux / median_flux - 1)
        ts.append(t)
        fs.append(f)
    t = np.concatenate(ts)                          # (lightkurve's stitch keeps the collection's order)
    f = np.concatenate(fs)
    if len(ts) > 1:
        t, f = interpolate_missing_data(t, f)
    return t, f, float(np.median(np.diff(t)))
