"""
``interpolate_missing_data`` on the device (mirror of /root/reference/gadfly/interp.py:6-60):
fill the missing cadences of an otherwise evenly sampled light curve by linear interpolation --
what the reference does before every FFT power spectrum (psd.py:495, :531).
The cadence statistic (a median) is taken on the host from the arrays the caller hands over; the
gap search, the prefix sum over the gaps and the fill run on the GPU (``gf_interp_plan``,
``gf_interp_fill``) and reproduce numpy's results bit for bit.
"""
import numpy as np

from . import _lib

__all__ = ["interpolate_missing_data", "stitch_quarters"]


def interpolate_missing_data(times, fluxes, cadences=None, device=None, return_device=False):
    """
    Assuming ``times`` are uniformly spaced with missing cadences, fill in the missing cadences
    with linear interpolation; ``cadences`` (integer cadence numbers) can be passed if known.
    Returns (interpolated_times, interpolated_fluxes) as numpy arrays, or as float64 device
    tensors with ``return_device=True`` (e.g. for :meth:`gadfly_amd.PowerSpectrum.from_flux`).
    """
    import torch
    lib = _lib.load()
    times = np.ascontiguousarray(times, dtype=np.float64)
    fluxes = np.ascontiguousarray(fluxes, dtype=np.float64)
    if times.ndim != 1 or times.shape != fluxes.shape or times.size < 2:
        raise ValueError("dimension mismatch")
    if np.any(np.diff(times) <= 0):
        raise ValueError("times must be strictly increasing")
    if cadences is None:
        dt = float(np.median(np.diff(times)))                       # reference interp.py:40
        cad = None
    else:
        cad = np.ascontiguousarray(cadences, dtype=np.int64)
        if cad.shape != times.shape:
            raise ValueError("dimension mismatch")
        dt = float(np.median(np.diff(times) / np.diff(cad)))        # reference interp.py:36
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = times.size
    with torch.cuda.device(dev):
        t_d = torch.as_tensor(times, device=dev)
        f_d = torch.as_tensor(fluxes, device=dev)
        c_d = None if cad is None else torch.as_tensor(cad, device=dev)
        offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
        work = torch.empty(max(1, int(lib.gf_interp_work(n))), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        cp = None if c_d is None else _lib.ptr(c_d)
        _lib.check(lib.gf_interp_plan(n, _lib.ptr(t_d), cp, dt, _lib.ptr(offsets), _lib.ptr(work), st),
                   "gf_interp_plan")
        total = int(offsets[n].item())
        t_out = torch.empty(total, dtype=torch.float64, device=dev)
        f_out = torch.empty(total, dtype=torch.float64, device=dev)
        _lib.check(lib.gf_interp_fill(n, _lib.ptr(t_d), _lib.ptr(f_d), cp, dt, _lib.ptr(offsets),
                                      _lib.ptr(t_out), _lib.ptr(f_out), st), "gf_interp_fill")
        if return_device:
            return t_out, f_out
        return t_out.cpu().numpy(), f_out.cpu().numpy()


def stitch_quarters(quarters, detrend_poly_order=3, in_ppm=False, device=None):
    """
    Several observing quarters of one star -> ONE gap-filled series in ppm, the way the reference prepares a
    ``LightCurveCollection`` (/root/reference/gadfly/psd.py:483-531): per quarter fill the missing cadences
    (:func:`interpolate_missing_data`), divide by a polynomial of order ``detrend_poly_order`` in ``t - mean(t)`` and
    by the median of the result, ``1e6 (f / median - 1)`` (skipped when the fluxes are in ppm already,
    ``in_ppm=True``); concatenate the quarters in the order given, which must be chronological (lightkurve's
    ``stitch(lambda x: x)``); fill the gaps
    BETWEEN the quarters the same way.  ``quarters``: list of (times [d], fluxes) pairs, NaNs and outliers already
    removed (``remove_nans().remove_outliers()`` is lightkurve's, outside this path).  Returns
    ``(times, flux_ppm, cadence)`` with cadence = median spacing (psd.py:534).  The gap filling runs on the device;
    the O(N) polynomial fit on the host, as numpy does it in the reference.
    """
    ts, fs = [], []
    for t, f in quarters:
        t, f = interpolate_missing_data(t, f, device=device)
        if not in_ppm:
            x = t - t.mean()
            fit = np.polyval(np.polyfit(x, f, detrend_poly_order), x)
            normed = f / fit
            f = 1e6 * np.array(normed / np.median(normed) - 1)
        ts.append(t)
        fs.append(f)
    t = np.concatenate(ts)                          # (lightkurve's stitch keeps the collection's order)
    f = np.concatenate(fs)
    if len(ts) > 1:
        t, f = interpolate_missing_data(t, f, device=device)
    return t, f, float(np.median(np.diff(t)))
