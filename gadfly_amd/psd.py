"""
Observed power spectra on the device: ``PowerSpectrum`` (mirror of
/root/reference/gadfly/psd.py:364-650 for the FFT estimate and its binning).

This is the step *after* ``GaussianProcess.sample`` in the reference's own hot-path test
(/root/reference/gadfly/tests/test_core.py:29-34: ``PowerSpectrum(...).bin(...)`` of a draw is
compared with ``kernel.get_psd``) and SURVEY.md 8f rank 3.  The transform is hipFFT's (through
``torch.fft.rfft`` on the device tensor), ``gf_psd_power`` forms the normalised power
(psd.py:566-587) and ``gf_psd_bin`` the two binned statistics of ``bin_power_spectrum``
(psd.py:186-300) for all series of a batch in one launch.  Frequencies are in uHz, power in
ppm^2/uHz, sampling intervals in 1/uHz (= 1e6 s) -- gadfly's native units; astropy Quantities are
accepted where astropy is installed.  Plotting and the Lomb-Scargle estimate (astropy.timeseries)
stay with the reference.
"""
import numpy as np

from . import _lib
from . import units as _units

__all__ = ["PowerSpectrum", "bin_power_spectrum"]


def _value(x, unit_name):
    """Strip an astropy unit (converted to gadfly's native one) if there is one."""
    if _units.has_unit(x):
        _units.require_astropy("a Quantity argument")
        u = _units.u
        target = {"uHz": u.uHz, "psd": u.cds.ppm ** 2 / u.uHz, "1/uHz": 1 / u.uHz,
                  "ppm": u.cds.ppm}[unit_name]
        return np.asarray(x.to(target).value, dtype=np.float64)
    return x


def _bin_starts(axis, bins):
    """Bin edges and index ranges on an ascending axis, with scipy.stats.binned_statistic's rules
    (what /root/reference/gadfly/psd.py:257-274 relies on): an integer means
    ``linspace(min, max, bins + 1)``; bins are half-open, the last edge is closed (scipy's rounding
    test).  Returns (edges, start) with bin b = ``axis[start[b]:start[b+1]]``."""
    if np.ndim(bins) == 0:
        if int(bins) < 1:
            raise ValueError("`bins` must be a positive integer or an array of edges")
        lo, hi = float(axis.min()), float(axis.max())
        if lo == hi:
            lo, hi = lo - 0.5, hi + 0.5
        edges = np.linspace(lo, hi, int(bins) + 1)
    else:
        edges = np.asarray(bins, dtype=np.float64)
        if edges.ndim != 1 or len(edges) < 2 or np.any(np.diff(edges) < 0):
            raise ValueError("bin edges must be a monotonically increasing 1-D array")
    widths = np.diff(edges)
    if widths.min() == 0:
        raise ValueError("The smallest edge difference is numerically 0.")
    start = np.searchsorted(axis, edges, side="left").astype(np.int64)
    decimal = int(-np.log10(widths.min())) + 6
    on_edge = (axis >= edges[-1]) & (np.around(axis, decimal) == np.around(edges[-1], decimal))
    start[-1] += int(np.count_nonzero(on_edge))
    return edges, start


class PowerSpectrum:
    """
    An observed power spectrum (reference psd.py:364-396): ``frequency`` [uHz], ``power``
    [ppm^2/uHz] (shape (M,), or (R, M) for a batch of R series sharing the frequency axis),
    optional ``error``, ``name``, ``norm`` and ``detrended_lc``.
    """

    def __init__(self, frequency, power, error=None, name=None, norm=None, detrended_lc=None):
        self.frequency = np.asarray(_value(frequency, "uHz"), dtype=np.float64)
        self.power = _value(power, "psd")
        self.error = None if error is None else _value(error, "psd")
        self.name = name
        self.norm = norm
        self.detrended_lc = detrended_lc
        self._power_dev = None              # device copy of `power` when it was made there

    @property
    def omega(self):
        """Angular frequency 2 pi f, f in uHz (reference psd.py:397-409)."""
        return 2 * np.pi * self.frequency

    @property
    def light_curve_rms(self):
        """Approximate rms of the light curve [ppm] (reference psd.py:411-421)."""
        return (np.asarray(self.power) * self.norm) ** 0.5

    def bin(self, bins=None, **kwargs):
        """Binned power spectrum (reference psd.py:423-441)."""
        return bin_power_spectrum(self, bins, **kwargs)

    def cutout(self, frequency_min=None, frequency_max=None):
        """Measurements with frequency_min <= f <= frequency_max (reference psd.py:611-650)."""
        lo = 0.0 if frequency_min is None else float(_value(frequency_min, "uHz"))
        hi = np.inf if frequency_max is None else float(_value(frequency_max, "uHz"))
        keep = (self.frequency <= hi) & (self.frequency >= lo)
        name = (self.name if self.name is not None else "Power spectrum") + " (cutout)"
        args = []
        if self.error is not None:
            args.append(np.asarray(self.error)[..., keep])
        return PowerSpectrum(self.frequency[keep], np.asarray(self.power)[..., keep], *args,
                             name=name, norm=self.norm)

    def plot(self, **kwargs):
        raise NotImplementedError(
            "plotting is outside the GP hot path (gadfly/psd.py:36-183 is left as-is)")

    # ------------------------------------------------------------------------------------
    @classmethod
    def from_flux(cls, flux, d, include_zero_freq=False, name=None, device=None):
        """FFT power spectrum of evenly sampled fluxes [ppm], sampling interval ``d`` [1/uHz].

        ``flux`` is (N,) or (R, N): a numpy array, or a float64 tensor already on the device
        (e.g. draws that never left it).  Same estimate as ``PowerSpectrum._fft``
        (reference psd.py:566-587) with the zero frequency dropped unless asked for
        (psd.py:559-561); everything after the upload runs on the GPU.
        """
        import torch
        lib = _lib.load()
        d = float(_value(d, "1/uHz"))
        if torch.is_tensor(flux):
            x = flux.to(dtype=torch.float64)
            if not x.is_cuda:
                x = x.to(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
        else:
            flux = np.ascontiguousarray(_value(flux, "ppm"), dtype=np.float64)
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            x = torch.as_tensor(flux, device=dev)
        if x.ndim not in (1, 2) or x.shape[-1] < 2:
            raise ValueError("flux must have shape (N,) or (R, N) with N >= 2")
        single = x.ndim == 1
        x2 = x.reshape(1, -1) if single else x.contiguous()
        R, N = x2.shape
        with torch.cuda.device(x2.device):
            spec = torch.view_as_real(torch.fft.rfft(x2, dim=-1)).contiguous()     # (R, M, 2)
            M = spec.shape[1]
            first = 0 if include_zero_freq else 1
            norm = d / (2 * np.pi) ** 0.5 / N
            power = torch.empty((R, M - first), dtype=torch.float64, device=x2.device)
            st = torch.cuda.current_stream(x2.device).cuda_stream
            _lib.check(lib.gf_psd_power(R, M, first, norm, _lib.ptr(spec), _lib.ptr(power), st),
                       "gf_psd_power")
        freq = np.fft.rfftfreq(N, d)[first:]
        host = power.cpu().numpy()
        ps = cls(freq, host[0] if single else host, name=name, norm=norm)
        ps._power_dev = power
        return ps

    @classmethod
    def from_light_curve(cls, light_curve, method="fft", include_zero_freq=False, name=None,
                         detrend=False, **kwargs):
        """Power spectrum of a light-curve-like object (``.time`` [days or Time], ``.flux`` [ppm]).

        Only the plain FFT estimate of an evenly sampled, already normalised light curve runs
        here (reference psd.py:537-563, ``detrend=False`` branch); detrending / gap interpolation
        (psd.py:474-535, lightkurve) and Lomb-Scargle (psd.py:589-601, astropy) are the
        reference's.
        """
        if method.lower() not in ("fft", "lomb-scargle"):
            raise ValueError(f'PowerSpectrum.from_lightcurve was given method="{method}", but it '
                             "must be one of: ['fft', 'lomb-scargle'].")
        if method.lower() != "fft" or detrend:
            raise NotImplementedError(
                "only method='fft' with detrend=False runs on the device; detrending and "
                "Lomb-Scargle need lightkurve / astropy (reference psd.py:474-535, :589-601)")
        time = light_curve.time
        jd = np.asarray(getattr(time, "jd", time), dtype=np.float64)
        d = np.median(np.diff(jd)) * 86400.0 / _units.SECONDS_PER_INVERSE_UHZ
        meta = getattr(light_curve, "meta", None) or {}
        return cls.from_flux(light_curve.flux, d, include_zero_freq=include_zero_freq,
                             name=meta.get("name", name))


def bin_power_spectrum(power_spectrum, bins=None, log=True, constant=1, device=None):
    """
    Bin a power spectrum into (by default log-spaced) frequency bins
    (reference psd.py:229-297): per bin the trapezoid mean of the power over the bin's span and
    the error estimate std / sqrt(n) * mean_x / span / constant (psd.py:186-227), for every
    series of the batch in one ``gf_psd_bin`` launch.  Returns a new :class:`PowerSpectrum`
    at the bin centres.
    """
    import torch
    lib = _lib.load()
    freq = np.asarray(power_spectrum.frequency, dtype=np.float64)
    if np.any(np.diff(freq) < 0):
        raise ValueError("the frequencies of the power spectrum must be sorted")
    axis = np.log10(freq) if log else freq
    if bins is None:
        bins = len(axis) // 10000
    edges, start = _bin_starts(axis, bins)
    nb = len(edges) - 1

    power = power_spectrum._power_dev
    if power is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        power = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(power_spectrum.power),
                                                     dtype=np.float64), device=dev)
    R, M = power.shape
    if M != len(axis):
        raise ValueError("dimension mismatch")
    with torch.cuda.device(power.device):
        x_d = torch.as_tensor(axis, device=power.device)
        s_d = torch.as_tensor(start, device=power.device)
        stat = torch.empty((R, nb), dtype=torch.float64, device=power.device)
        err = torch.empty_like(stat)
        st = torch.cuda.current_stream(power.device).cuda_stream
        _lib.check(lib.gf_psd_bin(R, M, nb, _lib.ptr(x_d), _lib.ptr(power), _lib.ptr(s_d),
                                  float(constant), _lib.ptr(stat), _lib.ptr(err), st), "gf_psd_bin")
        stat_h, err_h = stat.cpu().numpy(), err.cpu().numpy()
    mid = 0.5 * (edges[1:] + edges[:-1])
    centers = 10 ** mid if log else mid
    single = np.ndim(power_spectrum.power) == 1
    name = (power_spectrum.name if power_spectrum.name is not None else "Power spectrum") + " (binned)"
    return PowerSpectrum(centers, stat_h[0] if single else stat_h,
                         err_h[0] if single else err_h, name=name)
