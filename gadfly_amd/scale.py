"""
Asteroseismic scaling relations (host-side, run once per star) -- the step *before* the GP hot
path.  Unit-free port of the relations that
``Hyperparameters.for_star`` needs from /root/reference/gadfly/scale.py: every function takes plain
floats in solar units (mass [M_sun], radius [R_sun], luminosity [L_sun]), temperatures in K,
frequencies in uHz, wavelengths in micron.  astropy Quantities are accepted when astropy is
installed (they are converted on entry).

Differences from the reference (documented, not hidden):
* ``astropy.modeling.Voigt1D`` -> ``scipy.special.wofz`` with the same parameterisation;
* ``astropy.modeling.BlackBody`` -> the Planck function written out (only ratios are used);
* named ``tynt`` filters need ``tynt`` (not installable offline): without it only the bolometric
  "SOHO VIRGO" pseudo-filter and user-supplied (wavelength, transmittance) curves work.
"""
import json
import os
import warnings

import numpy as np
from scipy.special import wofz

from . import units as _units

__all__ = [
    "p_mode_amplitudes", "delta_nu", "nu_max", "tau_eff", "tau_gran",
    "granulation_amplitude", "c_K", "amplitude_with_wavelength", "p_mode_intensity",
    "Filter", "broomhall_p_mode_freqs", "hyperparameters_for_star",
]

# Solar parameters (reference scale.py:20-36)
_solar_temperature = 5777.0          # K
_solar_mass = 1.0                    # M_sun
_solar_radius = 1.0                  # R_sun
_solar_luminosity = 1.0              # L_sun
_solar_nu_max = 3090.0               # uHz, Huber et al. (2011)
_solar_delta_nu = 135.1              # uHz

# Huber et al. (2011) amplitude relation exponents (reference scale.py:38-42)
_huber_r, _huber_s, _huber_t = 2, 0.886, 1.89

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _val(x, unit=None):
    """float (or ndarray) from a number or an astropy Quantity."""
    if _units.has_unit(x):
        _units.require_astropy("Quantity arguments")
        return np.asarray(x.to(unit).value if unit is not None else x.value, dtype=float)
    return np.asarray(x, dtype=float)


def _solar(x, kind):
    if not _units.has_unit(x):
        return float(x)
    u = _units.u
    return float(x.to({"mass": u.M_sun, "radius": u.R_sun, "lum": u.L_sun, "temp": u.K}[kind]).value)


def c_K(temperature):
    """Bolometric correction factor (reference scale.py:50-73)."""
    return float((_solar(temperature, "temp") / 5934.0) ** 0.8)


def _amplitudes_huber(mass, temperature, luminosity):
    return luminosity ** _huber_s / (mass ** _huber_t * temperature ** (_huber_r - 1)
                                     * c_K(temperature))


def p_mode_amplitudes(mass, temperature, luminosity):
    """p-mode amplitude scaling, Huber et al. (2011) (reference scale.py:83-107)."""
    m, T, L = _solar(mass, "mass"), _solar(temperature, "temp"), _solar(luminosity, "lum")
    return float(_amplitudes_huber(m, T, L)
                 / _amplitudes_huber(_solar_mass, _solar_temperature, _solar_luminosity))


def delta_nu(mass, radius):
    """Large frequency separation scaling (reference scale.py:176-198)."""
    return float(_solar(mass, "mass") ** 0.5 * _solar(radius, "radius") ** (-3 / 2))


def nu_max(mass, temperature, radius):
    """Frequency of maximum power scaling (reference scale.py:201-225)."""
    return float(_solar(mass, "mass") * _solar(radius, "radius") ** -2
                 * (_solar(temperature, "temp") / _solar_temperature) ** -0.5)


def tau_eff(nu_max_uHz):
    """Granulation time scale vs nu_max, Kallinger et al. (2014) (reference scale.py:229-250)."""
    return float((float(_val(nu_max_uHz, None if not _units.has_unit(nu_max_uHz) else _units.u.uHz))
                  / _solar_nu_max) ** -0.89)


def _tau_gran(mass, temperature, luminosity):
    return luminosity / (mass * temperature ** 3.5)


def tau_gran(mass, temperature, luminosity):
    """Granulation time scale, Kjeldsen & Bedding (2011) (reference scale.py:382-406)."""
    m, T, L = _solar(mass, "mass"), _solar(temperature, "temp"), _solar(luminosity, "lum")
    return float(_tau_gran(m, T, L)
                 / _tau_gran(_solar_mass, _solar_temperature, _solar_luminosity))


def _granulation_power_factor(mass, temperature, luminosity):
    return luminosity ** 2 / (mass ** 3 * temperature ** 5.5)


def granulation_amplitude(mass, temperature, luminosity):
    """Granulation amplitude scaling, Kjeldsen & Bedding (2011) eq. 24 (reference scale.py:458-484)."""
    m, T, L = _solar(mass, "mass"), _solar(temperature, "temp"), _solar(luminosity, "lum")
    return float(_granulation_power_factor(m, T, L)
                 / _granulation_power_factor(_solar_mass, _solar_temperature, _solar_luminosity))


def _voigt1d(x, x_0, amplitude_L, fwhm_L, fwhm_G):
    """astropy.modeling.models.Voigt1D.evaluate, written with the Faddeeva function."""
    sqrt_ln2 = np.sqrt(np.log(2.0))
    z = (2.0 * (np.asarray(x, dtype=float) - x_0) + 1j * fwhm_L) * sqrt_ln2 / fwhm_G
    return wofz(z).real * np.sqrt(np.log(2.0) * np.pi) * amplitude_L * fwhm_L / fwhm_G


def _v_osc_kiefer_scaled(freq, nu_max_uHz, delta_nu_uHz):
    """Velocity power envelope of Kiefer et al. (2018), stretched by delta_nu
    (reference scale.py:515-539); freq, nu_max, delta_nu in uHz; result in m^2/s^2 (x uHz/Hz)."""
    stretch = _solar_delta_nu / delta_nu_uHz
    sigma = 181.8 / stretch           # stddev of Gaussian
    gamma = 150.9 / stretch           # HWHM of Lorentzian
    Sigma = 611.8 / stretch           # FWHM of Voigt
    S = -0.1                          # asymmetry parameter
    a = 3299 * 1e4                    # height factor   [m^2/s^2/Hz]
    b = -581.0                        # offset factor   [m^2/s^2/Hz]
    freq = np.asarray(freq, dtype=float)
    A = 1 / np.pi * (np.arctan(S * (freq - nu_max_uHz) / Sigma) + 0.5)
    voigt = _voigt1d(freq, nu_max_uHz, a, 2 * gamma, 2.355 * sigma)
    return A * (b + voigt) * 1e-6     # (m^2/s^2/Hz) * uHz


def p_mode_intensity(temperature, freq, nu_max_uHz, delta_nu_uHz, wavelength=None):
    """p-mode envelope relative to its value at nu_max (reference scale.py:591-632).  The
    velocity->intensity conversion (wavelength, temperature) cancels in the ratio."""
    return np.asarray(_v_osc_kiefer_scaled(freq, nu_max_uHz, delta_nu_uHz)
                      / _v_osc_kiefer_scaled(nu_max_uHz, nu_max_uHz, delta_nu_uHz))


def _planck_nu(wavelength_um, temperature):
    """B_nu(T) at the given wavelengths, arbitrary normalisation (only ratios are used)."""
    h, c, k = 6.62607015e-34, 2.99792458e8, 1.380649e-23
    nu = c / (np.asarray(wavelength_um, dtype=float) * 1e-6)
    with np.errstate(over="ignore"):
        return 2 * h * nu ** 3 / c ** 2 / np.expm1(h * nu / (k * temperature))


class Filter:
    """
    Photometric bandpass (reference core.py:547-621, a ``tynt.Filter`` subclass there).
    ``Filter("SOHO VIRGO")`` is the bolometric pseudo-filter; ``Filter((wavelength_um,
    transmittance))`` wraps a user curve; any other name is looked up in ``tynt`` if installed.
    """
    default_filter = "Kepler/Kepler.K"

    def __init__(self, identifier_or_filter, download=False):
        if identifier_or_filter is None:
            _units.warn(
                "An observing bandpass is required to construct the kernel. gadfly "
                f'will assume the default filter "{self.default_filter}". To prevent '
                "this warning, supply the Hyperparameters with the `bandpass` "
                "keyword argument.")
            identifier_or_filter = self.default_filter
        if isinstance(identifier_or_filter, str) and identifier_or_filter.upper() == "SOHO VIRGO":
            self.wavelength = np.logspace(-1.5, 1.5, 1000)          # micron
            self.transmittance = np.ones_like(self.wavelength)
        elif hasattr(identifier_or_filter, "wavelength") and hasattr(identifier_or_filter, "transmittance"):
            self.wavelength = _val(identifier_or_filter.wavelength,
                                   _units.u.um if _units.HAS_ASTROPY else None)
            self.transmittance = np.asarray(identifier_or_filter.transmittance, dtype=float)
        elif isinstance(identifier_or_filter, (tuple, list)) and len(identifier_or_filter) == 2:
            self.wavelength = np.asarray(identifier_or_filter[0], dtype=float)
            self.transmittance = np.asarray(identifier_or_filter[1], dtype=float)
        else:
            try:
                import tynt
            except ImportError as err:
                raise ValueError(
                    f'The observing bandpass "{identifier_or_filter}" needs the `tynt` package, '
                    'which is not installed; use bandpass="SOHO VIRGO" (bolometric) or pass a '
                    "(wavelength_um, transmittance) pair.") from err
            gen = tynt.FilterGenerator()
            if identifier_or_filter in gen.available_filters() and not download:
                filt = gen.reconstruct(identifier_or_filter)
            elif not download:
                raise ValueError(
                    f'The observing bandpass "{identifier_or_filter}" is not recognized in the '
                    "pre-loaded bandpasses in tynt, and the `download` keyword is "
                    f'"{download}".')
            else:
                filt = gen.download_true_transmittance(identifier_or_filter)
            self.wavelength = _val(filt.wavelength, _units.u.um)
            self.transmittance = np.asarray(filt.transmittance, dtype=float)

    @property
    def mean_wavelength(self):
        """Transmittance-weighted mean wavelength [micron]; None for the bolometric filter."""
        if np.all(self.transmittance == 1):
            return None
        return float(np.average(self.wavelength, weights=self.transmittance))


def amplitude_with_wavelength(filter, temperature, n_wavelengths=10_000, **kwargs):
    """Scale factor alpha of p-mode / granulation amplitudes with the bandpass, Morris et al.
    (2020) eq. 11 (reference scale.py:635-729).  Exactly 1 for the bolometric filter."""
    T = _solar(temperature, "temp")
    selected = filter if isinstance(filter, Filter) else Filter(filter, **kwargs)
    wl = np.logspace(-1.5, 1.5, n_wavelengths)                      # micron
    dT = 20.0
    I_nu = _planck_nu(wl, T)
    dI_dT = (_planck_nu(wl, T + 10.0) - _planck_nu(wl, T - 10.0)) / dT
    filt0 = np.ones_like(wl)
    filt1 = np.interp(wl, selected.wavelength, selected.transmittance, left=0, right=0)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    ratio_0 = trapz(dI_dT * wl * filt1, wl) / trapz(dI_dT * wl * filt0, wl)
    ratio_1 = trapz(I_nu * wl * filt0, wl) / trapz(I_nu * wl * filt1, wl)
    return float(ratio_0 * ratio_1)


def broomhall_p_mode_freqs(path=None):
    """(nu [uHz], degree) of Broomhall et al. (2009) table 2 (reference sun.py:22-33)."""
    from .synth import broomhall_modes
    return broomhall_modes(path)


def _p_mode_fit_to_sho_hyperparams(p_mode_parameters):
    """Per-degree (S0, Q) -> per-mode (S0, w0, Q) (reference sun.py:36-62)."""
    S0_l = np.asarray(p_mode_parameters[:4], dtype=float)
    Q_l = np.asarray(p_mode_parameters[4:], dtype=float)
    freq, ell = broomhall_p_mode_freqs()
    return np.vstack([S0_l[ell], 2 * np.pi * freq, Q_l[ell]]), ell


def _sho_psd(omega, S0, w0, Q):
    return (np.sqrt(2 / np.pi) * S0 * w0 ** 4
            / ((omega ** 2 - w0 ** 2) ** 2 + (omega ** 2 * w0 ** 2 / Q ** 2)))


def hyperparameters_for_star(cls, mass, radius, temperature, luminosity, bandpass=None,
                             name=None, quiet=False, magnitude=None):
    """
    Scale the SOHO VIRGO/PMO6 solar hyperparameters to a star
    (reference core.py:107-333, same sequence of relations, floats instead of Quantities).
    """
    m, R = _solar(mass, "mass"), _solar(radius, "radius")
    T, L = _solar(temperature, "temp"), _solar(luminosity, "lum")
    with open(os.path.join(_DATA, "hyperparameters.json")) as fh:
        hyperparameters = json.load(fh)

    granulation_hyperparams = [item for item in hyperparameters
                               if item["metadata"]["source"] == "granulation"]
    p_mode_hyperparams = [item for item in sorted(hyperparameters,
                                                  key=lambda x: x["metadata"].get("degree", -1))
                          if item["metadata"]["source"] == "oscillation"]
    p_mode_vector = np.transpose(
        [[ps["hyperparameters"].get(par) for par in ["S0", "Q"]] for ps in p_mode_hyperparams]
    ).ravel()
    (S0_fit, solar_w0, Q_fit), ell_labels = _p_mode_fit_to_sho_hyperparams(p_mode_vector)

    solar_gran_S0, solar_gran_w0, solar_gran_Q = np.transpose(
        [[ps["hyperparameters"].get(par) for par in ["S0", "w0", "Q"]]
         for ps in granulation_hyperparams])

    scaled_nu_max = _solar_nu_max * nu_max(m, T, R)
    filt = Filter(bandpass)
    amp_with_wavelength = amplitude_with_wavelength(filt, T)
    granulation_amp = granulation_amplitude(m, T, L)
    granulation_timescale = tau_gran(m, T, L)

    scaled = []
    for item in granulation_hyperparams:
        params = item["hyperparameters"]
        scale_S0 = params["S0"] * granulation_amp * amp_with_wavelength
        scaled_w0 = params["w0"] / granulation_timescale
        if scaled_w0 > 0:
            scaled.append(dict(hyperparameters=dict(S0=scale_S0, w0=scaled_w0, Q=params["Q"]),
                               metadata=item["metadata"]))
        elif not quiet:
            _units.warn(
                "The scaled solar hyperparameter with frequency "
                f"w0(old)={params['w0']:.0f} is being scaled to "
                f"w0(new)={scaled_w0:.0f}, which is not positive. "
                "This kernel term will be omitted.")

    solar_nu = solar_w0 / (2 * np.pi)                               # uHz
    granulation_background_solar = _sho_psd(
        2 * np.pi * solar_nu[:, None], solar_gran_S0[None, :], solar_gran_w0[None, :],
        solar_gran_Q[None, :]) * amp_with_wavelength

    scale_delta_nu = delta_nu(m, R)
    scaled_nu = scaled_nu_max + (solar_nu - _solar_nu_max) * scale_delta_nu
    scaled_w0 = 2 * np.pi * scaled_nu

    only_positive_omega = scaled_w0 > 0
    solar_nu = solar_nu[only_positive_omega]
    S0_fit = S0_fit[only_positive_omega]
    Q_fit = Q_fit[only_positive_omega]
    scaled_nu = scaled_nu[only_positive_omega]
    scaled_w0 = scaled_w0[only_positive_omega]

    wavelength = filt.mean_wavelength if filt.mean_wavelength is not None else 0.55
    p_mode_scale_factor = (
        p_mode_intensity(T, scaled_nu, scaled_nu_max, _solar_delta_nu * scale_delta_nu, wavelength)
        * p_mode_amplitudes(m, T, L))

    scaled_Gamma = 1.02 * np.exp((T - _solar_temperature) / 436.0)  # uHz
    solar_Gamma = solar_nu / Q_fit / 2                              # uHz
    scaled_Q = Q_fit * scaled_Gamma / solar_Gamma

    solar_psd_at_p_mode_peaks = _sho_psd(2 * np.pi * solar_nu, S0_fit,
                                         solar_w0[only_positive_omega], Q_fit)
    # Chaplin et al. (2008) eq. 3
    A = 2 * np.sqrt(4 * np.pi * solar_nu * solar_psd_at_p_mode_peaks)
    unscaled_height = 2 * A ** 2 / (np.pi * solar_Gamma)
    scaled_height = unscaled_height * p_mode_scale_factor
    scaled_A = np.sqrt(np.pi * scaled_Gamma * scaled_height / 2)
    scaled_psd_at_p_mode_peaks = (scaled_A / 2) ** 2 / (4 * np.pi * scaled_nu)

    scaled_S0 = (0.5 * (np.pi / 2) ** 0.5 * scaled_psd_at_p_mode_peaks / scaled_Q ** 2
                 * granulation_background_solar.sum(1)[only_positive_omega])

    for S0, w0, Q, degree in zip(scaled_S0, scaled_w0, np.ravel(scaled_Q), ell_labels):
        if np.all(np.array([S0, w0]) > 0):
            scaled.append(dict(
                hyperparameters=dict(S0=float(S0), w0=float(w0), Q=float(Q)),
                metadata=dict(source="oscillation", scaled=True, degree=int(degree))))
    return cls(scaled, name, magnitude)
