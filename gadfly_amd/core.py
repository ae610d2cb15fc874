"""
Kernel construction: ``Hyperparameters``, ``StellarOscillatorKernel``,
``SolarOscillatorKernel``, ``ShotNoiseKernel``.

Mirrors the public surface of /root/reference/gadfly/core.py (``__all__`` at
core.py:19-25) for the GP hot path: same constructor signatures, attribute names
(``kernel.term.terms``, ``kernel.delta``, ``kernel.name``,
``kernel.hyperparameters``), ``__add__`` name bookkeeping (core.py:405-427) and
the default-exposure warning (core.py:381-390).  The celerite term algebra the
reference inherits from ``celerite2.terms`` lives in :mod:`gadfly_amd.terms`.

The asteroseismic scaling relations behind ``Hyperparameters.for_star`` (reference
core.py:107-333 + scale.py) live in :mod:`gadfly_amd.scale` as a unit-free port
(scipy instead of astropy.modeling); named ``tynt`` bandpasses need ``tynt``, the
bolometric "SOHO VIRGO" pseudo-filter and user-supplied curves do not.  The PSD
plotting helpers are not part of the path (gadfly/psd.py is left as-is).
"""
import json
import os

import numpy as np

from . import terms as _terms
from . import units as _units
from .scale import Filter  # noqa: F401  (reference core.py exports Filter)

__all__ = [
    "Hyperparameters",
    "StellarOscillatorKernel",
    "SolarOscillatorKernel",
    "ShotNoiseKernel",
    "Filter",
]

dirname = os.path.dirname(os.path.abspath(__file__))
default_hyperparameter_path = os.path.join(
    dirname, "data", "hyperparameters.json"
)


def _sho_psd(omega, S0, w0, Q):
    """celerite2's SHO PSD (reference core.py:33-41)."""
    return (
        np.sqrt(2 / np.pi) * S0 * w0 ** 4
        / ((omega ** 2 - w0 ** 2) ** 2 + (omega ** 2 * w0 ** 2 / Q ** 2))
    )


class Hyperparameters(list):
    """
    List of ``{"hyperparameters": {S0, w0, Q}, "metadata": {...}}`` dicts
    (reference core.py:44-105).
    """

    def __init__(self, hyperparameters, name=None, magnitude=None):
        super().__init__(hyperparameters)
        self.name = name
        self.magnitude = magnitude

    def __repr__(self):
        first = json.dumps(self[0], indent=4)
        return (
            f"<{self.__class__.__name__} "
            + (f'"{self.name}" ' if self.name is not None else "")
            + f"(showing 1 of {len(self)}):\n[{first}...]>"
        )

    @staticmethod
    def _load_from_json(path):
        with open(path, "r") as param_file:
            return json.load(param_file)

    @classmethod
    def from_soho_virgo(cls, path=None, name="SOHO VIRGO/PMO6"):
        """Raw SOHO VIRGO/PMO6 fit (reference core.py:81-105).

        As in the reference, the oscillation entries of this file carry no
        ``w0`` and cannot be fed to ``StellarOscillatorKernel`` directly.
        """
        if path is None:
            path = default_hyperparameter_path
        return cls(cls._load_from_json(path), name=name)

    @classmethod
    def for_star(cls, mass, radius, temperature, luminosity,
                 bandpass=None, name=None, quiet=False, magnitude=None):
        """Asteroseismic scaling of the solar fit (reference core.py:107-333).

        Host-side, run once per star, outside the GP hot path: see
        :func:`gadfly_amd.scale.hyperparameters_for_star` (masses, radii, luminosities in
        solar units, temperature in K, or astropy Quantities).
        """
        from . import scale as _scale
        return _scale.hyperparameters_for_star(
            cls, mass, radius, temperature, luminosity, bandpass=bandpass,
            name=name, quiet=quiet, magnitude=magnitude)


class StellarOscillatorKernel(_terms.TermConvolution):
    """
    Sum of SHO kernels integrated over the exposure time
    (reference core.py:336-427): ``TermConvolution(TermSum(SHOTerm x J), delta)``.
    """

    def __init__(self, hyperparameters=None, texp=None, delta=None, name=None,
                 terms=None):
        kernel_components = []

        if hyperparameters is not None:
            self.hyperparameters = hyperparameters
            if name is None and getattr(hyperparameters, "name", None) is not None:
                name = hyperparameters.name
            kernel_components += [
                _terms.SHOTerm(**p["hyperparameters"])
                for p in self.hyperparameters
            ]

        if terms is not None:
            kernel_components += terms

        self.name = name
        term_sum = _terms.TermSum(*kernel_components)

        if delta is None:
            if texp is None:
                default_exp_s = 60.0
                _units.warn(
                    "An exposure time is required to construct the kernel. gadfly "
                    "will assume a default exposure time of 1.0 min. To prevent "
                    "this warning, supply the kernel with the `texp` keyword "
                    "argument."
                )
                texp = default_exp_s
            delta = _units.exposure_to_delta(texp)

        super().__init__(term_sum, delta)

    def plot(self, **kwargs):
        """PSD plotting lives in the reference's psd.py, which is left as-is."""
        raise NotImplementedError(
            "plotting is outside the GP hot path (gadfly/psd.py is left as-is); "
            "use kernel.get_psd(omega) with your own plotting code"
        )

    @classmethod
    def _from_terms(cls, terms, delta=None, name=None):
        return cls(terms=terms, delta=delta, name=name)

    def __add__(self, other):
        """Assumes ``other`` is an SHO term (``ShotNoiseKernel`` in the reference's use).

        A *list* is not accepted, as in the reference: its list branch reads ``other.terms``
        (/root/reference/gadfly/core.py:412), which a list does not have, so ``kernel + [term, ...]`` raises
        ``AttributeError`` there and here (add the terms one by one; SURVEY.md App. B.7 read the branch as
        working)."""
        if not isinstance(other, list):
            other_names = [other.name]
            other = [other]
        else:
            other_names = [t.name for t in other.terms]

        name = ""
        if self.name is not None:
            name += self.name
        for other_name in other_names:
            if other_name is not None:
                if len(name):
                    name += " + " + other_name
                else:
                    name += other_name

        return StellarOscillatorKernel._from_terms(
            list(self.term.terms) + other, delta=self.delta, name=name
        )


class SolarOscillatorKernel(StellarOscillatorKernel):
    """``StellarOscillatorKernel`` with the solar hyperparameters
    (reference core.py:430-461)."""

    def __init__(self, texp=None, delta=None, bandpass=None, name=None):
        hp = Hyperparameters.for_star(
            mass=1.0, radius=1.0, temperature=5777.0, luminosity=1.0,
            bandpass=bandpass,
        )
        super().__init__(hp, texp=texp, delta=delta, name=name)


class ShotNoiseKernel(_terms.SHOTerm):
    """SHO term approximating white shot noise (reference core.py:464-544)."""

    # intentionally very large, in [uHz] (reference core.py:471)
    w0 = 1e7
    # value does not matter much if w0 >>> 1 (reference core.py:474)
    Q = 0.5

    def __init__(self, *args, name=None, **kwargs):
        if name is None:
            name = "Shot noise"
        if args:
            # celerite2's SHOTerm is keyword-only; be explicit about it
            raise TypeError("ShotNoiseKernel takes keyword arguments S0, w0, Q")
        super().__init__(name=name, **kwargs)

    @classmethod
    def from_kepler_light_curve(cls, light_curve):
        """Shot noise from a Kepler light curve's ``KEPMAG``
        (reference core.py:492-520; Jenkins et al. 2010)."""
        try:
            from lightkurve import LightCurveCollection
            if isinstance(light_curve, LightCurveCollection):
                light_curve = light_curve.stitch(lambda x: x)
        except ImportError:
            pass
        kepler_mag = light_curve.meta["KEPMAG"]
        norm = 2 * np.pi / len(light_curve.time) ** 0.5
        _, unscaled_S0 = cls.kepler_mag_to_noise_amplitude(kepler_mag)
        S0 = (unscaled_S0 * norm) ** 0.5     # value in ppm
        return cls(S0=float(S0), w0=cls.w0, Q=cls.Q)

    @staticmethod
    def kepler_mag_to_noise_amplitude(kepler_mag):
        """Kepler noise in 6 hour bins (reference core.py:522-544); values in ppm^2."""
        c = 3.46 * 10 ** (0.4 * (12 - kepler_mag) + 8)
        sigma_lower = np.sqrt(
            c + 7e6 * np.max([np.ones_like(kepler_mag), kepler_mag / 14],
                             axis=0) ** 4
        ) / c
        sigma_upper = np.sqrt(c + 7e7) / c
        return 1e6 * np.array([sigma_lower, sigma_upper])
