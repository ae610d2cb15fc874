"""
Optional astropy interoperability.

The reference strips ``astropy.units`` at every entry point of the GP path
(/root/reference/gadfly/gp.py:61-165) and treats unit-less ndarrays as "already in
gadfly's native units" (gp.py:82-84, :111-113): times in 1/uHz (= 1e6 s), fluxes in
ppm.  astropy is not installed in the build image, so it is imported lazily and
only when a caller actually passes a ``Quantity`` / ``Time`` or asks for one back.
"""
import warnings

try:                                    # pragma: no cover - depends on environment
    import astropy.units as u
    from astropy.units import cds       # noqa: F401  (registers ppm)
    from astropy.time import Time
    from astropy.utils.exceptions import AstropyUserWarning as GadflyWarning
    HAS_ASTROPY = True
except Exception:                       # ImportError or a broken install
    u = None
    Time = None
    HAS_ASTROPY = False

    class GadflyWarning(UserWarning):
        """Stand-in for ``astropy.utils.exceptions.AstropyUserWarning``."""


SECONDS_PER_INVERSE_UHZ = 1.0e6


def has_unit(x):
    return hasattr(x, "unit")


def is_time(x):
    return Time is not None and isinstance(x, Time)


def require_astropy(what):
    if not HAS_ASTROPY:
        raise ImportError(
            f"{what} requires astropy, which is not installed; pass plain "
            "numpy arrays in gadfly's native units (times in 1/uHz = 1e6 s, "
            "fluxes in ppm) instead."
        )


def exposure_to_delta(texp):
    """Exposure time -> ``delta`` in 1/uHz (/root/reference/gadfly/core.py:392).

    Accepts an astropy Quantity (any time unit) or, as an offline extension, a
    plain number interpreted as seconds.
    """
    if has_unit(texp):
        require_astropy("a Quantity exposure time")
        return float(texp.to(1 / u.uHz).value)
    return float(texp) / SECONDS_PER_INVERSE_UHZ


def warn(msg):
    warnings.warn(msg, GadflyWarning, stacklevel=3)
