"""
A ~40-line stand-in for ``astropy.units`` / ``astropy.time.Time`` (astropy is not installed in the image): ``Q`` is an
ndarray subclass carrying ``.unit`` with ``.to`` / ``.value`` / ``is_equivalent`` and unit algebra for * and /;
``FakeTime`` carries ``.jd`` the way ``astropy.time.Time`` does (BKJD 0 = JD 2454833, which is what the reference's
``_time_to_freq`` turns into ``jd * u.day``, /root/reference/gadfly/gp.py:79-80).  Test inputs only: patched in as
``gadfly_amd.units.u`` / ``gadfly_amd.units.Time`` by the tests that exercise the Quantity branches.
"""
import types

import numpy as np


class Unit:
    __array_ufunc__ = None              # ndarray * Unit defers to Unit.__rmul__ (-> Q), as astropy's units do

    def __init__(self, scale, **dims):
        self.scale = float(scale)
        self.dims = {k: v for k, v in dims.items() if v}

    def _combine(self, other, sign):
        dims = dict(self.dims)
        for k, v in other.dims.items():
            dims[k] = dims.get(k, 0) + sign * v
        return Unit(self.scale * other.scale ** sign, **dims)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        return Q(np.asarray(other, dtype=float), self)       # unit * number

    def __rmul__(self, other):
        return Q(np.asarray(other, dtype=float), self)       # number * unit

    def __truediv__(self, other):
        return self._combine(other, -1)

    def __rtruediv__(self, other):
        assert other == 1
        return Unit(1.0)._combine(self, -1)

    def __pow__(self, p):
        return Unit(self.scale ** p, **{k: v * p for k, v in self.dims.items()})

    def is_equivalent(self, other):
        return self.dims == other.dims


ONE = Unit(1.0)


class Q(np.ndarray):
    def __new__(cls, value, unit=ONE):
        obj = np.asarray(value, dtype=float).view(cls)
        obj.unit = unit
        return obj

    def __array_finalize__(self, obj):
        self.unit = getattr(obj, "unit", ONE)

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        units = [getattr(i, "unit", ONE) for i in inputs]
        raw = [i.view(np.ndarray) if isinstance(i, Q) else i for i in inputs]
        if kw.get("out") is not None:
            kw["out"] = tuple(o.view(np.ndarray) if isinstance(o, Q) else o for o in kw["out"])
        out = getattr(ufunc, method)(*raw, **kw)
        if method == "__call__" and ufunc is np.multiply:
            unit = units[0]._combine(units[1], +1)
        elif method == "__call__" and ufunc is np.true_divide:
            unit = units[0]._combine(units[1], -1)
        elif method == "__call__" and ufunc in (np.add, np.subtract):
            assert units[0].is_equivalent(units[1]) and units[0].scale == units[1].scale
            unit = units[0]
        else:
            unit = units[0]
        if isinstance(out, np.ndarray) and out.dtype == bool:
            return out
        return Q(out, unit)

    def __mul__(self, other):
        if isinstance(other, Unit):                          # quantity * unit
            return Q(self.view(np.ndarray), self.unit._combine(other, +1))
        return super().__mul__(other)

    @property
    def value(self):
        v = self.view(np.ndarray)
        return v if v.ndim else float(v)

    def to(self, unit):
        assert self.unit.is_equivalent(unit), "incompatible units"
        return Q(self.view(np.ndarray) * (self.unit.scale / unit.scale), unit)


def fake_astropy_units():
    u = types.SimpleNamespace()
    u.s = Unit(1.0, s=1)
    u.min = Unit(60.0, s=1)
    u.day = u.d = Unit(86400.0, s=1)
    u.uHz = Unit(1e-6, s=-1)
    u.electron = Unit(1.0, electron=1)
    u.cds = types.SimpleNamespace(ppm=Unit(1e-6))
    u.dimensionless_unscaled = ONE
    u.Quantity = lambda value, unit=ONE: Q(value, unit)
    return u




class FakeTime:
    """``astropy.time.Time`` as far as the reference uses it: ``.jd`` (days)."""
    BKJD0 = 2454833.0

    def __init__(self, value, format="jd"):
        value = np.asarray(value, dtype=float)
        self.jd = value + (self.BKJD0 if format == "bkjd" else 0.0)

    def __add__(self, other):                                # Time + Quantity (days)
        u = fake_astropy_units()
        return FakeTime(self.jd + np.asarray(other.to(u.day).value))

    __radd__ = __add__

    def __len__(self):
        return len(self.jd)


def patch(monkeypatch, gunits):
    """Install the stand-ins into ``gadfly_amd.units``; returns the unit namespace."""
    u = fake_astropy_units()
    monkeypatch.setattr(gunits, "u", u)
    monkeypatch.setattr(gunits, "Time", FakeTime)
    monkeypatch.setattr(gunits, "HAS_ASTROPY", True)
    return u
