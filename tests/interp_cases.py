"""Seeded input series for the ``interpolate_missing_data`` fixtures (inputs only: shared by
tests/golden/reference/make_interp_golden.py, which runs the REFERENCE on them, and by the tests, which compare
the oracle restatement and the HIP kernels with the stored reference outputs)."""
import numpy as np

FULL_ARRAYS_BELOW = 6000        # cases with fewer output points store the full arrays


def _kepler_like(n_full, frac_missing, seed, jitter_s=0.0, long_gaps=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    keep = rng.uniform(size=n_full) > frac_missing
    for s in rng.integers(0, max(n_full - 500, 1), long_gaps):
        keep[s:s + int(rng.integers(2, 400))] = False
    keep[[0, -1]] = True
    cad = np.flatnonzero(keep) + 7000
    t = 2454833.0 + cad * (58.85 / 86400.0) + jitter_s * rng.uniform(-1, 1, cad.size) / 86400.0
    f = 1e4 + 50 * np.sin(cad * 0.01) + rng.normal(size=cad.size)
    return t, f, cad


def _drift():
    cad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
    t = cad * 1.0
    t[7] = 8.2                  # a late stamp: the grid time of missing cadence 8 lies before it
    return t, np.arange(10.0) ** 2, cad


def _drift_long(seed=5):
    """Barycentric-style slow drift of the stamps against the cadence grid: with cadence numbers
    given, dt is a median and several missing cadences' grid times cross their neighbours."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = 20000
    keep = rng.uniform(size=n) > 0.08
    keep[[0, -1]] = True
    cad = np.flatnonzero(keep)
    t = cad * 0.02 + 0.035 * np.sin(cad * 2e-3) + 2e-4 * rng.uniform(-1, 1, cad.size)
    t = np.sort(t)
    return t, np.cos(cad * 0.003) + 0.01 * rng.normal(size=cad.size), cad


CASES = {
    # name: (builder, uses cadence numbers)
    "uniform_300": (lambda: _kepler_like(300, 0.5, 1), False),
    "uniform_300_cad": (lambda: _kepler_like(300, 0.5, 1), True),
    "complete_2049": (lambda: _kepler_like(2049, 0.0, 2), False),
    "holes_5000": (lambda: _kepler_like(5000, 0.02, 3, long_gaps=4), False),
    "holes_5000_cad": (lambda: _kepler_like(5000, 0.02, 3, long_gaps=4), True),
    "jitter_5000": (lambda: _kepler_like(5000, 0.3, 4, jitter_s=2.0, long_gaps=4), False),
    "jitter_5000_cad": (lambda: _kepler_like(5000, 0.3, 4, jitter_s=2.0, long_gaps=4), True),
    "drift_10_cad": (_drift, True),
    "drift_20000_cad": (_drift_long, True),
    "jitter_100000": (lambda: _kepler_like(100000, 0.05, 6, jitter_s=0.5, long_gaps=4), False),
    "jitter_100000_cad": (lambda: _kepler_like(100000, 0.05, 6, jitter_s=0.5, long_gaps=4), True),
}


def make_case(name):
    builder, with_cad = CASES[name]
    t, f, cad = builder()
    return t, f, ({"cadences": cad} if with_cad else {})
