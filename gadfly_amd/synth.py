"""
Deterministic synthetic "solar-like" inputs for parity tests and benchmarks
(SURVEY.md section 8d).  No astropy / tynt needed.

Kernel recipe
-------------
* granulation: the first ``min(J, 5)`` (S0, w0, Q) triples verbatim from
  ``data/hyperparameters.json`` (same file as /root/reference/gadfly/data/hyperparameters.json:2-50);
* p-modes: the ``J - 5`` modes of ``data/broomhall2009_table2_labeled.ecsv``
  nearest nu_max = 3090 uHz (/root/reference/gadfly/scale.py:28), with
  w0 = 2 pi nu and the per-degree (S0_l, Q_l) of hyperparameters.json:51-90, reduced
  with the solar-case algebra of /root/reference/gadfly/core.py:279-310:
      Gamma_j = nu_j / (2 Q_l),  Q_j = Q_l * 1.02 / Gamma_j,
      S0_j = 0.5 * S0_l * (Gamma_j / 1.02) * E(nu_j) * B(w0_j)
  where B = sum of the granulation SHO PSDs at w0_j (core.py:33-41) and
  E = exp(-0.5 ((nu - 3090)/330)^2) is a Gaussian envelope standing in for the
  Kiefer-Voigt envelope of scale.py:515-539 (documented deviation: the envelope
  needs astropy.modeling, which is not available offline).

Times are in gadfly's native unit 1/uHz (= 1e6 s), fluxes in ppm.
"""
import json
import os

import numpy as np

from .core import Hyperparameters, default_hyperparameter_path

__all__ = [
    "solar_like_hyperparameters", "uniform_times", "jitter_hyperparameters",
    "scale_hyperparameters", "broomhall_modes", "cfg3_light_curves", "cfg4_walkers",
]

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_NU_MAX_SUN = 3090.0          # uHz
_ENVELOPE_SIGMA = 330.0       # uHz


def broomhall_modes(path=None):
    """(nu [uHz], degree) columns of the Broomhall et al. (2009) table, read as text."""
    if path is None:
        path = os.path.join(_DATA, "broomhall2009_table2_labeled.ecsv")
    nu, ell = [], []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#") or line.startswith("nu"):
                continue
            a, b = line.split()
            nu.append(float(a))
            ell.append(int(b))
    return np.array(nu), np.array(ell)


def _sho_psd(omega, S0, w0, Q):
    return (np.sqrt(2 / np.pi) * S0 * w0 ** 4
            / ((omega ** 2 - w0 ** 2) ** 2 + omega ** 2 * w0 ** 2 / Q ** 2))


def solar_like_hyperparameters(J, name=None):
    """``Hyperparameters`` with exactly ``J`` SHO terms (all underdamped => W = 2J)."""
    with open(default_hyperparameter_path) as fh:
        raw = json.load(fh)
    gran = [r for r in raw if r["metadata"]["source"] == "granulation"]
    osc = {r["metadata"]["degree"]: r["hyperparameters"]
           for r in raw if r["metadata"]["source"] == "oscillation"}
    out = []
    for r in gran[:min(J, len(gran))]:
        out.append(dict(hyperparameters=dict(r["hyperparameters"]),
                        metadata=dict(r["metadata"])))
    n_modes = J - len(out)
    if n_modes > 0:
        nu, ell = broomhall_modes()
        if n_modes > len(nu):
            raise ValueError(f"at most {len(nu) + len(gran)} terms available")
        order = np.argsort(np.abs(nu - _NU_MAX_SUN), kind="stable")[:n_modes]
        order = np.sort(order)
        gS0 = np.array([g["hyperparameters"]["S0"] for g in gran])
        gw0 = np.array([g["hyperparameters"]["w0"] for g in gran])
        gQ = np.array([g["hyperparameters"]["Q"] for g in gran])
        for j in order:
            hp = osc[int(ell[j])]
            w0 = 2 * np.pi * nu[j]
            Gamma = nu[j] / (2 * hp["Q"])
            Q = hp["Q"] * 1.02 / Gamma
            B = float(np.sum(_sho_psd(w0, gS0, gw0, gQ)))
            E = float(np.exp(-0.5 * ((nu[j] - _NU_MAX_SUN) / _ENVELOPE_SIGMA) ** 2))
            S0 = 0.5 * hp["S0"] * (Gamma / 1.02) * E * B
            out.append(dict(
                hyperparameters=dict(S0=float(S0), w0=float(w0), Q=float(Q)),
                metadata=dict(source="oscillation", scaled=True,
                              degree=int(ell[j]))))
    return Hyperparameters(out, name=name or f"synthetic solar-like J={J}")


def uniform_times(N, cadence_s=60.0):
    """t_n = n * cadence in units of 1/uHz (1e6 s)."""
    return np.arange(N, dtype=np.float64) * (cadence_s * 1e-6)


def jitter_hyperparameters(hp, seed, frac=0.10):
    """MCMC-walker style: log-uniform jitter of every (S0, w0, Q) by +-frac."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for p in hp:
        h = p["hyperparameters"]
        f = np.exp(rng.uniform(np.log1p(-frac), np.log1p(frac), size=3))
        out.append(dict(hyperparameters=dict(S0=h["S0"] * f[0], w0=h["w0"] * f[1],
                                             Q=max(h["Q"] * f[2], 0.5)),
                        metadata=dict(p["metadata"])))
    return Hyperparameters(out, name=hp.name)


def scale_hyperparameters(hp, nu_factor):
    """Batch-of-stars style: scale (w0, S0) by a nu_max factor."""
    out = []
    for p in hp:
        h = p["hyperparameters"]
        out.append(dict(hyperparameters=dict(S0=h["S0"] / nu_factor,
                                             w0=h["w0"] * nu_factor, Q=h["Q"]),
                        metadata=dict(p["metadata"])))
    return Hyperparameters(out, name=hp.name)


# ---- BASELINE.json's batched configurations, shared by tests/, tools/ and bench.py ----
def cfg3_light_curves(B=256, N=65_000, J=20, seed=2000, jitter=True):
    """cfg3: B stars = the solar-like kernel with (w0, S0) scaled by nu_max factors log-spaced
    0.3 ... 1.0, Kepler short cadence (58.85 s) with a per-star start time, own y and yerr.
    ``jitter``: odd stars get jittered time stamps (+-0.2 s), which the in-kernel row generator must
    take as exact rows (what the parity tests exercise; the benchmark's workload of record is the
    uniform cadence SURVEY.md 8d defines, with the jittered variant timed next to it).
    Returns (hyperparameter sets, t (B, N), y (B, N), yerr (B, N), exposure [s])."""
    base = solar_like_hyperparameters(J)
    hps = [scale_hyperparameters(base, f) for f in np.geomspace(0.3, 1.0, B)]
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(N)[None, :] * 58.85e-6 + rng.uniform(0.0, 1e-3, (B, 1))
    jit = rng.uniform(-2e-7, 2e-7, t[1::2].shape)
    if jitter:
        t[1::2] += jit
    y = rng.normal(size=(B, N)) * 50.0 + np.cumsum(rng.normal(size=(B, N)), axis=1)
    yerr = rng.uniform(20.0, 40.0, (B, 1)) * np.ones((1, N))
    return hps, t, y, yerr, 58.85


def cfg4_walkers(B=512, N=200_000, J=40, seed=12345):
    """cfg4: B MCMC walkers (hyperparameters jittered +-10 %, seed 1000 + id) on ONE series:
    shared t (60 s cadence) and y.  Returns (hyperparameter sets, t (N,), y (N,), exposure [s])."""
    base = solar_like_hyperparameters(J)
    hps = [jitter_hyperparameters(base, 1000 + i) for i in range(B)]
    rng = np.random.Generator(np.random.PCG64(seed))
    t = uniform_times(N, 60.0)
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    return hps, t, y, 60.0
