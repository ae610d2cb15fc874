"""Shared problem builders for the tests (inputs only; checkers live in oracle/)."""
import numpy as np

from gadfly_amd import StellarOscillatorKernel, Hyperparameters
from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times


def solar_problem(J, N, yerr=30.0, cadence=60.0, seed=12345, jitter_t=False, gaps=False):
    """Synthetic solar-like problem: returns dict with kernel, t, diag, y."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kernel = StellarOscillatorKernel(solar_like_hyperparameters(J), texp=cadence)
    t = uniform_times(N, cadence)
    if jitter_t:
        t = t + rng.uniform(-0.2, 0.2, N) * cadence * 1e-6
    if gaps:
        keep = np.ones(N, bool)
        keep[N // 3: N // 3 + N // 10] = False
        keep[rng.integers(0, N, N // 20)] = False
        t = t[keep]
    N = len(t)
    diag_user = np.full(N, float(yerr) ** 2)
    y = 100.0 * rng.normal(size=N) + np.cumsum(rng.normal(size=N))
    return dict(kernel=kernel, t=t, diag_user=diag_user, y=y, yerr=yerr)


def generic_kernel(kind):
    if kind == "sho_q100":       # notebooks/normalization_and_units.ipynb style single SHO
        return TermConvolution(TermSum(SHOTerm(S0=1.0, w0=2 * np.pi * 3.0, Q=100.0)), 0.01)
    if kind == "overdamped":
        return TermConvolution(TermSum(SHOTerm(S0=2.0, w0=1.5, Q=0.3)), 0.02)
    if kind == "q_half":
        return TermConvolution(TermSum(SHOTerm(S0=1.0, w0=2.0, Q=0.5)), 0.02)
    if kind == "mixed":
        return TermConvolution(TermSum(SHOTerm(S0=1.0, w0=3.0, Q=5.0),
                                       SHOTerm(S0=0.5, w0=1.0, Q=0.3),
                                       SHOTerm(S0=2.0, w0=0.7, Q=0.6)), 0.05)
    if kind == "plain_sum":      # no exposure integration
        return TermSum(SHOTerm(S0=1.0, w0=3.0, Q=5.0), SHOTerm(S0=0.3, w0=9.0, Q=20.0))
    raise KeyError(kind)


def generic_problem(kind, N, yerr=0.1, seed=3, irregular=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    kernel = generic_kernel(kind)
    if irregular:
        t = np.sort(rng.uniform(0, 0.08 * N, N))
        t = t[np.diff(t, prepend=-1.0) > 0.02]
    else:
        t = np.arange(N) * 0.08
    N = len(t)
    diag_user = np.full(N, float(yerr) ** 2)
    y = rng.normal(size=N)
    return dict(kernel=kernel, t=t, diag_user=diag_user, y=y, yerr=yerr)


def oracle_matrices(prob, seq, dtype=np.float64):
    """(c, a, U, V) from the oracle for a problem dict."""
    k = prob["kernel"]
    ar, cr, ac, bc, cc, dc, shift = k.get_device_coefficients()
    diag = prob["diag_user"] + shift
    return seq.celerite_matrices((ar, cr, ac, bc, cc, dc), prob["t"], diag, dtype=dtype)
