"""
TEST INFRASTRUCTURE (oracle): the celerite coefficient algebra of gadfly's kernels restated in plain real
arithmetic, independently of gadfly_amd/terms.py (which works with complex amplitudes):

    kernel = TermConvolution(TermSum(SHOTerm(S0, w0, Q) x J), delta)

as /root/reference/gadfly/core.py:371-373, :379, :394 assembles it from celerite2's terms (third-party,
not in /root/reference: SURVEY.md 8c).  Formulas: SURVEY.md Appendix A.1 (SHOTerm), A.2 (TermSum), A.3
(TermConvolution: transformed amplitudes and the diagonal correction), which restate celerite2's published
algorithm (Foreman-Mackey et al. 2017; Foreman-Mackey 2018) and were verified there against brute-force
double integration.  tests/golden/make_golden.py takes the coefficient inputs of every golden vector from
HERE, and tests/test_terms.py checks gadfly_amd's own algebra against it.  Never imported by the product.
"""
import numpy as np


def sho_coefficients(S0, w0, Q, eps=1e-5):
    """(a_r, c_r, a_c, b_c, c_c, d_c) of one SHO term."""
    S0, w0, Q = float(S0), float(w0), float(Q)
    e = np.empty(0)
    if Q < 0.5:                                         # overdamped: two real exponentials
        f = np.sqrt(max(1.0 - 4.0 * Q * Q, eps))
        return (0.5 * S0 * w0 * Q * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f]),
                w0 / (2.0 * Q) * np.array([1.0 - f, 1.0 + f]), e, e, e, e)
    f = np.sqrt(max(4.0 * Q * Q - 1.0, eps))           # underdamped (includes Q = 1/2)
    a = S0 * w0 * Q
    c = w0 / (2.0 * Q)
    return e, e, np.array([a]), np.array([a / f]), np.array([c]), np.array([c * f])


def sum_coefficients(terms):
    """concatenation over the terms, in term order; terms = [(S0, w0, Q), ...]."""
    parts = [sho_coefficients(*t) for t in terms]
    return tuple(np.concatenate([p[k] for p in parts]) if parts else np.empty(0) for k in range(6))


def convolve(coeffs, delta):
    """exposure-time integration over a boxcar of length delta.  Returns the transformed
    (a_r, c_r, a_c, b_c, c_c, d_c) and the correction of the diagonal."""
    ar, cr, ac, bc, cc, dc = (np.asarray(v, dtype=np.float64) for v in coeffs)
    delta = float(delta)
    # real terms: a -> 2 a (cosh(c d) - 1) / (c d)^2 ; diag += 2 a (c d - sinh(c d)) / (c d)^2
    x = cr * delta
    ar2 = 2.0 * ar * (np.cosh(x) - 1.0) / x ** 2
    shift = float(np.sum(2.0 * ar * (x - np.sinh(x)) / x ** 2))
    # complex terms
    c2, d2 = cc * cc, dc * dc
    C1 = ac * (c2 - d2) + 2.0 * bc * cc * dc
    C2 = bc * (c2 - d2) - 2.0 * ac * cc * dc
    den = (delta * (c2 + d2)) ** 2
    ch, sh = np.cosh(cc * delta), np.sinh(cc * delta)
    co, si = np.cos(dc * delta), np.sin(dc * delta)
    ct = ch * co - 1.0
    st = sh * si
    ac2 = 2.0 * (C1 * ct - C2 * st) / den
    bc2 = 2.0 * (C2 * ct + C1 * st) / den
    shift += float(np.sum(2.0 * (C2 * ch * si - C1 * sh * co + (ac * cc + bc * dc) * delta * (c2 + d2)) / den))
    return (ar2, cr.copy(), ac2, bc2, cc.copy(), dc.copy()), shift


def kernel_coefficients(terms, delta=None):
    """(a_r, c_r, a_c, b_c, c_c, d_c, diag_shift) of TermSum(SHO terms) [convolved with delta]."""
    co = sum_coefficients(terms)
    if delta is None:
        return co + (0.0,)
    co2, shift = convolve(co, delta)
    return co2 + (shift,)


def sho_psd(omega, S0, w0, Q):
    """SURVEY A.1 (= /root/reference/gadfly/core.py:33-41, pinned by tests/golden/reference)."""
    omega = np.asarray(omega, dtype=np.float64)
    return np.sqrt(2.0 / np.pi) * S0 * w0 ** 4 / ((omega ** 2 - w0 ** 2) ** 2 + omega ** 2 * w0 ** 2 / Q ** 2)
