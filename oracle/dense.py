"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/seq.py header for the rules).

Independent dense O(N^3) restatement used to pin ``oracle/seq.py`` and the C
restatement: build K explicitly from the kernel function and use LAPACK Cholesky.
Nothing here shares code with the semiseparable recurrences, so agreement
between the two is a genuine check of the algorithm.

K is the matrix celerite2 actually factorises:
    K = diag(a) + tril(U V^T o Phi, -1) + its transpose
    k(tau) = sum_r a_r e^{-c_r tau} + sum_c (a_c cos d_c tau + b_c sin d_c tau) e^{-c_c tau}
PARITY UNPINNED at the reference level (no numerical golden in /root/reference).
"""
import numpy as np
import scipy.linalg as sla


def kernel_value(coeffs, tau):
    ar, cr, ac, bc, cc, dc = (np.asarray(v, dtype=np.float64) for v in coeffs)
    tau = np.abs(np.asarray(tau, dtype=np.float64))[..., None]
    k = np.sum(ar * np.exp(-cr * tau), axis=-1)
    k = k + np.sum((ac * np.cos(dc * tau) + bc * np.sin(dc * tau))
                   * np.exp(-cc * tau), axis=-1)
    return k


def dense_K(coeffs, x, diag):
    x = np.asarray(x, dtype=np.float64)
    K = kernel_value(coeffs, x[:, None] - x[None, :])
    K[np.diag_indices_from(K)] += np.broadcast_to(
        np.asarray(diag, dtype=np.float64), x.shape)
    return K


def log_likelihood(coeffs, x, diag, y):
    K = dense_K(coeffs, x, diag)
    L = np.linalg.cholesky(K)
    z = sla.solve_triangular(L, y, lower=True)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    return -0.5 * (z @ z + logdet + len(x) * np.log(2 * np.pi))


def apply_inverse(coeffs, x, diag, Y):
    K = dense_K(coeffs, x, diag)
    return sla.cho_solve(sla.cho_factor(K, lower=True), Y)


def dot_tril(coeffs, x, diag, Y):
    """L_chol @ Y with K = L_chol L_chol^T  (== celerite's L D^{1/2} Y)."""
    return np.linalg.cholesky(dense_K(coeffs, x, diag)) @ Y


def predict(coeffs, x, diag, y, xstar=None, kstar_fn=None, kss_fn=None):
    """Conditional mean (and variance, covariance) with zero mean function.

    ``kstar_fn(tau)`` / ``kss_fn(tau)`` default to the coefficient kernel.
    Returns (mean, var, cov); for xstar=None the prediction is at x itself:
    mean = K0 K^-1 y where K0 = K - diag.
    """
    K = dense_K(coeffs, x, diag)
    alpha = sla.cho_solve(sla.cho_factor(K, lower=True), y)
    if kstar_fn is None:
        kstar_fn = lambda tau: kernel_value(coeffs, tau)   # noqa: E731
    if kss_fn is None:
        kss_fn = kstar_fn
    if xstar is None:
        mean = y - np.broadcast_to(diag, y.shape) * alpha
        return mean, None, None
    Ks = kstar_fn(np.asarray(xstar)[:, None] - np.asarray(x)[None, :])
    mean = Ks @ alpha
    Kss = kss_fn(np.asarray(xstar)[:, None] - np.asarray(xstar)[None, :])
    cov = Kss - Ks @ sla.cho_solve(sla.cho_factor(K, lower=True), Ks.T)
    return mean, np.diag(cov).copy(), cov
