"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path
(``gadfly_amd``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use anything under ``oracle/``.

Sequential numpy restatement of the celerite2 O(N W^2) semiseparable algorithms
that gadfly's GP path reaches through ``celerite2.GaussianProcess``
(/root/reference/gadfly/gp.py:202 compute, :350 log_likelihood, :370 apply_inverse,
:327 dot_tril, :232 conditional mean).  celerite2 (PyPI, version unpinned at
/root/reference/pyproject.toml:20) is NOT in /root/reference and not installed, so
these functions follow its published algorithm (Foreman-Mackey et al. 2017;
Foreman-Mackey 2018) as restated in SURVEY.md Appendix A.4-A.8.

PARITY UNPINNED at the reference level: no test in /root/reference pins a
log-likelihood, solve or draw numerically.  This
restatement is instead pinned against an independent dense O(N^3) Cholesky
(``oracle/dense.py``) and an 80-bit ``np.longdouble`` run of itself; the golden
vectors under ``tests/golden/`` were produced by ``tests/golden/make_golden.py``.

Every function accepts ``dtype`` (np.float64 or np.longdouble).
"""
import numpy as np


def celerite_matrices(coeffs, x, diag, dtype=np.float64):
    """``(c, a, U, V)`` from the six coefficient vectors.

    ``coeffs`` = (ar, cr, ac, bc, cc, dc); ``diag`` already includes any
    TermConvolution diagonal shift.
    """
    ar, cr, ac, bc, cc, dc = (np.asarray(v, dtype=dtype) for v in coeffs)
    x = np.asarray(x, dtype=dtype)
    N = x.shape[0]
    Jr, Jc = len(ar), len(ac)
    W = Jr + 2 * Jc
    c = np.empty(W, dtype=dtype)
    c[:Jr] = cr
    c[Jr::2] = cc
    c[Jr + 1::2] = cc
    a = np.asarray(diag, dtype=dtype) + (ar.sum() + ac.sum())
    a = np.broadcast_to(a, (N,)).astype(dtype)
    U = np.empty((N, W), dtype=dtype)
    V = np.empty((N, W), dtype=dtype)
    U[:, :Jr] = ar
    V[:, :Jr] = 1
    # theta = d * x : one rounded multiply (parity hazard i)
    theta = x[:, None] * dc[None, :]
    co, si = np.cos(theta), np.sin(theta)
    U[:, Jr::2] = ac * co + bc * si
    U[:, Jr + 1::2] = ac * si - bc * co
    V[:, Jr::2] = co
    V[:, Jr + 1::2] = si
    return c, a, U, V


def factor(t, c, a, U, V):
    """SURVEY.md A.5.  Returns ``(d, W, info)``; info = 0 or 1-based failing row."""
    dtype = U.dtype
    N, Wd = U.shape
    d = np.array(a, dtype=dtype, copy=True)
    Wm = np.array(V, dtype=dtype, copy=True)
    S = np.zeros((Wd, Wd), dtype=dtype)
    if d[0] <= 0:
        return d, Wm, 1
    Wm[0] = V[0] / d[0]
    for n in range(1, N):
        p = np.exp(c * (t[n - 1] - t[n]))
        S = np.outer(p, p) * (S + d[n - 1] * np.outer(Wm[n - 1], Wm[n - 1]))
        tmp = U[n] @ S
        d[n] = a[n] - tmp @ U[n]
        if not d[n] > 0:
            return d, Wm, n + 1
        Wm[n] = (V[n] - tmp) / d[n]
    return d, Wm, 0


def solve_lower(t, c, U, Wm, Y):
    """Z = L^-1 Y.  Y is (N,) or (N, R)."""
    dtype = U.dtype
    Y2 = np.asarray(Y, dtype=dtype)
    squeeze = Y2.ndim == 1
    Y2 = Y2.reshape(Y2.shape[0], -1)
    N, Wd = U.shape
    Z = Y2.copy()
    F = np.zeros((Wd, Y2.shape[1]), dtype=dtype)
    for n in range(1, N):
        p = np.exp(c * (t[n - 1] - t[n]))
        F = p[:, None] * (F + np.outer(Wm[n - 1], Z[n - 1]))
        Z[n] = Y2[n] - U[n] @ F
    return Z[:, 0] if squeeze else Z


def solve_upper(t, c, U, Wm, Y):
    """Z = L^-T Y."""
    dtype = U.dtype
    Y2 = np.asarray(Y, dtype=dtype)
    squeeze = Y2.ndim == 1
    Y2 = Y2.reshape(Y2.shape[0], -1)
    N, Wd = U.shape
    Z = Y2.copy()
    F = np.zeros((Wd, Y2.shape[1]), dtype=dtype)
    for n in range(N - 2, -1, -1):
        p = np.exp(c * (t[n] - t[n + 1]))
        F = p[:, None] * (F + np.outer(U[n + 1], Z[n + 1]))
        Z[n] = Y2[n] - Wm[n] @ F
    return Z[:, 0] if squeeze else Z


def matmul_lower(t, c, U, V, Y):
    """Z = Y + tril(U V^T o Phi, -1) Y (uses *input* rows)."""
    dtype = U.dtype
    Y2 = np.asarray(Y, dtype=dtype)
    squeeze = Y2.ndim == 1
    Y2 = Y2.reshape(Y2.shape[0], -1)
    N, Wd = U.shape
    Z = Y2.copy()
    F = np.zeros((Wd, Y2.shape[1]), dtype=dtype)
    for n in range(1, N):
        p = np.exp(c * (t[n - 1] - t[n]))
        F = p[:, None] * (F + np.outer(V[n - 1], Y2[n - 1]))
        Z[n] = Y2[n] + U[n] @ F
    return Z[:, 0] if squeeze else Z


def general_matmul_lower(t1, t2, c, U1, V2, Y):
    """Z[m] = sum_{t2[n] <= t1[m]} (U1[m] o exp(-c (t1[m]-t2[n]))) . V2[n] Y[n]."""
    dtype = U1.dtype
    Y2 = np.asarray(Y, dtype=dtype).reshape(len(t2), -1)
    M, Wd = U1.shape
    N = len(t2)
    Z = np.zeros((M, Y2.shape[1]), dtype=dtype)
    F = np.zeros((Wd, Y2.shape[1]), dtype=dtype)
    n = 0
    last = None
    for m in range(M):
        while n < N and t2[n] <= t1[m]:
            if last is not None:
                F = np.exp(c * (last - t2[n]))[:, None] * F
            F = F + np.outer(V2[n], Y2[n])
            last = t2[n]
            n += 1
        if last is not None:
            Z[m] = (U1[m] * np.exp(c * (last - t1[m]))) @ F
    return Z


def general_matmul_upper(t1, t2, c, V1, U2, Y):
    """Z[m] = sum_{t2[n] > t1[m]} (V1[m] o exp(-c (t2[n]-t1[m]))) . U2[n] Y[n]."""
    dtype = V1.dtype
    Y2 = np.asarray(Y, dtype=dtype).reshape(len(t2), -1)
    M, Wd = V1.shape
    N = len(t2)
    Z = np.zeros((M, Y2.shape[1]), dtype=dtype)
    F = np.zeros((Wd, Y2.shape[1]), dtype=dtype)
    n = N - 1
    last = None
    for m in range(M - 1, -1, -1):
        while n >= 0 and t2[n] > t1[m]:
            if last is not None:
                F = np.exp(c * (t2[n] - last))[:, None] * F
            F = F + np.outer(U2[n], Y2[n])
            last = t2[n]
            n -= 1
        if last is not None:
            Z[m] = (V1[m] * np.exp(c * (t1[m] - last))) @ F
    return Z


# ---- the GaussianProcess-level quantities (celerite2 core semantics) -------

def log_likelihood(t, c, a, U, V, y):
    """``_norm - 0.5 * sum(z^2/d)`` with z = solve_lower(y)."""
    d, Wm, info = factor(t, c, a, U, V)
    if info:
        return -np.inf, info
    z = solve_lower(t, c, U, Wm, y)
    N = len(t)
    logdet = np.sum(np.log(d))
    norm = -0.5 * (logdet + N * np.log(2 * np.pi))
    return norm - 0.5 * np.sum(z * z / d), 0


def apply_inverse(t, c, U, Wm, d, Y):
    Z = solve_lower(t, c, U, Wm, Y)
    Z = Z / (d if Z.ndim == 1 else d[:, None])
    return solve_upper(t, c, U, Wm, Z)


def dot_tril(t, c, U, Wm, d, Y):
    sq = np.sqrt(d)
    Y2 = np.asarray(Y, dtype=U.dtype)
    Y2 = Y2 * (sq if Y2.ndim == 1 else sq[:, None])
    return matmul_lower(t, c, U, Wm, Y2)


def predict_mean_at(t, c, U, V, alpha, tstar, Ustar, Vstar):
    """Conditional mean at new times, zero mean function."""
    lo = general_matmul_lower(tstar, t, c, Ustar, V, alpha)
    up = general_matmul_upper(tstar, t, c, Vstar, U, alpha)
    return (lo + up)[:, 0]
