"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/celerite_ref.c).  ctypes front-end of the
C restatement; used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libceleriteref.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False):
    if force or not os.path.exists(_SO) or (
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "celerite_ref.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libceleriteref.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ref_factor.restype = _i64
        _lib.ref_loglike.restype = _i64
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def get_matrices(coeffs, x, diag):
    ar, cr, ac, bc, cc, dc = (_f64(v) for v in coeffs)
    x = _f64(x)
    N, Jr, Jc = len(x), len(ar), len(ac)
    J = Jr + 2 * Jc
    diag = _f64(np.broadcast_to(diag, (N,)))
    c = np.empty(J)
    a = np.empty(N)
    U = np.empty((N, J))
    V = np.empty((N, J))
    lib().ref_get_matrices(_i64(N), _int(Jr), _int(Jc), _p(ar), _p(cr), _p(ac), _p(bc),
                           _p(cc), _p(dc), _p(x), _p(diag), _p(c), _p(a), _p(U), _p(V))
    return c, a, U, V


def factor(t, c, a, U, V):
    t, c, a, U, V = (_f64(v) for v in (t, c, a, U, V))
    N, J = U.shape
    d = np.empty(N)
    W = np.empty((N, J))
    info = lib().ref_factor(_i64(N), _int(J), _p(t), _p(c), _p(a), _p(U), _p(V), _p(d), _p(W))
    return d, W, int(info)


def _rhs(Y, N):
    Y = _f64(Y)
    squeeze = Y.ndim == 1
    return Y.reshape(N, -1).copy(), squeeze


def solve_lower(t, c, U, W, Y):
    t, c, U, W = (_f64(v) for v in (t, c, U, W))
    N, J = U.shape
    Y2, sq = _rhs(Y, N)
    Z = np.empty_like(Y2)
    lib().ref_solve_lower(_i64(N), _int(J), _int(Y2.shape[1]), _p(t), _p(c), _p(U), _p(W),
                          _p(Y2), _p(Z))
    return Z[:, 0] if sq else Z


def solve_upper(t, c, U, W, Y):
    t, c, U, W = (_f64(v) for v in (t, c, U, W))
    N, J = U.shape
    Y2, sq = _rhs(Y, N)
    Z = np.empty_like(Y2)
    lib().ref_solve_upper(_i64(N), _int(J), _int(Y2.shape[1]), _p(t), _p(c), _p(U), _p(W),
                          _p(Y2), _p(Z))
    return Z[:, 0] if sq else Z


def matmul_lower(t, c, U, V, Y):
    t, c, U, V = (_f64(v) for v in (t, c, U, V))
    N, J = U.shape
    Y2, sq = _rhs(Y, N)
    Z = np.empty_like(Y2)
    lib().ref_matmul_lower(_i64(N), _int(J), _int(Y2.shape[1]), _p(t), _p(c), _p(U), _p(V),
                           _p(Y2), _p(Z))
    return Z[:, 0] if sq else Z


def general_matmul(t1, t2, c, U1, V1, U2, V2, Y):
    """lower(U1, V2) + upper(V1, U2) applied to the vector Y (predict at new times)."""
    t1, t2, c, U1, V1, U2, V2, Y = (_f64(v) for v in (t1, t2, c, U1, V1, U2, V2, Y))
    M, J = U1.shape
    N = len(t2)
    Z = np.zeros(M)
    lib().ref_general_matmul_lower(_i64(M), _i64(N), _int(J), _p(t1), _p(t2), _p(c), _p(U1),
                                   _p(V2), _p(Y), _p(Z))
    lib().ref_general_matmul_upper(_i64(M), _i64(N), _int(J), _p(t1), _p(t2), _p(c), _p(V1),
                                   _p(U2), _p(Y), _p(Z))
    return Z


def loglike(coeffs, t, diag, y, work=None):
    """One full evaluation (build + factor + solve + reductions). Returns (ll, info)."""
    ar, cr, ac, bc, cc, dc = (_f64(v) for v in coeffs)
    t = _f64(t)
    N, Jr, Jc = len(t), len(ar), len(ac)
    J = Jr + 2 * Jc
    diag = _f64(np.broadcast_to(diag, (N,)))
    y = _f64(y)
    if work is None:
        work = np.empty(N * (3 * J + 3) + J)
    out = ctypes.c_double(0.0)
    info = lib().ref_loglike(_i64(N), _int(Jr), _int(Jc), _p(ar), _p(cr), _p(ac), _p(bc),
                             _p(cc), _p(dc), _p(t), _p(diag), _p(y), _p(work),
                             ctypes.byref(out))
    return out.value, int(info)
