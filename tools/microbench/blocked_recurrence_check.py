"""Blocked (rank-R) form of the scaled celerite recurrence; check against the sequential oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import util
from oracle import seq, cref

def run(J, N, R=16, block=64, ld=np.longdouble):
    prob = util.solar_problem(J, N)
    c, a, U, V = util.oracle_matrices(prob, seq)
    t, y = prob["t"], prob["y"]
    W = U.shape[1]
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    z_ref = cref.solve_lower(t, c, U, W_ref, y)
    # blocked, scaled coordinates (float64)
    T = np.zeros((64, 64))
    d = np.zeros(N); z = np.zeros(N)
    tref = t[0]
    n = 0
    while n < N:
        if n % block == 0 and n > 0:
            e = np.ones(64); e[:W] = np.exp(-c * (t[n] - tref))
            T *= np.outer(e, e)
            tref = t[n]
        m = min(R, N - n, block - n % block)
        Ut = np.zeros((64, R)); Vt = np.zeros((64, R)); av = np.ones(R)
        rho = np.exp(-c[:, None] * (t[n:n+m] - tref)[None, :])
        Ut[:W, :m] = U[n:n+m].T * rho
        Vt[:W, :m] = V[n:n+m].T / rho
        Vt[63, :m] = y[n:n+m]
        av[:m] = a[n:n+m]
        P = T @ Ut                       # MFMA 1
        B = Vt - P
        H = B.T @ Ut                     # MFMA 2 (upper part used) ; diag: a - u^T p
        Hd = av - np.einsum("jm,jm->m", Ut, P)
        # LDL^T on the R x R matrix (row-oriented, upper triangle), with L^-1 accumulation
        C = np.triu(H, 1).copy()
        dd = Hd.copy()
        for k in range(R):
            for j in range(k + 1, R):
                dd[j] -= C[k, j] ** 2 / dd[k]
                C[j, j+1:] -= C[k, j] * C[k, j+1:] / dd[k]
        Cp = C / dd[:, None]             # c_km / d_k, strictly upper
        Minv = np.linalg.inv(np.eye(R) + Cp)      # (I + C')^-1 ; R_blk = B Minv
        Rs = B @ (Minv / np.sqrt(dd)[None, :])    # MFMA 3: scaled r~_m / sqrt(d_m)
        d[n:n+m] = dd[:m]
        z[n:n+m] = Rs[63, :m] * np.sqrt(dd[:m])
        T += Rs @ Rs.T                   # MFMA 4
        n += m
    ll_ref = -0.5 * (np.sum(np.log(d_ref)) + np.sum(z_ref**2 / d_ref))
    ll = -0.5 * (np.sum(np.log(d)) + np.sum(z**2 / d))
    print(f"J={J} N={N} R={R}: max rel d {np.max(np.abs(d-d_ref)/d_ref):.2e}  z {np.max(np.abs(z-z_ref))/np.max(np.abs(z_ref)):.2e}"
          f"  ll rel {abs(ll-ll_ref)/abs(ll_ref):.2e}   sum z^2/d via T[63,63]: n/a")
for J, N in [(6, 3000), (30, 5000), (30, 40000)]:
    run(J, N)
