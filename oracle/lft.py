"""
TEST INFRASTRUCTURE (oracle): numpy restatement of the chunk-map algebra of the exact time-parallel
factorisation, used by tests/ to check the library's dense combine
(gf_lft_tree_scan / gf_wide_combine, gadfly_amd/csrc/gadfly_dense.hip).  Never imported by the product.

A chunk of rows maps its start state (X, Y) -- X = S + pending rank-1 update of celerite2's `factor`
recurrence, Y the same for `solve_lower` (celerite2.driver.factor and solve_lower,
reached from /root/reference/gadfly/gp.py:202 and :350) -- to its end state by

    X+ = Xbar + Phi K Phi^T,  K = (I - X G)^-1 X ;     Y+ = Ybar + Phi (I - X G)^-1 (Y - X m)

with M = (Phi, G, Xbar, Ybar, m) measured on the chunk swept from a zero start state.  The reference has
no counterpart (its factorisation is sequential); the property that pins these formulas is that applying
the maps one after the other reproduces the sequential sweep, which tests/test_gpu_configs.py checks
against oracle/celerite_ref.c row by row.
"""
import numpy as np


def apply(M, X, Y):
    """State (X, Y) through one chunk map."""
    Ph, G, Xb, Yb, m = M
    n = Ph.shape[0]
    A = np.eye(n) - X @ G
    sol = np.linalg.solve(A, np.concatenate([X, (Y - X @ m)[:, None]], axis=1))
    K, v = sol[:, :n], sol[:, n]
    K = 0.5 * (K + K.T)
    Xn = Xb + Ph @ K @ Ph.T
    return 0.5 * (Xn + Xn.T), Yb + Ph @ v


def compose(M1, M2):
    """The map of chunk 1 followed by chunk 2."""
    P1, G1, X1, Y1, m1 = M1
    P2, G2, X2, Y2, m2 = M2
    n = P1.shape[0]
    A = np.eye(n) - X1 @ G2
    sol = np.linalg.solve(A, np.concatenate([P1, X1, (Y1 - X1 @ m2)[:, None]], axis=1))
    DP1, DX1, v = sol[:, :n], sol[:, n:2 * n], sol[:, 2 * n]
    X12 = X2 + P2 @ DX1 @ P2.T
    G12 = G1 + P1.T @ G2 @ DP1
    return (P2 @ DP1, 0.5 * (G12 + G12.T), 0.5 * (X12 + X12.T), Y2 + P2 @ v, m1 + P1.T @ (m2 - G2 @ v))


def start_states(maps):
    """Start state of every chunk by applying the maps one after the other from a zero state."""
    n = maps[0][0].shape[0]
    X, Y = np.zeros((n, n)), np.zeros(n)
    out = []
    for M in maps:
        out.append((X, Y))
        X, Y = apply(M, X, Y)
    return out


def random_maps(rng, nch, W, rank=6):
    """Well-posed random chunk maps: contractive Phi, PSD Xbar, NSD G (the signs the recurrence produces
    keep I - X G well conditioned)."""
    maps = []
    for _ in range(nch):
        Ph = 0.7 * rng.normal(size=(W, W)) / np.sqrt(W)
        L = rng.normal(size=(W, rank))
        Mx = rng.normal(size=(W, rank))
        maps.append((Ph, -(Mx @ Mx.T) / rank, L @ L.T / rank, rng.normal(size=W), rng.normal(size=W)))
    return maps
