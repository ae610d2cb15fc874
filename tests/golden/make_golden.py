#!/usr/bin/env python
"""
Generates tests/golden/*.npz (run from the repo root: python tests/golden/make_golden.py).

The reference's own implementation of this path cannot be imported here (celerite2, astropy
and tynt are not installed: ordinary ImportError) and holds no numerical golden
for it, so these vectors are NOT captures of the reference: they come from this repo's two
independent oracles --
  * oracle/dense.py : explicit K + LAPACK Cholesky (float64);
  * oracle/seq.py   : the celerite recurrences run in 80-bit np.longdouble
and are stored only when the two agree (gates below).  PARITY UNPINNED at the reference level.

The coefficient vectors of every case come from the oracle's own algebra (oracle/terms_ref.py), so the
product's SHOTerm / TermSum / TermConvolution (gadfly_amd/terms.py) is checked by something it did not
write (tests/test_terms.py compares the two).

Each file holds inputs (coefficient vectors, t, diag_user, diag_shift, y, normal draws n,
prediction times ts) and expected outputs (loglike, logdet, alpha = K^-1 y, Ln = L D^1/2 n,
mean at t (t=None), mean/var at ts), or info for the not-positive-definite case.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import dense, seq, terms_ref # noqa: E402
from tests import util                   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "sho_q100_n64": ("generic", dict(kind="sho_q100", N=64, yerr=0.05)),
    "sho_q100_n512_uniform": ("generic", dict(kind="sho_q100", N=512, irregular=False)),
    "overdamped_n300": ("generic", dict(kind="overdamped", N=300)),
    "q_half_n300": ("generic", dict(kind="q_half", N=300)),
    "mixed_n512": ("generic", dict(kind="mixed", N=512, yerr=0.2)),
    "gran5_n512": ("solar", dict(J=5, N=512)),
    "solar_j6_n2048": ("solar", dict(J=6, N=2048)),
    "solar_j6_n512_yerr0": ("solar", dict(J=6, N=512, yerr=0.0)),
    "solar_j20_n512_jitter": ("solar", dict(J=20, N=512, jitter_t=True)),
    "solar_j30_n1024_gaps": ("solar", dict(J=30, N=1024, gaps=True)),
    "solar_j30_n512_yerr0": ("solar", dict(J=30, N=512, yerr=0.0)),
}


def oracle_coefficients(k):
    """Celerite coefficients of a test kernel from the ORACLE's algebra (oracle/terms_ref.py, real-arithmetic
    restatement of SURVEY A.1 - A.3), not from the product's: the kernel object only supplies its
    parameters -- the (S0, w0, Q) triples and the exposure."""
    delta = getattr(k, "delta", None)
    base = k.term if delta is not None else k
    triples = [(s.S0, s.w0, s.Q) for s in base.terms]
    return terms_ref.kernel_coefficients(triples, delta)


def build(name, kind, kw):
    prob = util.solar_problem(**kw) if kind == "solar" else util.generic_problem(**kw)
    k, t, y, du = prob["kernel"], prob["t"], prob["y"], prob["diag_user"]
    ar, cr, ac, bc, cc, dc, shift = oracle_coefficients(k)
    co = (ar, cr, ac, bc, cc, dc)
    diag = du + shift
    N = len(t)
    np.random.seed(42)                      # the reference's test seed (test_core.py:21)
    n = np.random.randn(N, 2)
    rng = np.random.default_rng(7)
    ts = np.sort(rng.uniform(t[0] - 2 * (t[1] - t[0]), t[-1] + 2 * (t[1] - t[0]), 16))

    # dense float64
    ll_d = dense.log_likelihood(co, t, diag, y)
    alpha_d = dense.apply_inverse(co, t, diag, y)
    Ln_d = dense.dot_tril(co, t, diag, n)
    mean_t = y - du * alpha_d
    mean_ts, var_ts, _ = dense.predict(co, t, diag, y, ts)

    # semiseparable, 80-bit
    ld = np.longdouble
    c, a, U, V = seq.celerite_matrices(co, t, diag, dtype=ld)
    d, Wm, info = seq.factor(t.astype(ld), c, a, U, V)
    assert info == 0, name
    z = seq.solve_lower(t.astype(ld), c, U, Wm, y.astype(ld))
    logdet = np.sum(np.log(d))
    ll_s = -0.5 * (logdet + N * np.log(2 * ld(np.pi))) - 0.5 * np.sum(z * z / d)
    alpha_s = seq.apply_inverse(t.astype(ld), c, U, Wm, d, y.astype(ld))
    Ln_s = seq.dot_tril(t.astype(ld), c, U, Wm, d, n.astype(ld))
    _, _, Us, Vs = seq.celerite_matrices(co, ts, 0.0, dtype=ld)
    mean_ts_s = seq.predict_mean_at(t.astype(ld), c, U, V, alpha_s, ts.astype(ld), Us, Vs)

    def rel(x, ref):
        return float(np.max(np.abs(np.asarray(x, float) - ref)) / max(np.max(np.abs(ref)), 1e-300))

    gates = dict(ll=abs(float(ll_s) - ll_d) / abs(ll_d), alpha=rel(alpha_s, alpha_d),
                 Ln=rel(Ln_s, Ln_d), mean_ts=rel(mean_ts_s, mean_ts))
    assert gates["ll"] < 1e-11 and gates["alpha"] < 1e-8 and gates["Ln"] < 1e-9 \
        and gates["mean_ts"] < 1e-8, (name, gates)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        ar=ar, cr=cr, ac=ac, bc=bc, cc=cc, dc=dc, diag_shift=shift,
        t=t, diag_user=du, y=y, n=n, ts=ts,
        loglike=float(ll_s), logdet=float(logdet), alpha=np.asarray(alpha_s, float),
        Ln=np.asarray(Ln_s, float), mean_t=mean_t, mean_ts=np.asarray(mean_ts_s, float),
        var_ts=var_ts)
    return gates


def build_failing():
    prob = util.generic_problem("mixed", 200)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    ar, cr, ac, bc, cc, dc, shift = oracle_coefficients(k)
    du = prob["diag_user"].copy()
    du[120:] = -3.0 * (np.sum(ar) + np.sum(ac) + shift)      # k(0), from the oracle's coefficients
    c, a, U, V = seq.celerite_matrices((ar, cr, ac, bc, cc, dc), t, du + shift)
    _, _, info = seq.factor(t, c, a, U, V)
    assert info == 121
    np.savez_compressed(os.path.join(OUT, "not_positive_definite.npz"),
                        ar=ar, cr=cr, ac=ac, bc=bc, cc=cc, dc=dc, diag_shift=shift,
                        t=t, diag_user=du, y=y, info=info)


if __name__ == "__main__":
    for name, (kind, kw) in CASES.items():
        print(name, build(name, kind, kw))
    build_failing()
    print("wrote", OUT)
