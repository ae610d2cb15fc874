"""CPU: the oracles against each other and against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from oracle import seq, dense, cref
from tests import util

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLDEN = [g for g in GOLDEN if "not_positive" not in g]


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - b)) / max(np.max(np.abs(b)), 1e-300)


def test_golden_files_present():
    assert len(GOLDEN) >= 10


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracles_match_golden(path):
    g = np.load(path)
    co = tuple(g[k] for k in ("ar", "cr", "ac", "bc", "cc", "dc"))
    t, y, n, ts = g["t"], g["y"], g["n"], g["ts"]
    diag = g["diag_user"] + float(g["diag_shift"])
    N = len(t)
    for impl in (seq, cref):                      # numpy float64 and the C restatement
        if impl is seq:
            c, a, U, V = seq.celerite_matrices(co, t, diag)
        else:
            c, a, U, V = cref.get_matrices(co, t, diag)
        d, Wm, info = impl.factor(t, c, a, U, V)
        assert info == 0
        z = impl.solve_lower(t, c, U, Wm, y)
        ll = -0.5 * (np.sum(np.log(d)) + N * np.log(2 * np.pi)) - 0.5 * np.sum(z * z / d)
        assert abs(ll - float(g["loglike"])) <= 1e-10 * abs(float(g["loglike"]))
        assert abs(np.sum(np.log(d)) - float(g["logdet"])) <= 1e-10 * abs(float(g["logdet"]))
        alpha = impl.solve_upper(t, c, U, Wm, z / d)
        assert _rel(alpha, g["alpha"]) < 1e-8
        Ln = impl.matmul_lower(t, c, U, Wm, n * np.sqrt(d)[:, None])
        assert _rel(Ln, g["Ln"]) < 1e-9
        assert _rel(y - g["diag_user"] * alpha, g["mean_t"]) < 1e-7
        _, _, Us, Vs = seq.celerite_matrices(co, ts, 0.0)
        if impl is seq:
            mu = seq.predict_mean_at(t, c, U, V, alpha, ts, Us, Vs)
        else:
            mu = cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha)
        assert _rel(mu, g["mean_ts"]) < 1e-8
    # the fused C entry point used as the CPU baseline
    ll2, info = cref.loglike(co, t, diag, y)
    assert info == 0 and abs(ll2 - float(g["loglike"])) <= 1e-10 * abs(float(g["loglike"]))
    # dense variance is self-consistent with the stored one
    _, var, _ = dense.predict(co, t, diag, y, ts)
    assert _rel(var, g["var_ts"]) < 1e-8


def test_not_positive_definite_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "not_positive_definite.npz"))
    co = tuple(g[k] for k in ("ar", "cr", "ac", "bc", "cc", "dc"))
    diag = g["diag_user"] + float(g["diag_shift"])
    c, a, U, V = seq.celerite_matrices(co, g["t"], diag)
    assert seq.factor(g["t"], c, a, U, V)[2] == int(g["info"])
    assert cref.factor(g["t"], c, a, U, V)[2] == int(g["info"])
    ll, info = cref.loglike(co, g["t"], diag, g["y"])
    assert info == int(g["info"]) and ll == -np.inf


def test_config0_plumbing_cpu():
    """BASELINE config[0]: N = 10,000, J = 6 log-likelihood on the CPU oracle; float64 vs
    80-bit agree to ~1e-14."""
    prob = util.solar_problem(6, 10_000)
    c, a, U, V = util.oracle_matrices(prob, seq)
    co = prob["kernel"].get_device_coefficients()
    ll_c, info = cref.loglike(co[:6], prob["t"], prob["diag_user"] + co[6], prob["y"])
    assert info == 0
    n = 2000                                    # 80-bit python loop on a prefix
    cl, al, Ul, Vl = util.oracle_matrices({k: (v[:n] if isinstance(v, np.ndarray) else v)
                                           for k, v in prob.items()}, seq, dtype=np.longdouble)
    ll_l, _ = seq.log_likelihood(prob["t"][:n].astype(np.longdouble), cl, al, Ul, Vl,
                                 prob["y"][:n].astype(np.longdouble))
    ll_p, _ = cref.loglike(co[:6], prob["t"][:n], prob["diag_user"][:n] + co[6], prob["y"][:n])
    assert abs(ll_p - float(ll_l)) <= 1e-12 * abs(float(ll_l))
    assert np.isfinite(ll_c)


def test_general_matmul_edge_cases():
    """Query times before the first / after the last observation and coincident times."""
    prob = util.generic_problem("mixed", 120)
    c, a, U, V = util.oracle_matrices(prob, seq)
    t = prob["t"]
    co = prob["kernel"].get_device_coefficients()[:6]
    alpha = np.linspace(-1, 1, len(t))
    ts = np.array([t[0] - 1.0, t[0], t[3], 0.5 * (t[5] + t[6]), t[-1], t[-1] + 2.0])
    _, _, Us, Vs = seq.celerite_matrices(co, ts, 0.0)
    mu = seq.predict_mean_at(t, c, U, V, alpha, ts, Us, Vs)
    ref = dense.kernel_value(co, ts[:, None] - t[None, :]) @ alpha
    assert _rel(mu, ref) < 1e-10
    assert _rel(cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha), ref) < 1e-10
