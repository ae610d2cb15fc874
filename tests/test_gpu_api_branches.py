"""
Branches of the drop-in class that no other test runs (rows a6-a8 of SURVEY.md 8a): ``condition`` / ``predict`` with
``kernel=other`` at the observed times and at new ones, ``include_mean=False``, ``return_cov``,
``ConditionalDistribution.sample`` and ``inplace=True`` -- celerite2's ``ConditionalDistribution`` /
``GaussianProcess`` semantics reached from /root/reference/gadfly/gp.py:206-239 (condition), :241-306 (predict),
:308-327 (dot_tril), :352-370 (apply_inverse).  Checked against the dense O(N^3) oracle (oracle/dense.py), which
shares nothing with the recurrences; bars: vectors 1e-6, variances / covariances 1e-5.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

TOL_VEC = 1e-6
TOL_COV = 1e-5


def _relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _other_kernel():
    """A different kernel for ``kernel=``: the granulation part alone (what the docs' gap filling separates)."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters
    return gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(3), texp=60.0)


@pytest.mark.parametrize("N,J", [(1300, 6), (9000, 20)], ids=["factor-on-demand", "factor-kept-by-compute"])
def test_conditional_distribution_with_another_kernel_and_without_mean(hip, N, J):
    import gadfly_amd
    from oracle import dense
    prob = util.solar_problem(J, N)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    other = _other_kernel()
    co = k.get_device_coefficients()
    coo = other.get_device_coefficients()
    diag = prob["diag_user"] + co[6]
    mean = 2.5
    gp = gadfly_amd.GaussianProcess(k, t=t, mean=mean, diag=prob["diag_user"])
    if N <= 2000:
        K = dense.dense_K(co[:6], t, diag)
        alpha = np.linalg.solve(K, y - mean)
    else:
        # (a dense solve of 9000 unknowns takes half a minute: K^-1 (y - mean) from the oracle's recurrences, which
        # tests/test_oracle.py pins to the dense formulation; the products with K_other below stay dense)
        from oracle import cref, seq
        c, a, U, V = util.oracle_matrices(prob, seq)
        d_ref, W_ref, info = cref.factor(t, c, a, U, V)
        assert info == 0
        alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y - mean) / d_ref)
    rng = np.random.default_rng(3)
    ts = np.sort(rng.uniform(t[0], t[-1], 70))

    # own kernel, include_mean=False: the mean function is left out of the result
    mu0 = gp.predict(y, include_mean=False)
    assert _relmax(mu0, y - prob["diag_user"] * alpha - mean) < TOL_VEC
    mus = gp.predict(y, t=ts, include_mean=False)
    assert _relmax(mus, dense.kernel_value(co[:6], ts[:, None] - t[None, :]) @ alpha) < TOL_VEC
    assert _relmax(gp.predict(y, t=ts), mus + mean) < 1e-12

    # kernel=other at the observed times: K_other(t, t) alpha in celerite form (lag 0 counts sum(a), no diagonal
    # shift: general_matmul_lower takes t_n <= t*_m with the other kernel's U*, V*)
    Koo = dense.kernel_value(coo[:6], t[:, None] - t[None, :])
    mo = gp.predict(y, kernel=other)
    assert _relmax(mo, Koo @ alpha + mean) < TOL_VEC
    assert _relmax(gp.predict(y, kernel=other, include_mean=False), Koo @ alpha) < TOL_VEC
    cond = gp.condition(y, kernel=other)
    assert isinstance(cond, gadfly_amd.ConditionalDistribution) and _relmax(cond.mean, mo) < 1e-12
    # ... and at new times
    Kos = dense.kernel_value(coo[:6], ts[:, None] - t[None, :])
    mos = gp.predict(y, t=ts, kernel=other)
    assert _relmax(mos, Kos @ alpha + mean) < TOL_VEC

    if N > 2000:
        return
    # variance and covariance with the other kernel: celerite2 takes K(t, t*) and k(0) from `kernel.get_value`
    Ks = other.get_value(t[:, None] - ts[None, :])
    sol = np.linalg.solve(K, Ks)
    var_ref = other.get_value(np.zeros(1))[0] - np.sum(Ks * sol, axis=0)
    cov_ref = other.get_value(ts[:, None] - ts[None, :]) - Ks.T @ sol
    m2, var = gp.predict(y, t=ts, kernel=other, return_var=True)
    assert _relmax(m2, mos) < 1e-12 and _relmax(var, var_ref) < TOL_COV
    m3, cov = gp.predict(y, t=ts, kernel=other, return_cov=True)
    assert _relmax(m3, mos) < 1e-12 and _relmax(cov, cov_ref) < TOL_COV
    # own kernel, return_cov at the observed times (t=None): N x N, the dense formulation celerite2 uses
    sub = slice(0, 400)
    gps = gadfly_amd.GaussianProcess(k, t=t[sub], mean=mean, diag=prob["diag_user"][sub])
    Ksub = K[sub, sub]
    Kxx = k.get_value(t[sub, None] - t[None, sub])
    mu_c, cov_c = gps.predict(y[sub], return_cov=True)
    cov_cref = Kxx - Kxx @ np.linalg.solve(Ksub, Kxx)
    assert _relmax(cov_c, cov_cref) < TOL_COV
    assert _relmax(mu_c, y[sub] - prob["diag_user"][sub] * np.linalg.solve(Ksub, y[sub] - mean)) < TOL_VEC


def test_conditional_sample_is_numpys_multivariate_normal_of_mean_and_covariance(hip):
    """celerite2: ``np.random.multivariate_normal(self.mean, self.covariance [+ regularize on the diagonal], size)``
    from the legacy global RNG."""
    import gadfly_amd
    from oracle import dense
    prob = util.solar_problem(6, 900)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    K = dense.dense_K(co[:6], t, prob["diag_user"] + co[6])
    gp = gadfly_amd.GaussianProcess(k, t=t, diag=prob["diag_user"], mean=1.0)
    ts = np.linspace(t[100], t[160], 25) + 7e-6
    Ks = k.get_value(t[:, None] - ts[None, :])
    mu_ref = dense.kernel_value(co[:6], ts[:, None] - t[None, :]) @ np.linalg.solve(K, y - 1.0) + 1.0
    cov_ref = k.get_value(ts[:, None] - ts[None, :]) - Ks.T @ np.linalg.solve(K, Ks)
    cond = gp.condition(y, t=ts)
    for size, reg in ((None, None), (4, None), (3, 1e-3)):
        np.random.seed(11)
        got = cond.sample(size=size, regularize=reg)
        np.random.seed(11)
        c = cov_ref.copy()
        if reg is not None:
            c[np.diag_indices_from(c)] += reg
        want = np.random.multivariate_normal(mu_ref, c, size=size)
        assert got.shape == want.shape
        scale = np.sqrt(np.max(np.diag(c)))
        assert np.max(np.abs(got - want)) < 1e-5 * scale, (size, reg)


@pytest.mark.parametrize("N", [700, 9000], ids=["short", "stored-factor"])
def test_inplace_overwrites_and_returns_the_callers_array(hip, N):
    """``inplace=True``: celerite2 hands the caller's float64 array to the driver, which overwrites it, and returns
    it (gp.py:327, :370 pass the flag through)."""
    import gadfly_amd
    from oracle import cref, seq
    prob = util.solar_problem(6, N)
    k, t = prob["kernel"], prob["t"]
    gp = gadfly_amd.GaussianProcess(k, t=t, diag=prob["diag_user"])
    c, a, U, V = util.oracle_matrices(prob, seq)
    d, Wm, info = cref.factor(t, c, a, U, V)
    assert info == 0
    rng = np.random.default_rng(8)
    for shape in ((N,), (N, 3)):
        y = rng.normal(size=shape)
        y2 = y.reshape(N, -1)
        want_dt = cref.matmul_lower(t, c, U, Wm, y2 * np.sqrt(d)[:, None]).reshape(shape)
        want_ai = cref.solve_upper(t, c, U, Wm, cref.solve_lower(t, c, U, Wm, y2) / d[:, None]).reshape(shape)
        a1 = y.copy()
        out = gp.dot_tril(a1, inplace=True)
        assert out is a1 and _relmax(a1, want_dt) < TOL_VEC
        a2 = y.copy()
        out = gp.apply_inverse(a2, inplace=True)
        assert out is a2 and _relmax(a2, want_ai) < TOL_VEC
        # the default leaves the input alone
        a3 = y.copy()
        out = gp.dot_tril(a3)
        assert out is not a3 and np.array_equal(a3, y) and _relmax(out, want_dt) < TOL_VEC
        out = gp.apply_inverse(a3)
        assert out is not a3 and np.array_equal(a3, y)
    # log_likelihood takes the flag too (the value does not depend on it)
    y = rng.normal(size=N) * 50.0
    assert gp.log_likelihood(y, inplace=True) == gp.log_likelihood(y)


def test_covariance_of_a_long_series_tail_rows_and_slab_budget(hip):
    """``covariance`` cuts the long dimension of K(t, t*)^T K^-1 K(t, t*) into slabs: N = 9001 is two slabs of 4500
    rows and ONE tail row (a sliced view handed to the GEMM by pointer), and the slab count is capped so that the
    partial products stay below a quarter of an N x M block."""
    import gadfly_amd
    from oracle import dense
    N = 9001
    prob = util.solar_problem(6, N)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    gp = gadfly_amd.GaussianProcess(k, t=t, diag=prob["diag_user"])
    rng = np.random.default_rng(4)
    ts = np.sort(rng.uniform(t[0], t[-1], 70))
    Ks = k.get_value(t[:, None] - ts[None, :])
    # (K^-1 K(t, t*) through the oracle's recurrences: a dense solve of 9001 unknowns takes half a minute)
    from oracle import cref, seq
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    sol = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Ks) / d_ref[:, None])
    cov_ref = k.get_value(ts[:, None] - ts[None, :]) - Ks.T @ sol
    _, cov = gp.predict(y, t=ts, return_cov=True)
    assert _relmax(cov, cov_ref) < TOL_COV
    _, var = gp.predict(y, t=ts, return_var=True)
    assert _relmax(var, np.diag(cov_ref)) < TOL_COV


@pytest.mark.parametrize("case", ["sorted-J30", "unsorted-mixed-terms", "solar-W172"])
def test_cross_covariance_blocks_against_the_kernel_function(hip, case):
    """gf_cross_covariance factors the exponentials per 256-row workgroup (queries beyond the workgroup's time span
    on either side) and evaluates queries inside the span entry by entry: every entry against ``Term.get_value`` of
    the coefficient form at 1e-12 of the prior variance -- sorted and unsorted time stamps, queries before, inside,
    between and after the data, 70 queries (two tiles of 64 / three of 32), real and complex terms, W = 172."""
    import torch
    from gadfly_amd import _lib
    from gadfly_amd.terms import TermSum, SHOTerm, Term
    lib = _lib.load()
    rng = np.random.default_rng(11)
    if case == "sorted-J30":
        k = util.solar_problem(30, 10)["kernel"]
        t = np.sort(rng.uniform(0.0, 0.6, 5000))
    elif case == "unsorted-mixed-terms":
        k = util.generic_kernel("mixed")
        t = rng.uniform(0.0, 40.0, 3000)               # not sorted: the workgroups' spans overlap
    else:
        import gadfly_amd
        k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
        t = np.arange(2500) * 60e-6
    span = t.max() - t.min()
    ts = np.concatenate([t.min() - span * rng.uniform(0.01, 0.5, 10), rng.uniform(t.min(), t.max(), 50),
                         t.max() + span * rng.uniform(0.01, 0.5, 8), t[[7, 1234]]])
    co = k.get_device_coefficients()
    dev = [torch.as_tensor(np.ascontiguousarray(v)).cuda() for v in co[:6]]
    td, tsd = torch.as_tensor(t).cuda(), torch.as_tensor(ts).cuda()
    N, R = len(t), len(ts)
    out = torch.empty((N, R), dtype=torch.float64, device="cuda")
    p = _lib.ptr
    rc = lib.gf_cross_covariance(1, N, R, len(co[0]), len(co[2]), *[p(v) if v.numel() else None for v in dev],
                                 p(td), N, p(tsd), R, p(out), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gf_cross_covariance")
    want = Term.get_value(_Coefficients(co), t[:, None] - ts[None, :])
    scale = Term.get_value(_Coefficients(co), np.zeros(1))[0]
    assert np.max(np.abs(out.cpu().numpy() - want)) <= 1e-12 * scale


class _Coefficients:
    """A term given by its coefficient arrays (the celerite form ``gf_cross_covariance`` evaluates)."""

    def __init__(self, co):
        self._co = tuple(np.asarray(v, dtype=np.float64) for v in co[:6])

    def get_coefficients(self):
        return self._co
