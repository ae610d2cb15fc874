"""
The two-sweep log-likelihood (engine.two_sweep: nominal pass + transition sweep + per-chunk corrections, no final
pass) must never turn a matrix celerite2 rejects into a finite number (/root/reference/gadfly/gp.py:188-192:
``LinAlgError``, or -inf under ``quiet=True``).

det(I - X G) <= 0 catches an ODD number of non-positive pivots in a chunk.  An EVEN number with positive nominal
pivots does exist -- rows whose negative diagonal sits between their conditional variance given the rest of the
chunk and the one given the past as well -- and is planted here: the corrections' pivot-sign check (a Cholesky
attempt on (I - X G) X, k_spd_check) must turn the value non-finite, and ``BatchedLogLikelihood`` must come back
with -inf and celerite2's failing row.  Also: the evaluator does not take the route at all for an
exposure-integrated kernel sampled closer than its exposure time (the celerite form is then not a covariance).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8


def _pivots(t, c, a, U, V, n0, n1, X0=None):
    """Pivots of rows n0 .. n1 - 1 of celerite's recurrence from the state X0 before row n0,
    continuing THROUGH non-positive pivots (the oracle stops at the first one, as celerite2 does).  Returns (d, X)
    with X = S + pending update after row n1 - 1.  Test-input validation only."""
    W = U.shape[1]
    X = np.zeros((W, W)) if X0 is None else X0.copy()
    d = np.empty(n1 - n0)
    for n in range(n0, n1):
        p = np.exp(c * (t[n - 1] - t[n])) if n > 0 else np.ones(W)
        S = np.outer(p, p) * X
        tmp = S @ U[n]
        dn = a[n] - U[n] @ tmp
        w = (V[n] - tmp) / dn
        d[n - n0] = dn
        X = S + dn * np.outer(w, w)
    return d, X


def _planted(extra, N=8192, L=512, row=4096):
    """Three high-Q oscillators (a nominal pass needs ~6 rows of a chunk to learn their state; the true sweep knows
    it from the past) + `extra` negligible terms to reach a wide kernel; rows `row` and `row + 3` -- the first rows
    of a chunk of L -- carry a negative diagonal inside the window described above: two TRUE pivots a few rows
    later come out negative, no nominal one does."""
    from gadfly_amd.terms import SHOTerm, TermSum
    from oracle import seq
    Q = 1.0e4
    terms = [SHOTerm(S0=1.0 / (w * Q), w0=w, Q=Q) for w in 2 * np.pi * np.array([0.31, 0.47, 0.73])]
    terms += [SHOTerm(S0=1e-13, w0=2 * np.pi * (0.05 + 0.4 * i / max(extra, 1)), Q=30.0) for i in range(extra)]
    kernel = TermSum(*terms)
    assert row % L == 0
    t = np.arange(N) * 1.0
    diag = np.full(N, 1e-2)
    diag[[row, row + 3]] = -0.006
    y = np.random.default_rng(0).normal(size=N)
    co = kernel.get_device_coefficients()
    c, a, U, V = seq.celerite_matrices(co[:6], t, diag + co[6])
    # the construction holds: nominal pivots of the chunk all positive, exactly two true pivots negative
    d_true, X = _pivots(t, c, a, U, V, row - 2 * L, row)        # (two chunks of history: the state has converged)
    assert np.all(d_true > 0)
    d_nom, _ = _pivots(t, c, a, U, V, row, row + L)
    d_chunk, _ = _pivots(t, c, a, U, V, row, row + L, X)
    assert np.all(d_nom > 0) and np.sum(d_chunk <= 0) == 2, (d_nom.min(), np.sum(d_chunk <= 0))
    return kernel, t, diag, y, L, row + int(np.argmax(d_chunk <= 0))       # (the first failing row, 0-based)


@pytest.mark.parametrize("extra", [0, 30], ids=["W6-fused", "W66-wide"])
def test_even_number_of_broken_pivots_is_caught(hip, extra):
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    # (the evaluator's own chunking for two series of 8192 rows: 512 rows at W <= 63, 2048 rows for a wide kernel)
    kernel, t, diag, y, L, row = _planted(extra, L=512 if extra == 0 else 2048)
    co = kernel.get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, diag + co[6], y)
    assert ref == -np.inf and info == row + 1               # celerite2's answer: that row (1-based) fails
    eng = StreamingBatch([co, co], t, y, diag=diag)
    eng.generator_period = 1
    eng.two_sweep = False
    ll3 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    assert np.all(ll3 == -np.inf) and np.all(eng.info.cpu().numpy() == row + 1)
    eng.two_sweep = True
    ll2 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    assert eng._two_sweep_used
    assert not np.any(np.isfinite(ll2)), ll2                # (round 3 returned a finite number here)
    # through the batched evaluator, even if its by-construction gate were wrong about this matrix: the
    # non-finite value is repeated with the final pass, which names the row
    ev = gadfly_amd.BatchedLogLikelihood([kernel, kernel], t, y, diag=diag)
    assert not ev._init_safe                                # a negative diagonal: the route is not taken ...
    ev._init_safe = True                                    # ... force it
    ev.engine.wide_tp_min_rows = 512
    got = ev.evaluate()
    assert (ev.engine._tp_key if extra == 0 else (ev.engine._wide_tp["chunk_len"], ev.engine._wide_tp["nch"])) \
        == (L, len(t) // L)
    assert np.all(got == -np.inf) and ev.guard_reruns == 2  # the two-sweep values were not finite: both repeated
    assert np.all(ev.engine.info.cpu().numpy() == row + 1)
    # the same series with a healthy diagonal: finite, equal to the oracle, no repeat
    good = np.full(len(t), 1e-2)
    ev2 = gadfly_amd.BatchedLogLikelihood([kernel, kernel], t, y, diag=good)
    assert ev2._init_safe
    ev2.engine.wide_tp_min_rows = 512
    got2 = ev2.evaluate()
    ref2, info2 = cref.loglike(co[:6], t, good + co[6], y)
    assert info2 == 0 and ev2.engine._two_sweep_used and ev2.guard_reruns == 0
    assert np.max(np.abs(got2 - ref2)) <= RTOL_LL * abs(ref2)


def test_integrated_kernel_sampled_inside_its_exposure_is_not_taken_for_a_covariance(hip):
    """``TermConvolution`` is exact for lags >= delta only; with stamps 0.2 apart and delta = 1 the celerite matrix is
    indefinite.  The default evaluator must answer what celerite2 answers: -inf, at the oracle's failing row."""
    import gadfly_amd
    from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
    from oracle import cref
    kernel = TermConvolution(TermSum(SHOTerm(S0=1.0, w0=2.0, Q=0.7)), 1.0)
    N = 20_000
    t = np.arange(N) * 0.2
    y = np.random.default_rng(1).normal(size=N)
    co = kernel.get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, np.zeros(N) + co[6], y)
    assert ref == -np.inf and info > 0
    ev = gadfly_amd.BatchedLogLikelihood([kernel], t, y)
    assert not ev._init_safe
    got = ev.evaluate()
    assert not ev.engine.two_sweep and got[0] == -np.inf and int(ev.engine.info[0]) == info
    # the same kernel resolved by the cadence IS taken (and matches)
    t2 = np.arange(N) * 1.0
    ev2 = gadfly_amd.BatchedLogLikelihood([kernel], t2, y, yerr=0.1)
    assert ev2._init_safe
    got2 = ev2.evaluate()
    ref2, info2 = cref.loglike(co[:6], t2, np.full(N, 0.01) + co[6], y)
    assert info2 == 0 and ev2.engine._two_sweep_used and abs(got2[0] - ref2) <= RTOL_LL * abs(ref2)


@pytest.mark.parametrize("Q,yerr", [(1.0e3, 0.02), (1.0e4, 1.0), (30.0, 0.0)], ids=["Q1e3", "Q1e4-noisy", "Q30-yerr0"])
def test_two_sweep_condition_estimate_covers_the_true_pivots(hip, Q, yerr):
    """A two-sweep evaluation sees the NOMINAL pivots only (every chunk from a zero state, never smaller than the
    true ones); its condition estimate divides their minimum by engine.TWO_SWEEP_MARGIN.  Long-coherence terms and
    the shortest chunks the engine cuts: the estimate must not fall below the true max(a) / min(d)."""
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.terms import SHOTerm, TermSum
    from oracle import cref, seq
    ws = 2 * np.pi * np.array([0.013, 0.031, 0.047, 0.073, 0.11, 0.19])
    kernel = TermSum(*[SHOTerm(S0=1.0 / (w * Q), w0=w, Q=Q) for w in ws])
    N = 4096
    t = np.arange(N) * 1.0
    y = np.random.default_rng(2).normal(size=N)
    diag = np.full(N, yerr ** 2)
    co = kernel.get_device_coefficients()
    c, a, U, V = seq.celerite_matrices(co[:6], t, diag + co[6])
    d, _, info = cref.factor(t, c, a, U, V)
    assert info == 0
    true_cond = a.max() / d.min()
    eng = StreamingBatch([co], t, y, diag=diag)
    eng.generator_period = 1
    eng.two_sweep = True
    ll = float(eng.log_likelihood_time_parallel(chunk_len=64)[0])
    assert eng._two_sweep_used
    ref, _ = cref.loglike(co[:6], t, diag + co[6], y)
    assert abs(ll - ref) <= RTOL_LL * abs(ref)
    est = eng.condition_estimate()
    assert est >= 0.99 * true_cond, (est, true_cond)


@pytest.mark.parametrize("what", ["W2", "W12", "W62", "mixed-real-terms", "overdamped"])
def test_single_kernel_corrections_at_every_width(hip, what):
    """The W <= 63 corrections run in ONE kernel per chunk map on the active width rounded up to 4 (k_corr_small: 4 x 4
    register tiles, two n x (n + 1) LDS buffers -- 70 KB at W = 62, which needs the large-LDS opt-in): the narrowest and
    the widest kernels, and kernels with real terms (the one-column-per-lane sweep), two sweeps against three and
    against the oracle, several chunkings incl. a ragged last chunk."""
    from tests import util
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    if what.startswith("W"):
        prob = util.solar_problem({"W2": 1, "W12": 6, "W62": 31}[what], 2600, yerr=30.0 if what != "W12" else 0.0)
    else:
        prob = util.generic_problem("mixed" if what.startswith("mixed") else "overdamped", 1500)
    co = prob["kernel"].get_device_coefficients()
    t, y = prob["t"], prob["y"]
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    eng = StreamingBatch([co, co, co], t, y, diag=prob["diag_user"])
    eng.generator_period = 1
    for L in (64, 192, 1000):
        eng.two_sweep = False
        ll3 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
        eng.two_sweep = True
        ll2 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
        assert eng._two_sweep_used or eng._tp_chunking(L)[1] == 1
        assert np.max(np.abs(ll2 - ref)) <= RTOL_LL * abs(ref), (what, L, ll2, ref)
        assert np.max(np.abs(ll2 - ll3)) <= 1e-10 * abs(ref), (what, L, ll2, ll3)


def test_jittered_stamps_keep_the_two_sweep_route(hip):
    """Real cadences jitter: cfg3's odd stars carry +-0.2 s on a 58.85 s cadence with a 58.85 s exposure, so some pairs
    of stamps are 0.7 % closer than the exposure.  The route is still taken (batch.EXPOSURE_SLACK) -- the pivot-sign
    check, not the construction, guards it there -- and every entry matches the oracle."""
    import gadfly_amd
    from gadfly_amd.synth import cfg3_light_curves
    from oracle import cref
    B, N = 6, 9000
    hps, t, y, yerr, texp = cfg3_light_curves(B, N, 20, jitter=True)
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
    assert float(np.min(np.diff(t, axis=1))) < texp * 1e-6
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr)
    assert ev._init_safe
    ll = ev.evaluate()
    assert ev.engine._two_sweep_used and ev.guard_reruns == 0
    for i, k in enumerate(kernels):
        co = k.get_device_coefficients()
        ref, info = cref.loglike(co[:6], t[i], yerr[i] ** 2 + co[6], y[i])
        assert info == 0 and abs(ll[i] - ref) <= RTOL_LL * abs(ref), (i, ll[i], ref)
