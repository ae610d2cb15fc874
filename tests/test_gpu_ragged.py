"""
Ragged batches and multi-quarter ingestion (BASELINE cfg3 is "independent Kepler-cadence light
curves" -- real quarters differ in length): ``BatchedLogLikelihood`` / ``log_likelihood_batch`` /
``dist.sharded_log_likelihood`` take lists of series of different lengths (own time axis, data and errors each), and
``stitch_quarters`` prepares the quarters of one star the way /root/reference/gadfly/psd.py:483-531 does
(per-quarter gap filling + polynomial normalisation, stitch, gap filling across the quarters; the reference's
``core.py:509-512`` stitches a ``LightCurveCollection`` the same way).

Every log-likelihood against the oracle's C restatement on the UNPADDED series, 1e-8; the stitched series against
the oracle's numpy restatement bit for bit.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8


def _series(J, lengths, seed=7, cadence=58.85):
    """B stars: own kernel (nu_max scaled), own start time and cadence, own errors, `lengths[i]` rows."""
    import gadfly_amd
    from gadfly_amd.synth import scale_hyperparameters, solar_like_hyperparameters
    rng = np.random.default_rng(seed)
    base = solar_like_hyperparameters(J)
    B = len(lengths)
    kernels = [gadfly_amd.StellarOscillatorKernel(scale_hyperparameters(base, f), texp=cadence)
               for f in np.geomspace(0.4, 1.0, B)]
    t = [rng.uniform(0.0, 1e-3) + np.arange(n) * cadence * 1e-6 * (1.0 + 1e-3 * i) for i, n in enumerate(lengths)]
    y = [50.0 * rng.normal(size=n) + np.cumsum(rng.normal(size=n)) for n in lengths]
    yerr = [np.full(n, rng.uniform(20.0, 40.0)) for n in lengths]
    return kernels, t, y, yerr


def _refs(kernels, t, y, yerr, mean=0.0):
    from oracle import cref
    out = []
    for k, tt, yy, ee in zip(kernels, t, y, yerr):
        co = k.get_device_coefficients()
        v, info = cref.loglike(co[:6], tt, ee ** 2 + co[6], yy - mean)
        assert info == 0
        out.append(v)
    return np.array(out)


@pytest.mark.parametrize("route", ["streamed", "time-parallel", "time-parallel-three-sweeps"])
def test_nine_light_curves_of_nine_lengths(hip, route):
    import gadfly_amd
    lengths = [9000, 20011, 12345, 17000, 9001, 15500, 20000, 11111, 13000]
    kernels, t, y, yerr = _series(20, lengths)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr, mean=1.5)
    assert list(ev.rows) == lengths and ev.engine.N == max(lengths)
    if route == "streamed":
        ev.engine.force_streaming = True
    if route == "time-parallel-three-sweeps":
        ev.two_sweep = False
    ref = _refs(kernels, t, y, yerr, mean=1.5)
    for _ in range(2):                      # (the second evaluation runs at the calibrated generator period)
        ll = ev.evaluate()
        assert ev.engine._tp_used == (route != "streamed")
        assert ev.engine._two_sweep_used == (route == "time-parallel")
        assert np.max(np.abs(ll - ref) / np.abs(ref)) <= RTOL_LL, (route, ll, ref)
    # the condition estimate reads the REAL rows' diagonal, not the missing-data rows'
    cond, _ = ev.calibrate()
    assert 1.0 < cond < 1e6
    # one-shot form
    ll1 = gadfly_amd.log_likelihood_batch(kernels, t, y, yerr=yerr, mean=1.5)
    assert np.max(np.abs(ll1 - ref) / np.abs(ref)) <= RTOL_LL


def test_ragged_wide_kernels_scalar_error_and_a_failing_series(hip):
    import gadfly_amd
    lengths = [20000, 17001, 18500]
    kernels, t, y, _ = _series(40, lengths, seed=3, cadence=60.0)
    yerr = [np.full(n, 30.0) for n in lengths]
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)        # one error bar for all
    ev.engine.wide_tp_min_rows = 4096
    ll = ev.evaluate()
    assert ev.engine._last_wide_tp
    ref = _refs(kernels, t, y, yerr)
    assert np.max(np.abs(ll - ref) / np.abs(ref)) <= RTOL_LL
    ev.engine.force_streaming = True
    ll2 = ev.evaluate()
    assert not ev.engine._last_wide_tp and np.max(np.abs(ll2 - ref) / np.abs(ref)) <= RTOL_LL
    # per-series diagonals; the middle series is not positive definite from one of its OWN rows on: -inf with that
    # row, the others untouched
    diag = [np.full(n, 900.0) for n in lengths]
    diag[1][9000:] = -2.0 * kernels[1].get_value(np.zeros(1))[0]
    evb = gadfly_amd.BatchedLogLikelihood(kernels, t, y, diag=diag)
    evb.engine.wide_tp_min_rows = 4096
    got = evb.evaluate()
    assert got[1] == -np.inf and int(evb.engine.info[1]) == 9001
    assert np.max(np.abs(got[[0, 2]] - ref[[0, 2]]) / np.abs(ref[[0, 2]])) <= RTOL_LL


def _quarters(rng, nq=3, n=3000, cadence_d=58.85 / 86400.0, level=5.0e4):
    """`nq` quarters of one star in e-/s: holes inside each, a gap of a few hundred cadences between them, a slow
    instrumental trend and its own flux level per quarter."""
    out, t0 = [], 120.0
    truth = np.cumsum(rng.normal(size=nq * (n + 400))) * 3e-6
    for q in range(nq):
        idx = np.arange(n)
        keep = np.ones(n, bool)
        keep[rng.integers(1, n - 1, n // 25)] = False
        keep[n // 2: n // 2 + 17] = False
        t = t0 + idx * cadence_d
        x = (idx - n / 2) / n
        trend = level * (1.0 + 0.3 * q) * (1.0 + 2e-3 * x - 1e-3 * x * x)
        f = trend * (1.0 + truth[q * (n + 400): q * (n + 400) + n] + 40e-6 * rng.normal(size=n))
        out.append((t[keep], f[keep]))
        t0 = t[-1] + (250 + 13 * q) * cadence_d
    return out


def test_multi_quarter_ingestion_then_a_ragged_batch(hip):
    import gadfly_amd
    from oracle import cref, interp_ref
    rng = np.random.default_rng(12)
    stars = [_quarters(rng, nq=3, n=3000), _quarters(rng, nq=2, n=4100), _quarters(rng, nq=1, n=5000)]
    ts, ys = [], []
    for quarters in stars:
        t, f, cad = gadfly_amd.stitch_quarters(quarters, detrend_poly_order=3)
        t_ref, f_ref, cad_ref = interp_ref.stitch_quarters(quarters, detrend_poly_order=3)
        assert np.array_equal(t, t_ref) and np.array_equal(f, f_ref) and cad == cad_ref
        assert np.all(np.diff(t) > 0) and abs(np.median(f)) < 50.0
        # every cadence of the span is there after the second gap filling
        assert len(t) == int(np.rint((t[-1] - t[0]) / cad)) + 1
        ts.append(t * 0.0864)               # days -> 1e6 s (the unit GaussianProcess works in)
        ys.append(f)
    assert len({len(t) for t in ts}) == 3
    from gadfly_amd.synth import solar_like_hyperparameters
    kernels = [gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(12), texp=58.85) for _ in stars]
    ll = gadfly_amd.log_likelihood_batch(kernels, ts, ys, yerr=40.0)
    for i, k in enumerate(kernels):
        co = k.get_device_coefficients()
        ref, info = cref.loglike(co[:6], ts[i], np.full(len(ts[i]), 1600.0) + co[6], ys[i])
        assert info == 0 and abs(ll[i] - ref) <= RTOL_LL * abs(ref), (i, ll[i], ref)
    # fluxes that are in ppm already skip the normalisation (psd.py:511-512)
    q = [(t, 1e6 * (f / np.median(f) - 1.0)) for t, f in stars[1]]
    t1, f1, _ = gadfly_amd.stitch_quarters(q, in_ppm=True)
    t2, f2, _ = interp_ref.stitch_quarters(q, in_ppm=True)
    assert np.array_equal(t1, t2) and np.array_equal(f1, f2)


def test_ragged_input_errors(hip):
    import gadfly_amd
    kernels, t, y, yerr = _series(6, [700, 900])
    with pytest.raises(ValueError, match="dimension mismatch"):
        gadfly_amd.BatchedLogLikelihood(kernels, t, [y[0], y[1][:-1]], yerr=yerr)
    with pytest.raises(ValueError, match="dimension mismatch"):
        gadfly_amd.BatchedLogLikelihood(kernels[:1], t, y, yerr=yerr)
    with pytest.raises(ValueError, match="sorted"):
        gadfly_amd.BatchedLogLikelihood(kernels, [t[0][::-1], t[1]], y, yerr=yerr)
    with pytest.raises(ValueError, match="only one"):
        gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr, diag=yerr)
