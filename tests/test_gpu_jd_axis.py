"""
JD-based time axes: the reference turns an astropy ``Time`` into ``jd * day`` and then into 1/uHz
(/root/reference/gadfly/gp.py:79-80), so ``GaussianProcess(kernel, light_curve=lc)`` (docs/gadfly/synth.rst:73-91,
:150-155) and the runtime-speed notebook's ``Time(0, format='bkjd')`` axis (notebooks/paper/runtime-speed.ipynb:40)
hand the GP times around 2.12e5 (units of 1e6 s), where the solar p-modes reach phases d t of 4-5e9 rad.
Round 3's in-kernel sincos stopped at 3e9 and every fast path switched off there; it now holds to 1e12 (64-bit
quadrant, oracle/fastmath_check.c).  And celerite2 takes cos / sin of the ROUNDED product theta = fl(d t) -- half an
ulp of the phase, 5e-7 rad at 5e9 -- so the generator's rotation steps between anchors take their angle from the
difference of those rounded products (RowGen::qmode): every generator period reproduces celerite2's rows there.

Every route -- streamed fused sweep, three-sweep and two-sweep time-parallel evaluation, the stored scaled factor,
the wide kernels (W = 80 and the 86-term solar kernel's W = 172: streamed, time-parallel, `WideFactor`), the
batched evaluator and the drop-in class fed a light curve with a ``Time`` axis -- against the oracle's C
restatement on t = 2.12e5 + n 60e-6, log-likelihood at 1e-8, vectors at 1e-6.
"""
import types

import numpy as np
import pytest

from tests import util
from tests import fake_units

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8
TOL_VEC = 1e-6
BKJD0 = 2454833.0 * 0.0864          # Time(0, format='bkjd') in units of 1e6 s: 2.12e5


def _relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _jd_problem(J, N, **kw):
    prob = util.solar_problem(J, N, **kw)
    prob["t"] = BKJD0 + prob["t"]
    return prob


def _ref(prob, y=None):
    from oracle import cref
    co = prob["kernel"].get_device_coefficients()
    ref, info = cref.loglike(co[:6], prob["t"], prob["diag_user"] + co[6], prob["y"] if y is None else y)
    assert info == 0
    return ref


@pytest.mark.parametrize("J,N,kw", [(30, 6000, dict()), (30, 5000, dict(yerr=0.0)), (20, 4000, dict(gaps=True)),
                                    (6, 3000, dict(jitter_t=True))],
                         ids=["J30", "J30-yerr0", "J20-gaps", "J6-jitter"])
def test_fused_routes_on_a_jd_axis(hip, J, N, kw):
    import torch
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref, seq
    prob = _jd_problem(J, N, **kw)
    t, y = prob["t"], prob["y"]
    co = prob["kernel"].get_device_coefficients()
    ref = _ref(prob)
    eng = StreamingBatch([co], t, y, diag=prob["diag_user"], tile_rows=1024)
    assert eng._pack[6] * eng._tmax > 3.5e9 and eng._fused_ok()        # beyond round 3's range, still fused
    assert eng.generator_period == 4                                    # the default: rotation steps are allowed
    cond = None
    for period in (4, 1, 64):
        # (period 64 where the conditioning allows it by the product's own rule -- yerr = 0 does not)
        if period == 64 and eng.period_for_condition(cond) < 64:
            continue
        eng.generator_period = period
        ll = float(eng.log_likelihood()[0])
        cond = eng.condition_estimate() if cond is None else cond
        assert eng.kernel_used == "fused" and abs(ll - ref) <= RTOL_LL * abs(ref), (period, ll, ref)
        for two in (False, True):
            eng.two_sweep = two
            ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=512)[0])
            assert eng._two_sweep_used == two and abs(ll_tp - ref) <= RTOL_LL * abs(ref), (period, two, ll_tp, ref)
        eng.two_sweep = False
    eng.generator_period = 1
    # the stored scaled factor: solves and draws
    fac = eng.stored_factor(chunk_len=512)
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    Y = np.random.default_rng(J).normal(size=(len(t), 3))
    Yd = torch.as_tensor(Y).cuda().reshape(1, len(t), 3)
    ref_ai = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    assert _relmax(fac.apply_inverse(Yd)[0].cpu().numpy(), ref_ai) < TOL_VEC
    ref_dt = cref.matmul_lower(t, c, U, W_ref, Y * np.sqrt(d_ref)[:, None])
    assert _relmax(fac.dot_tril(Yd)[0].cpu().numpy(), ref_dt) < TOL_VEC


@pytest.mark.parametrize("J,N", [(40, 5000), (86, 4200)], ids=["W80", "W172-solar"])
def test_wide_routes_on_a_jd_axis(hip, J, N):
    from gadfly_amd.engine import StreamingBatch
    prob = _jd_problem(J, N)
    t, y = prob["t"], prob["y"]
    co = prob["kernel"].get_device_coefficients()
    ref = _ref(prob)
    eng = StreamingBatch([co, co], t, y, diag=prob["diag_user"], tile_rows=1024)
    assert eng._pack[6] * eng._tmax > 3.5e9 and eng._wide_ok() and not eng._fused_ok()
    for period in (1, 64):
        eng.generator_period = period
        ll = eng.log_likelihood().cpu().numpy()
        assert eng.kernel_used == "fused-wide"
        assert np.max(np.abs(ll - ref)) <= RTOL_LL * abs(ref), (period, ll, ref)
        for two in (False, True):
            eng.two_sweep = two
            ll_tp = eng.log_likelihood_time_parallel(chunk_len=640).cpu().numpy()
            assert eng._last_wide_tp and eng._two_sweep_used == two
            assert np.max(np.abs(ll_tp - ref)) <= RTOL_LL * abs(ref), (period, two, ll_tp, ref)
        eng.two_sweep = False


def test_batched_evaluator_lengthens_the_generator_period_on_a_jd_axis(hip):
    """BatchedLogLikelihood calibrates the generator period from the measured conditioning.  On a JD axis the
    rotation steps follow celerite2's rounded phases (RowGen::qmode), so the calibrated period may lengthen there as
    it does on the same series moved to t = 0 -- and the values stay within 1e-8 of the oracle (before qmode a
    period of 64 was 5e-9 ... 1e-7 off at this phase range: profiles/r04_phase_quantum.txt)."""
    import gadfly_amd
    from gadfly_amd.synth import jitter_hyperparameters, solar_like_hyperparameters
    N, B = 20_000, 5
    prob = _jd_problem(30, N)
    hps = [jitter_hyperparameters(solar_like_hyperparameters(30), seed=1000 + i) for i in range(B)]
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=60.0) for hp in hps]
    refs = [_ref(dict(prob, kernel=k)) for k in kernels]
    for t in (prob["t"], prob["t"] - BKJD0):
        ev = gadfly_amd.BatchedLogLikelihood(kernels, t, prob["y"], yerr=30.0)
        periods = []
        for _ in range(3):                              # the later evaluations run at the calibrated period
            periods.append(int(ev.engine.generator_period))
            ll = ev.evaluate()
            assert ev.engine._fused_ok() and ev.engine._tp_used          # time-parallel, fused
            if t is prob["t"]:
                for i in range(B):
                    assert abs(ll[i] - refs[i]) <= RTOL_LL * abs(refs[i]), (periods, i, ll[i], refs[i])
        assert periods[0] == 4 and periods[-1] == 64, periods


@pytest.mark.parametrize("J,N", [(30, 12_000), (86, 18_000)], ids=["J30", "solar-W172"])
def test_gaussian_process_from_a_light_curve_with_a_time_axis(hip, monkeypatch, J, N):
    """The documented entry: ``GaussianProcess(kernel, light_curve=lc)`` with ``lc.time`` an astropy ``Time``
    (here the stand-in carrying ``.jd``), fluxes in e-/s (docs/gadfly/synth.rst:73-91); then ``predict`` at gap
    times given as ``Time(310, format='bkjd') + ... * u.d`` (synth.rst:193-201) and a draw."""
    import gadfly_amd
    from gadfly_amd import units as gunits
    from gadfly_amd.engine import ScaledFactor, WideFactor
    from oracle import cref, seq
    u = fake_units.patch(monkeypatch, gunits)
    prob = util.solar_problem(J, N)
    k = prob["kernel"]
    rng = np.random.default_rng(5)
    days = 310.0 + np.arange(N) / 1440.0                              # BKJD, one-minute cadence
    med = 5.0e4
    y_ppm = 40.0 * rng.normal(size=N) + 3.0 * np.cumsum(rng.normal(size=N))
    y_ppm -= np.median(y_ppm)                                           # (so that the cached median is `med`)
    lc = types.SimpleNamespace(time=fake_units.FakeTime(days, format="bkjd"),
                               flux=fake_units.Q(med * (1.0 + 1e-6 * y_ppm), u.electron / u.s),
                               flux_err=fake_units.Q(np.full(N, 30e-6 * med), u.electron / u.s))
    gp = gadfly_amd.GaussianProcess(k, light_curve=lc)
    t = (days + 2454833.0) * 0.0864
    np.testing.assert_allclose(gp._t, t, rtol=1e-15)
    assert gp._t[0] > 2.1e5
    assert isinstance(gp._factor, WideFactor if J > 31 else ScaledFactor)      # the fast routes, not k_build + k_factor
    if J > 31:
        assert gp._factor.time_parallel and gp._factor.nch > 1
    t = gp._t
    diag = gp._diag
    np.testing.assert_allclose(diag, 900.0, rtol=1e-12)
    y = gp._flux_to_ppm(lc.flux)
    co = k.get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, diag + co[6], y)
    assert info == 0
    ll = gp.log_likelihood(lc.flux)
    assert abs(ll - ref) <= RTOL_LL * abs(ref), (ll, ref)
    c, a, U, V = seq.celerite_matrices(co[:6], t, diag + co[6])
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y) / d_ref)
    assert _relmax(gp.predict(lc.flux), y - diag * alpha) < TOL_VEC
    gap = fake_units.FakeTime(310.0, format="bkjd") + fake_units.Q(np.linspace(0.5, N / 1440.0 - 0.5, 50), u.day)
    ts = gp._time_to_freq(gap)
    _, _, Us, Vs = seq.celerite_matrices(co[:6], ts, 0.0)
    mu_ref = cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha)
    mu = gp.predict(lc.flux, t=gap)
    assert _relmax(mu, mu_ref) < TOL_VEC
    flux = gp.predict(lc.flux, t=gap, return_quantity=True)
    np.testing.assert_allclose(flux.value, (1e-6 * mu_ref + 1.0) * med, rtol=1e-9)
    # a draw with the reference's RNG contract (np.random.seed(42), celerite2 draws randn(N))
    np.random.seed(42)
    draw = gp.sample()
    np.random.seed(42)
    n = np.random.randn(N)
    want = cref.matmul_lower(t, c, U, W_ref, (n * np.sqrt(d_ref))[:, None])[:, 0]
    want = want - want.mean()
    assert _relmax(draw, want) < TOL_VEC
