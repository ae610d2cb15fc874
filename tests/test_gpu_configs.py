"""
GPU parity at the BATCHED shapes of BASELINE.json's configs 3, 4 and 5, against the
oracle's C restatement (oracle/cref.py) on identical inputs:

  cfg3  batch of light curves, J = 20 (W = 40): own t, y and kernel per problem -> k_factor7<40> with
        per-problem t / y / diag strides (streamed) and the time-parallel route small batches take;
  cfg4  MCMC walkers, J = 40 (W = 80): shared t, y, own hyperparameters -> the wide kernels
        (fused wide sweep / k_build2 + k_factor2w / k_build + k_factor) with B > 1;
  cfg5  dot_tril of 64 normal vectors, J = 30 (W = 60): multi-right-hand-side chunk sweeps.

Small sizes compare EVERY entry; the full sizes (256 x 65 000, 512 x 200 000, 64 x 500 000) run once
each and compare sampled entries / columns (an oracle evaluation costs about a second there).
Bars: log-likelihood 1e-8 relative (north_star), draws 1e-6.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8
TOL_VEC = 1e-6


def _relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _cfg3_problem(B, N):
    import gadfly_amd
    from gadfly_amd.synth import cfg3_light_curves
    hps, t, y, yerr, texp = cfg3_light_curves(B, N)
    return [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps], t, y, yerr


def _cfg4_problem(B, N):
    import gadfly_amd
    from gadfly_amd.synth import cfg4_walkers
    hps, t, y, texp = cfg4_walkers(B, N)
    return [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps], t, y


def _ref_ll(kernel, t, diag, y):
    from oracle import cref
    co = kernel.get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, diag + co[6], y)
    assert info == 0
    return ref


# ---------------------------------------------------------------------------------------------
# cfg3 shape: B > 1 at W = 40 with own t, y, diag per problem
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("route", ["streamed", "chunked"])
def test_cfg3_shape_every_entry(hip, route):
    import gadfly_amd
    B, N = 9, 2100
    kernels, t, y, yerr = _cfg3_problem(B, N)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr, tile_rows=512)
    eng = ev.engine
    assert eng.W == 40 and eng._fused_ok() and eng.t.shape[0] == B and eng.diag.shape[0] == B
    eng.generator_period = 1
    if route == "streamed":
        ll = eng.log_likelihood().cpu().numpy()            # 5 tiles with state hand-off
    else:
        ll = eng.log_likelihood_time_parallel(chunk_len=256).cpu().numpy()
        assert eng._tp_used
    for i, k in enumerate(kernels):
        ref = _ref_ll(k, t[i], yerr[i] ** 2, y[i])
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref), (route, i, ll[i], ref)


def test_cfg3_full_size_sampled(hip):
    """256 light curves x N = 65 000, J = 20, through the route the product picks (B <= 256 and long
    series: time-parallel) AND the streamed sweep; entries 0, 100 and 255 against the oracle, all
    256 between the two routes."""
    import gadfly_amd
    B, N = 256, 65_000
    kernels, t, y, yerr = _cfg3_problem(B, N)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr)
    ll_tp = ev.evaluate()                                  # auto route + generator calibration
    assert ev.engine._tp_used
    ev.engine.force_streaming = True
    ll_st = ev.evaluate()
    assert not ev.engine._tp_used
    assert np.all(np.isfinite(ll_tp)) and np.all(np.isfinite(ll_st))
    assert np.max(np.abs(ll_tp - ll_st) / np.abs(ll_st)) <= RTOL_LL
    for i in (0, 100, 255):
        ref = _ref_ll(kernels[i], t[i], yerr[i] ** 2, y[i])
        assert abs(ll_tp[i] - ref) <= RTOL_LL * abs(ref), (i, ll_tp[i], ref)
        assert abs(ll_st[i] - ref) <= RTOL_LL * abs(ref), (i, ll_st[i], ref)


# ---------------------------------------------------------------------------------------------
# cfg4 shape: B > 1 walkers at W = 80 on the wide kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["auto", "scaled_wide", "v1"])
def test_cfg4_shape_every_entry(hip, path):
    from gadfly_amd.engine import StreamingBatch
    B, N = 9, 1700
    kernels, t, y = _cfg4_problem(B, N)
    diag = np.full(N, 900.0)
    eng = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=diag,
                         tile_rows=512, force_v1=(path == "v1"),
                         allow_fused=(path == "auto"))
    assert eng.W == 80 and eng.t.shape[0] == 1
    if path == "scaled_wide":
        assert eng.scaled_wide and not eng._fused_ok()
    if path == "v1":
        assert not eng.scaled_wide and not eng.scaled
    eng.generator_period = 1
    ll = eng.log_likelihood().cpu().numpy()
    for i, k in enumerate(kernels):
        ref = _ref_ll(k, t, diag, y)
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref), (path, i, ll[i], ref)
    # second evaluation with fresh coefficient packs (what a sampler does per step)
    kernels2, _, _ = _cfg4_problem(B, N)
    pack = eng.pack_coefficients([k.get_device_coefficients() for k in kernels2[::-1]])
    eng.use_coefficients(pack)
    ll2 = eng.log_likelihood().cpu().numpy()
    assert np.max(np.abs(ll2 - ll[::-1]) / np.abs(ll[::-1])) <= 1e-12


@pytest.mark.parametrize("N,L", [(3000, 256), (2500, 640)], ids=["12chunks", "4chunks-ragged"])
def test_cfg4_shape_time_parallel_every_entry(hip, N, L):
    """B = 9 walkers at W = 80, CHUNKED in time (what a GPU's shard of cfg4 runs: 64 walkers cannot fill
    the chip, so every walker's series is cut into chunks swept concurrently and stitched exactly by the
    batched dense combine, gf_wide_combine): every entry against the oracle, and against the sequential
    wide sweep of the same engine."""
    from gadfly_amd.engine import StreamingBatch
    B = 9
    kernels, t, y = _cfg4_problem(B, N)
    diag = np.full(N, 900.0)
    eng = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=diag)
    assert eng.W == 80 and eng._wide_ok() and not eng._fused_ok()
    eng.generator_period = 1
    ll_seq = eng.log_likelihood().cpu().numpy()
    ll = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    assert eng._last_wide_tp and eng._wide_tp["nch"] == -(-N // L)
    d_tp = eng._wide_tp["d"][:B * N].view(B, N).cpu().numpy()
    for i, k in enumerate(kernels):
        ref = _ref_ll(k, t, diag, y)
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref), (i, ll[i], ref)
        assert abs(ll[i] - ll_seq[i]) <= 1e-10 * abs(ref), (i, ll[i], ll_seq[i])
    # every pivot of two walkers against the oracle's factor
    from oracle import cref, seq
    for i in (0, B - 1):
        prob = dict(kernel=kernels[i], t=t, diag_user=diag)
        c, a, U, V = util.oracle_matrices(prob, seq)
        d_ref, _, _ = cref.factor(t, c, a, U, V)
        assert _relmax(d_tp[i], d_ref) < 1e-9
    # the route the product picks for such a shard (long series, few walkers)
    assert not eng._wide_tp_ok()                       # (too short here: the sequential sweep is used)
    eng.wide_tp_min_rows = 1024
    assert eng._wide_tp_ok()
    out, _ = eng.evaluate()
    assert eng._last_wide_tp and eng._wide_tp["nch"] > 1
    assert np.max(np.abs(out.cpu().numpy() - ll) / np.abs(ll)) <= 1e-10
    # one failing walker (negative diagonal from the middle on): -inf and the first failing row, the others untouched
    bad = np.tile(diag, (B, 1))
    bad[4, N // 2:] = -2.0 * kernels[4].get_value(np.zeros(1))[0]
    engb = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=bad)
    engb.generator_period = 1
    llb = engb.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    assert llb[4] == -np.inf and int(engb.info[4]) == N // 2 + 1
    ok = np.arange(B) != 4
    assert np.max(np.abs(llb[ok] - ll[ok]) / np.abs(ll[ok])) <= 1e-10


@pytest.mark.parametrize("shape", ["cfg3", "cfg4"])
def test_two_sweep_log_likelihood(hip, shape):
    """Time-parallel log-likelihood WITHOUT a final pass (engine.two_sweep): nominal sums + per-chunk corrections
    log det(I - X G) and e^T G v - 2 m^T e - m^T X m from the start states.  Every entry against the oracle at
    1e-8 and against the three-sweep evaluation of the same engine at 1e-11, at cfg3's shape (W = 40, own t / y /
    yerr per problem, sequential and tree combine) and cfg4's (W = 80, the dense combine)."""
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    B = 9
    if shape == "cfg3":
        N = 2600
        kernels, t, y, yerr = _cfg3_problem(B, N)
        diag = yerr ** 2
        chunkings = (256, 64)               # 11 chunks: sequential combine; 41 chunks: tree combine
    else:
        N = 3000
        kernels, t, y = _cfg4_problem(B, N)
        diag = np.full(N, 900.0)
        chunkings = (256, 640)
    eng = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=diag)
    eng.generator_period = 1
    refs = [_ref_ll(k, t[i] if shape == "cfg3" else t, diag[i] if shape == "cfg3" else diag,
                    y[i] if shape == "cfg3" else y) for i, k in enumerate(kernels)]
    for L in chunkings:
        eng.two_sweep = False
        ll3 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
        assert not eng._two_sweep_used
        eng.two_sweep = True
        ll2 = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
        assert eng._two_sweep_used
        for i, ref in enumerate(refs):
            assert abs(ll2[i] - ref) <= RTOL_LL * abs(ref), (shape, L, i, ll2[i], ref)
            assert abs(ll2[i] - ll3[i]) <= 1e-11 * abs(ref), (shape, L, i, ll2[i], ll3[i])
    # a matrix that is not positive definite: the two-sweep value is not finite (never a wrong finite number
    # here: the nominal pass meets the negative diagonal itself), and BatchedLogLikelihood does not take the
    # route at all for a negative diagonal
    bad = np.array(np.broadcast_to(diag, (B, N)))
    bad[4, N // 2:] = -2.0 * kernels[4].get_value(np.zeros(1))[0]
    engb = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=bad)
    engb.generator_period, engb.two_sweep = 1, True
    llb = engb.log_likelihood_time_parallel(chunk_len=chunkings[0]).cpu().numpy()
    assert not np.isfinite(llb[4]) and np.all(np.isfinite(np.delete(llb, 4)))
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, diag=bad)
    ev.engine.wide_tp_min_rows = 1024
    got = ev.evaluate()
    assert not ev.engine.two_sweep and got[4] == -np.inf and int(ev.engine.info[4]) == N // 2 + 1


def test_two_sweep_is_the_batched_default_and_resolves_broken_pivots(hip):
    """BatchedLogLikelihood takes the two-sweep route for SHO kernels with a non-negative diagonal; an entry
    whose value comes out non-finite is repeated with the final pass by resolve()."""
    import gadfly_amd
    B, N = 6, 20_000
    kernels, t, y = _cfg4_problem(B, N)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
    ll = ev.evaluate()
    eng = ev.engine
    assert eng.two_sweep and eng._two_sweep_used and eng._last_wide_tp
    for i in (0, B - 1):
        ref = _ref_ll(kernels[i], t, np.full(N, 900.0), y)
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref)
    # poison one entry as a broken two-sweep evaluation would leave it: resolve() must repair exactly that one
    out = ev.evaluate_device()
    keep = out.clone()
    out[2] = float("nan")
    ev._unresolved[-1] = (out,) + (ev._unresolved[-1][1] | ~out.isfinite(),) + ev._unresolved[-1][2:] \
        if ev._unresolved else None
    if not ev._unresolved or ev._unresolved[-1] is None:
        ev._unresolved = [(out, ~out.isfinite(), eng._pack, int(eng.generator_period))]
    assert ev.resolve() >= 1
    got = out.cpu().numpy()
    assert np.all(np.isfinite(got)) and abs(got[2] - keep.cpu().numpy()[2]) <= 1e-10 * abs(got[2])
    assert eng.two_sweep                                # the setting survives the repair


def test_evaluate_picks_the_wide_time_parallel_route(hip):
    """ONE long series with a wide kernel through evaluate() / BatchedLogLikelihood (no route argument):
    the exact time-parallel evaluation must be the route taken, and the condition estimate afterwards
    must read that run's accumulators."""
    import gadfly_amd
    N = 20_000
    kernels, t, y = _cfg4_problem(1, N)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
    ll = ev.evaluate()
    eng = ev.engine
    assert eng._last_wide_tp and eng._wide_tp["nch"] > 1
    ref = _ref_ll(kernels[0], t, np.full(N, 900.0), y)
    assert abs(ll[0] - ref) <= RTOL_LL * abs(ref)
    cond = eng.condition_estimate()
    assert np.isfinite(cond) and cond > 1.0
    eng.force_streaming = True                          # the streamed sweep on request
    ll2 = ev.evaluate()
    assert not eng._last_wide_tp and abs(ll2[0] - ref) <= RTOL_LL * abs(ref)


def test_cfg4_full_size_sampled(hip):
    """512 walkers x N = 200 000, J = 40 (W = 80), shared t, y: walkers 0 and 511 against the
    oracle (about 2 s of host time each), all finite."""
    import gadfly_amd
    B, N = 512, 200_000
    kernels, t, y = _cfg4_problem(B, N)
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
    ll = ev.evaluate()
    assert ll.shape == (B,) and np.all(np.isfinite(ll))
    for i in (0, 511):
        ref = _ref_ll(kernels[i], t, np.full(N, 900.0), y)
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref), (i, ll[i], ref)


# ---------------------------------------------------------------------------------------------
# cfg5: dot_tril of 64 draws at N = 5e5, J = 30
# ---------------------------------------------------------------------------------------------
def test_cfg5_full_size_sampled(hip):
    """GaussianProcess.dot_tril on (5e5, 64) normal vectors: columns 0, 31 and 63 against the
    oracle's matmul_lower on its own factor; then sample(size=64) with the same vectors through
    numpy's legacy global RNG reproduces gadfly's mean-subtraction quirk (gp.py:392)."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    from oracle import cref, seq
    N, J, R = 500_000, 30, 64
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
    t = uniform_times(N, 60.0)
    gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
    np.random.seed(42)
    n = np.random.randn(N, R)
    Z = gp.dot_tril(n)
    assert Z.shape == (N, R) and np.all(np.isfinite(Z))
    prob = dict(kernel=k, t=t, diag_user=np.full(N, 900.0))
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    cols = [0, 31, 63]
    ref = cref.matmul_lower(t, c, U, W_ref, n[:, cols] * np.sqrt(d_ref)[:, None])
    assert _relmax(Z[:, cols], ref) < TOL_VEC
    np.random.seed(42)
    draws = gp.sample(size=R)
    assert draws.shape == (R, N)
    want = Z.T - Z.T.mean(axis=0)
    assert _relmax(draws, want) < 1e-12


# ---------------------------------------------------------------------------------------------
# the drop-in class on wide kernels (W = 80, and the solar kernel's W = 172): fused wide sweep with
# the factor stored in scaled form, general-width solves on it
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("J,N,kw", [(40, 2600, dict(gaps=True)), (86, 1500, dict()),
                                    (33, 900, dict(jitter_t=True)), (64, 1100, dict(yerr=0.0))],
                         ids=["W80-gaps", "W172", "W66-jitter", "W128-yerr0"])
def test_gaussian_process_wide_kernel(hip, J, N, kw):
    import gadfly_amd
    from gadfly_amd.engine import WideFactor
    from oracle import cref, seq
    prob = util.solar_problem(J, N, **kw)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    n = len(t)
    gp = gadfly_amd.GaussianProcess(k, t=t, diag=prob["diag_user"], mean=0.5)
    assert isinstance(gp._factor, WideFactor) and gp._fast is None
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    assert abs(gp._log_det - np.sum(np.log(d_ref))) <= 1e-10 * abs(gp._log_det)
    assert _relmax(gp._factor.d[0].cpu().numpy(), d_ref) < 1e-9
    co = k.get_device_coefficients()
    ref, _ = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y - 0.5)
    assert abs(gp.log_likelihood(y) - ref) <= RTOL_LL * abs(ref)
    rng = np.random.default_rng(J)
    for R in (1, 3):
        Y = rng.normal(size=(n, R)) if R > 1 else rng.normal(size=n)
        Y2 = Y.reshape(n, -1)
        ref_ai = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y2) / d_ref[:, None])
        assert _relmax(gp.apply_inverse(Y).reshape(n, -1), ref_ai) < TOL_VEC
        ref_dt = cref.matmul_lower(t, c, U, W_ref, Y2 * np.sqrt(d_ref)[:, None])
        assert _relmax(gp.dot_tril(Y).reshape(n, -1), ref_dt) < TOL_VEC
    # conditional mean at the observed and at new times
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y - 0.5) / d_ref)
    assert _relmax(gp.predict(y), y - prob["diag_user"] * alpha) < TOL_VEC
    ts = np.sort(rng.uniform(t[0], t[-1], 40))
    _, _, Us, Vs = seq.celerite_matrices(co[:6], ts, 0.0)
    mu_ref = cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha) + 0.5
    assert _relmax(gp.predict(y, t=ts), mu_ref) < TOL_VEC
    # conditional variance at a few new times (multi-right-hand-side sweeps on the wide factor) against the
    # dense formulation k(0) - K*^T K^-1 K*
    if n <= 1600:
        from oracle import dense
        tv = ts[::8]
        _, var = gp.predict(y, t=tv, return_var=True)
        K = dense.dense_K(co[:6], t, prob["diag_user"] + co[6])
        Ks = k.get_value(t[:, None] - tv[None, :])
        var_ref = k.get_value(np.zeros(1))[0] - np.sum(Ks * np.linalg.solve(K, Ks), axis=0)
        assert _relmax(var, var_ref) < 1e-5
    # a failing matrix is reported at its first non-positive pivot
    bad = prob["diag_user"].copy()
    bad[n // 2:] = -2.0 * k.get_value(np.zeros(1))[0]
    with pytest.raises(gadfly_amd.LinAlgError, match=f"pivot {n // 2 + 1} "):
        gp.compute(t, diag=bad)
    gp.compute(t, diag=bad, quiet=True)
    assert gp.log_likelihood(y) == -np.inf


@pytest.mark.parametrize("J,N,tile,kw", [(40, 2600, 512, dict()), (33, 1900, 8192, dict(jitter_t=True)),
                                          (64, 2100, 640, dict(gaps=True)), (86, 1500, 256, dict(yerr=0.0)),
                                          (50, 3000, 1024, dict(gaps=True)), (72, 1200, 8192, dict())],
                         ids=["W80", "W66-jitter", "W128-gaps", "W172-yerr0", "W100-gaps", "W144"])
def test_wide_streamed_log_likelihood(hip, J, N, tile, kw):
    """The streamed log-likelihood of wide kernels (k_factorw without row stores) against the oracle at six widths,
    over several tiles (state hand-off through the slots), with gaps and jittered stamps, every pivot / z row of
    a one-tile run, and a failing pivot.  (Round 3's archived blocked sweep, tools/archive, was developed
    against this test.)"""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref, seq
    prob = util.solar_problem(J, N, **kw)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    eng = StreamingBatch([co, co], t, y, diag=prob["diag_user"], tile_rows=tile)
    assert eng._wide_ok() and not eng._fused_ok()
    eng.force_streaming = True
    ll = eng.log_likelihood().cpu().numpy()
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0 and np.all(np.abs(ll - ref) <= RTOL_LL * abs(ref)), (ll, ref)
    if tile >= len(t):                                  # one tile: the rows d, z of the whole series are there
        c, a, U, V = util.oracle_matrices(prob, seq)
        d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
        z_ref = cref.solve_lower(t, c, U, W_ref, y)
        n = len(t)
        assert _relmax(eng.d[0, :n].cpu().numpy(), d_ref) < 1e-9
        assert _relmax(eng.z[1, :n].cpu().numpy(), z_ref) < 1e-8
    # a matrix that is not positive definite stops at the oracle's row
    bad = prob["diag_user"].copy()
    bad[len(t) // 2:] = -2.0 * k.get_value(np.zeros(1))[0]
    engb = StreamingBatch([co], t, y, diag=bad, tile_rows=tile)
    engb.force_streaming = True
    llb = engb.log_likelihood()
    assert float(llb[0]) == float("-inf") and int(engb.info[0]) == len(t) // 2 + 1


def test_wide_kernel_with_real_terms(hip):
    """A WIDE kernel with overdamped terms (Q < 1/2: two real exponentials each; W = 6 + 66 = 72): the engine
    writes its real terms as degenerate complex ones (engine._complexify_pack) so that it rides on the fused wide
    sweep and its stored factor -- through the drop-in class and the batched evaluator (streamed and
    time-parallel), against the oracle on the ORIGINAL term structure."""
    import gadfly_amd
    from gadfly_amd.engine import WideFactor, StreamingBatch
    from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
    from gadfly_amd.synth import solar_like_hyperparameters
    from oracle import cref, seq
    prob = util.solar_problem(33, 2400, gaps=True)
    hp = solar_like_hyperparameters(33)
    # w0 in rad / (1e6 s): slow, overdamped components next to the solar-like terms
    terms = [SHOTerm(S0=4.0e3, w0=60.0, Q=0.3), SHOTerm(S0=9.0e2, w0=400.0, Q=0.45), SHOTerm(S0=2.0e2, w0=2500.0, Q=0.2)]
    terms += [SHOTerm(S0=float(h["hyperparameters"]["S0"]), w0=float(h["hyperparameters"]["w0"]),
                      Q=float(h["hyperparameters"]["Q"])) for h in hp]
    k = TermConvolution(TermSum(*terms), 60.0e-6)
    co = k.get_device_coefficients()
    assert len(co[0]) == 6 and len(co[0]) + 2 * len(co[2]) == 72
    t, y, du = prob["t"], prob["y"], prob["diag_user"]
    n = len(t)
    ref, info = cref.loglike(co[:6], t, du + co[6], y)
    assert info == 0
    eng = StreamingBatch([co, co], t, y, diag=du)
    assert eng._complexified and eng.Jr == 0 and eng.W == 78 and eng._wide_ok()
    eng.generator_period = 1
    ll = eng.log_likelihood().cpu().numpy()
    assert np.all(np.abs(ll - ref) <= RTOL_LL * abs(ref)), (ll, ref)
    for two in (False, True):
        eng.two_sweep = two
        ll_tp = eng.log_likelihood_time_parallel(chunk_len=256).cpu().numpy()
        assert np.all(np.abs(ll_tp - ref) <= RTOL_LL * abs(ref)), (two, ll_tp, ref)
    pack = eng.pack_coefficients([co, co])              # packs of the original structure are accepted
    eng.use_coefficients(pack)
    assert abs(float(eng.log_likelihood()[1]) - ref) <= RTOL_LL * abs(ref)
    # the drop-in class: stored factor, solves, conditional mean at new times
    gp = gadfly_amd.GaussianProcess(k, t=t, diag=du)
    assert isinstance(gp._factor, WideFactor)
    prob2 = dict(kernel=k, t=t, diag_user=du)
    c, a, U, V = util.oracle_matrices(prob2, seq)
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    assert abs(gp.log_likelihood(y) - ref) <= RTOL_LL * abs(ref)
    rng = np.random.default_rng(5)
    Y = rng.normal(size=(n, 3))
    ref_ai = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    assert _relmax(gp.apply_inverse(Y), ref_ai) < TOL_VEC
    ref_dt = cref.matmul_lower(t, c, U, W_ref, Y * np.sqrt(d_ref)[:, None])
    assert _relmax(gp.dot_tril(Y), ref_dt) < TOL_VEC
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y) / d_ref)
    ts = np.sort(rng.uniform(t[0], t[-1], 40))
    _, _, Us, Vs = seq.celerite_matrices(co[:6], ts, 0.0)
    assert _relmax(gp.predict(y, t=ts), cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha)) < TOL_VEC


@pytest.mark.parametrize("J,N,L,kw", [(40, 3000, 256, dict()), (86, 2600, 192, dict(gaps=True)),
                                      (33, 2100, 128, dict(jitter_t=True)), (64, 1500, 512, dict(yerr=0.0)),
                                      (86, 20000, None, dict())],
                         ids=["W80", "W172-gaps", "W66-jitter", "W128-yerr0", "W172-auto"])
def test_wide_time_parallel(hip, J, N, L, kw):
    """Exact time-parallel factorisation of ONE series with a wide kernel (nominal pass, k_phiw
    transitions, dense LFT tree combine, final pass) against the sequential wide sweep and the oracle:
    log-likelihood, every pivot d_n and z_n, and the stored factor through a solve."""
    import torch
    from gadfly_amd.engine import StreamingBatch, WideFactor
    from oracle import cref, seq
    prob = util.solar_problem(J, N, **kw)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    n = len(t)
    co = k.get_device_coefficients()
    eng = StreamingBatch([co], t, y, diag=prob["diag_user"])
    eng.generator_period = 1
    assert eng._wide_ok() and not eng._fused_ok()
    ll_seq = float(eng.log_likelihood()[0])
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0 and abs(ll_seq - ref) <= RTOL_LL * abs(ref)
    ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=L)[0])
    assert eng._wide_tp["nch"] > 1
    assert abs(ll_tp - ref) <= RTOL_LL * abs(ref), (ll_tp, ref)
    assert abs(ll_tp - ll_seq) <= 1e-10 * abs(ref)
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    z_ref = cref.solve_lower(t, c, U, W_ref, y)
    assert _relmax(eng._wide_tp["d"][:n].cpu().numpy(), d_ref) < 1e-9
    assert _relmax(eng._wide_tp["z"][:n].cpu().numpy(), z_ref) < 1e-8
    # the factor stored by the time-parallel run, through the general-width solves
    fac = WideFactor(eng, time_parallel=True, chunk_len=L)
    assert fac.time_parallel and _relmax(fac.d[0].cpu().numpy(), d_ref) < 1e-9
    Y = np.random.default_rng(J).normal(size=(n, 2))
    ref_ai = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    got = fac.apply_inverse(torch.as_tensor(Y).cuda().reshape(1, n, 2))[0].cpu().numpy()
    assert _relmax(got, ref_ai) < TOL_VEC
    # several right-hand sides are chunk-parallel too (k_solve_rhs in chunk mode, W x R states chained as
    # batched GEMMs): two tiles of right-hand sides, both sweeps
    Y70 = np.random.default_rng(J + 1).normal(size=(n, 70))
    Y70d = torch.as_tensor(Y70).cuda().reshape(1, n, 70)
    assert _relmax(fac.solve_lower(Y70d)[0].cpu().numpy(), cref.solve_lower(t, c, U, W_ref, Y70)) < TOL_VEC
    assert _relmax(fac.solve_upper(Y70d)[0].cpu().numpy(), cref.solve_upper(t, c, U, W_ref, Y70)) < TOL_VEC
    assert torch.equal(Y70d, torch.as_tensor(Y70).cuda().reshape(1, n, 70))       # inputs untouched
    # ONE right-hand side takes the chunk-parallel sweeps (gf_solve_chunk): local pass, combine through the
    # TRUE factor's chunk transitions (the solves) / the diagonal decays (dot_tril), final pass
    assert fac.nch > 1
    y1 = torch.as_tensor(Y[:, 0].copy()).cuda().reshape(1, n, 1)
    zl = cref.solve_lower(t, c, U, W_ref, Y[:, 0])
    assert _relmax(fac.solve_lower(y1)[0, :, 0].cpu().numpy(), zl) < TOL_VEC
    assert _relmax(fac.solve_upper(y1)[0, :, 0].cpu().numpy(), cref.solve_upper(t, c, U, W_ref, Y[:, 0])) < TOL_VEC
    assert _relmax(fac.apply_inverse(y1)[0, :, 0].cpu().numpy(), ref_ai[:, 0]) < TOL_VEC
    ref_dt = cref.matmul_lower(t, c, U, W_ref, Y[:, 0] * np.sqrt(d_ref))
    assert _relmax(fac.dot_tril(y1)[0, :, 0].cpu().numpy(), ref_dt) < TOL_VEC
    assert fac._Phi is not None and fac._tp_bufs is None
    # failing matrix: the first non-positive pivot, although later chunks fail as well
    bad = prob["diag_user"].copy()
    bad[n // 2:] = -2.0 * k.get_value(np.zeros(1))[0]
    engb = StreamingBatch([co], t, y, diag=bad)
    engb.generator_period = 1
    assert engb.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()[0] == -np.inf
    assert int(engb.info[0]) == n // 2 + 1
