"""
GPU parity: every C-ABI entry point and the GaussianProcess API against the oracle
(oracle/cref.py C restatement, itself pinned by tests/test_oracle.py) on identical inputs.
Bars: log-likelihood 1e-8 relative (north_star), draws 1e-6; observed ~1e-12.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8
TOL_VEC = 1e-6


def _relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


CASES = [
    ("solar", dict(J=6, N=3000)),
    ("solar", dict(J=6, N=2000, yerr=0.0)),
    ("solar", dict(J=20, N=1500, jitter_t=True)),
    ("solar", dict(J=30, N=2500, gaps=True)),
    ("solar", dict(J=40, N=1200)),
    ("solar", dict(J=86, N=600)),
    ("generic", dict(kind="sho_q100", N=512)),
    ("generic", dict(kind="overdamped", N=300)),
    ("generic", dict(kind="q_half", N=300)),
    ("generic", dict(kind="mixed", N=700)),
    ("generic", dict(kind="plain_sum", N=64, irregular=False)),
]


def _make(case):
    kind, kw = case
    return util.solar_problem(**kw) if kind == "solar" else util.generic_problem(**kw)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_engine_vs_oracle(hip, case):
    import torch
    from oracle import seq, cref
    from gadfly_amd.engine import DeviceBatch
    prob = _make(case)
    c, a, U, V = util.oracle_matrices(prob, seq)
    t, y = prob["t"], prob["y"]
    N, W = U.shape

    eng = DeviceBatch([prob["kernel"].get_device_coefficients()], t, diag=prob["diag_user"])
    ld = eng.ld
    # K0: generator rows
    assert _relmax(eng.U[0, :, :W].cpu().numpy(), U) < 1e-12
    assert _relmax(eng.V[0, :, :W].cpu().numpy(), V) < 1e-12
    assert _relmax(eng.a[0].cpu().numpy(), a) < 1e-14
    if ld > W:
        assert float(eng.U[0, :, W:].abs().max()) == 0.0
    P = np.ones((N, W))
    P[1:] = np.exp(c[None, :] * (t[:-1] - t[1:])[:, None])
    assert _relmax(eng.P[0, :, :W].cpu().numpy(), P) < 1e-13

    # K1 (+ fused forward solve) and the reductions
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    yd = torch.as_tensor(y).cuda()
    ll = eng.log_likelihood(yd, keep_W=True)
    assert int(eng.info[0]) == 0
    assert _relmax(eng.d[0].cpu().numpy(), d_ref) < 1e-9
    assert _relmax(eng.Wm[0, :, :W].cpu().numpy(), W_ref) < 1e-8
    z_ref = cref.solve_lower(t, c, U, W_ref, y)
    assert _relmax(eng.z[0].cpu().numpy(), z_ref) < 1e-9
    ll_ref = -0.5 * (np.sum(np.log(d_ref)) + N * np.log(2 * np.pi)) - 0.5 * np.sum(z_ref ** 2 / d_ref)
    assert abs(float(ll[0]) - ll_ref) <= RTOL_LL * abs(ll_ref)

    # K2/K3: sweeps, one and several right-hand sides
    rng = np.random.default_rng(0)
    for R in (1, 3, 70):
        Y = rng.normal(size=(N, R))
        Yd = torch.as_tensor(Y).cuda().reshape(1, N, R)
        Zl = eng.solve_lower(Yd)[0].cpu().numpy()
        assert _relmax(Zl, cref.solve_lower(t, c, U, W_ref, Y)) < TOL_VEC
        Zu = eng.solve_upper(Yd)[0].cpu().numpy()
        assert _relmax(Zu, cref.solve_upper(t, c, U, W_ref, Y)) < TOL_VEC
        Ai = eng.apply_inverse(Yd)[0].cpu().numpy()
        ref = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
        assert _relmax(Ai, ref) < TOL_VEC
        # K4: dot_tril
        Dt = eng.dot_tril(Yd)[0].cpu().numpy()
        ref = cref.matmul_lower(t, c, U, W_ref, Y * np.sqrt(d_ref)[:, None])
        assert _relmax(Dt, ref) < TOL_VEC

    # K5: conditional mean at new times
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y) / d_ref)
    ts = np.sort(np.concatenate([rng.uniform(t[0] - 0.5, t[-1] + 0.5, 57), t[5:8]]))
    co = prob["kernel"].get_device_coefficients()[:6]
    _, _, Us, Vs = seq.celerite_matrices(co, ts, 0.0)
    mu_ref = cref.general_matmul(ts, t, c, Us, Vs, U, V, alpha)
    tsd, Usd, Vsd = eng.matrices_at(ts)
    mu = eng.predict_at(torch.as_tensor(alpha).cuda().reshape(1, N), tsd, Usd, Vsd)[0].cpu().numpy()
    assert _relmax(mu, mu_ref) < TOL_VEC


def test_not_positive_definite(hip):
    """celerite2 raises LinAlgError at the first non-positive pivot; quiet=True -> -inf."""
    import gadfly_amd
    prob = util.generic_problem("mixed", 200)
    gp = gadfly_amd.GaussianProcess(prob["kernel"])
    bad = -2.0 * prob["kernel"].get_value(np.zeros(1))[0] * np.ones(len(prob["t"]))
    with pytest.raises(gadfly_amd.LinAlgError):
        gp.compute(prob["t"], diag=bad)
    gp.compute(prob["t"], diag=bad, quiet=True)
    assert gp.log_likelihood(prob["y"]) == -np.inf
    from oracle import seq
    c, a, U, V = util.oracle_matrices(dict(prob, diag_user=bad), seq)
    _, _, info = seq.factor(prob["t"], c, a, U, V)
    assert gp._failed_row == info == 1


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[9]], ids=["solar6", "solar30gaps", "mixed"])
def test_gaussian_process_api(hip, case):
    """The drop-in class against the dense O(N^3) oracle (independent of the recurrences)."""
    import gadfly_amd
    from oracle import dense
    prob = _make(case)
    if len(prob["t"]) > 1500:
        for k in ("t", "diag_user", "y"):
            prob[k] = prob[k][:1500]
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    N = len(t)
    co = k.get_device_coefficients()
    diag = prob["diag_user"] + co[6]
    mean = 3.0
    gp = gadfly_amd.GaussianProcess(k, t=t, mean=mean, diag=prob["diag_user"])
    ll_ref = dense.log_likelihood(co[:6], t, diag, y - mean)
    assert abs(gp.log_likelihood(y) - ll_ref) <= RTOL_LL * abs(ll_ref)
    ai = gp.apply_inverse(y)
    assert _relmax(ai, dense.apply_inverse(co[:6], t, diag, y)) < TOL_VEC
    rng = np.random.default_rng(5)
    n = rng.normal(size=(N, 4))
    assert _relmax(gp.dot_tril(n), dense.dot_tril(co[:6], t, diag, n)) < TOL_VEC
    # predict at the observed times (t=None): y - diag_user * alpha
    mu = gp.predict(y)
    alpha = dense.apply_inverse(co[:6], t, diag, y - mean)
    assert _relmax(mu, y - prob["diag_user"] * alpha) < TOL_VEC
    # predict at new times with variance (celerite2 semantics; K(t, t*) built on the device in
    # blocks of 64 query times: 100 queries = two blocks, some of them closer than the exposure
    # time to an observed point)
    ts = np.sort(np.concatenate([rng.uniform(t[0], t[-1], 97), t[10:13] + 0.3 * getattr(k, "delta", 0.0)]))
    mu_s, var_s = gp.predict(y, t=ts, return_var=True)
    Ks = k.get_value(ts[:, None] - t[None, :])
    K = dense.dense_K(co[:6], t, diag)
    mu_ref = Ks @ np.linalg.solve(K, y - mean) + mean
    var_ref = k.get_value(np.zeros(1))[0] - np.sum(Ks.T * np.linalg.solve(K, Ks.T), axis=0)
    # the solver sees transformed coefficients at every lag; Ks uses the exact kernel,
    # identical whenever |t* - t| >= delta, which generic/solar spacing guarantees here
    # except for a handful of close pairs -> compare the device mean with the device-consistent one
    co_k = dense.kernel_value(co[:6], ts[:, None] - t[None, :])
    assert _relmax(mu_s, co_k @ np.linalg.solve(K, y - mean) + mean) < TOL_VEC
    assert _relmax(var_s, var_ref) < 1e-5
    cov = gp.condition(y, ts).covariance
    cov_ref = k.get_value(ts[:, None] - ts[None, :]) - Ks @ np.linalg.solve(K, Ks.T)
    assert _relmax(cov, cov_ref) < 1e-5
    assert _relmax(np.diag(cov), var_s) < 1e-9
    # sample(): gadfly's quirk -- across-realisation mean removed (gp.py:392)
    np.random.seed(42)
    s = gp.sample(size=3)
    np.random.seed(42)
    nn = np.random.randn(N, 3)
    ref = dense.dot_tril(co[:6], t, diag, nn).T + mean
    ref -= ref.mean(axis=0)
    assert s.shape == (3, N)
    assert _relmax(s, ref) < TOL_VEC
    np.random.seed(7)
    s1 = gp.sample()
    np.random.seed(7)
    r1 = dense.dot_tril(co[:6], t, diag, np.random.randn(N)) + mean
    assert _relmax(s1, r1 - r1.mean()) < TOL_VEC


def test_batched_walkers_and_light_curves(hip):
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters, scale_hyperparameters
    from oracle import cref
    N, J = 1800, 6
    base = solar_like_hyperparameters(J)
    prob = util.solar_problem(J, N)
    t, y = prob["t"], prob["y"]
    # walkers: shared t, y
    kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 1000 + i), texp=60.0)
               for i in range(5)]
    ll = gadfly_amd.log_likelihood_batch(kernels, t, y, yerr=30.0)
    for i, k in enumerate(kernels):
        co = k.get_device_coefficients()
        ref, info = cref.loglike(co[:6], t, np.full(N, 900.0) + co[6], y)
        assert info == 0 and abs(ll[i] - ref) <= RTOL_LL * abs(ref)
    # light curves: own t, y
    rng = np.random.default_rng(2)
    kernels = [gadfly_amd.StellarOscillatorKernel(scale_hyperparameters(base, f), texp=58.85)
               for f in (0.3, 0.6, 1.0)]
    ts = np.stack([np.cumsum(rng.uniform(50e-6, 70e-6, N)) for _ in kernels])
    ys = rng.normal(size=(3, N)) * 50
    ll = gadfly_amd.log_likelihood_batch(kernels, ts, ys, yerr=30.0)
    for i, k in enumerate(kernels):
        co = k.get_device_coefficients()
        ref, info = cref.loglike(co[:6], ts[i], np.full(N, 900.0) + co[6], ys[i])
        assert info == 0 and abs(ll[i] - ref) <= RTOL_LL * abs(ref)


def test_edge_sizes(hip):
    """N = 1, 2, 3 and a single-column kernel."""
    import gadfly_amd
    from oracle import dense
    from gadfly_amd.terms import SHOTerm, TermSum
    k = TermSum(SHOTerm(S0=1.0, w0=3.0, Q=5.0))
    co = k.get_device_coefficients()
    for N in (1, 2, 3, 65):
        t = np.arange(N) * 0.3
        y = np.linspace(-1, 1, N) + 0.1
        gp = gadfly_amd.GaussianProcess(k, t=t, yerr=0.2)
        ref = dense.log_likelihood(co[:6], t, np.full(N, 0.04), y)
        assert abs(gp.log_likelihood(y) - ref) <= RTOL_LL * abs(ref)
        assert _relmax(gp.apply_inverse(y), dense.apply_inverse(co[:6], t, np.full(N, 0.04), y)) < TOL_VEC


STREAM_CASES = [
    ("solar", dict(J=6, N=3000), 256),
    ("solar", dict(J=20, N=2500, jitter_t=True), 512),
    ("solar", dict(J=30, N=3000, gaps=True), 1000),      # gap rule resets + ragged last tile
    ("solar", dict(J=30, N=2048, yerr=0.0), 8192),       # single tile
    ("solar", dict(J=32, N=1000), 64),                   # W = 64: no pad lanes
    ("solar", dict(J=40, N=1500), 256),                  # W = 80: multi-wave kernels (scaled-wide / v1)
    ("generic", dict(kind="overdamped", N=700), 128),    # real terms
    ("generic", dict(kind="mixed", N=900), 64),
    ("generic", dict(kind="sho_q100", N=512), 100),      # tile_rows not a multiple of 8
    ("solar", dict(J=70, N=700), 128),                   # W = 140: three column tiles, eight waves
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: f"{c[0]}-{c[1]}-T{c[2]}")
@pytest.mark.parametrize("mode", ["fused", "scaled", "v1"])
def test_streaming_loglike(hip, case, mode):
    """Tile-streamed evaluation (all three kernel families) against the oracle, including the
    state hand-off between tiles and the block-scaled coordinates' reset rows."""
    force_v1, allow_fused = mode == "v1", mode == "fused"
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    kind, kw, tile = case
    prob = _make((kind, kw))
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    eng = StreamingBatch([co], t, y, diag=prob["diag_user"], tile_rows=tile, force_v1=force_v1,
                         allow_fused=allow_fused)
    if allow_fused:
        assert eng._fused_ok() == (eng.W <= 63)
    if eng.W > 64:
        assert eng.scaled_wide == (not force_v1) and not eng.scaled
    ll = float(eng.log_likelihood()[0])
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    assert abs(ll - ref) <= RTOL_LL * abs(ref), (ll, ref)
    # a second evaluation re-uses the buffers (state must be re-zeroed)
    assert float(eng.log_likelihood()[0]) == ll


def test_streaming_fast_term_resets_every_row(hip):
    """A shot-noise-like term (c dt >> 4) forces a reset on every row of the scaled path."""
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    prob = util.solar_problem(6, 1500)
    k = prob["kernel"] + gadfly_amd.ShotNoiseKernel(S0=1e-3, w0=2.0e5, Q=0.5)
    co = k.get_device_coefficients()
    assert np.max(co[4]) * 60e-6 > 4.0
    t, y = prob["t"], prob["y"]
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    for force_v1, allow_fused in ((False, True), (False, False), (True, False)):
        eng = StreamingBatch([co], t, y, diag=prob["diag_user"], tile_rows=512,
                             force_v1=force_v1, allow_fused=allow_fused)
        ll = float(eng.log_likelihood()[0])
        assert abs(ll - ref) <= RTOL_LL * abs(ref), (force_v1, allow_fused, ll, ref)


def test_streaming_not_positive_definite(hip):
    from gadfly_amd.engine import StreamingBatch
    prob = util.generic_problem("mixed", 300)
    co = prob["kernel"].get_device_coefficients()
    bad = prob["diag_user"].copy()
    bad[137:] = -5.0 * prob["kernel"].get_value(np.zeros(1))[0]
    for force_v1, allow_fused in ((False, True), (False, False), (True, False)):
        eng = StreamingBatch([co, co], np.stack([prob["t"]] * 2), np.stack([prob["y"]] * 2),
                             diag=np.stack([prob["diag_user"], bad]), tile_rows=64,
                             force_v1=force_v1, allow_fused=allow_fused)
        ll = eng.log_likelihood().cpu().numpy()
        assert np.isfinite(ll[0]) and ll[1] == -np.inf
        assert int(eng.info[0]) == 0 and int(eng.info[1]) == 138


def test_streaming_large_phases_fall_back(hip):
    """Phases d*t beyond the fused kernel's sincos range (1e12 rad -- a JD-based axis reaches 5e9 and stays on
    the fused kernels, test_gpu_jd_axis.py): the engine must fall back to the materialised path (OCML sincos)
    and still match."""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    prob = util.solar_problem(20, 1200)
    t = prob["t"] + 1.0e8                      # phases of 2e12 rad
    co = prob["kernel"].get_device_coefficients()
    eng = StreamingBatch([co], t, prob["y"], diag=prob["diag_user"], tile_rows=256)
    assert not eng._fused_ok()
    ll = float(eng.log_likelihood()[0])
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], prob["y"])
    assert info == 0 and abs(ll - ref) <= RTOL_LL * abs(ref), (ll, ref)


def test_fused_kernels_take_large_phases(hip):
    """Time axes far from zero (phases d t up to ~1e9 rad, e.g. mission days or a 1e6-point series at one
    minute cadence): the in-kernel sincos reduces its argument with FMA over |x| < 1e12, so the fused
    sweeps -- streamed, time-parallel, and the wide kernel -- keep running and still match the oracle
    (both form theta = d t as ONE rounded product)."""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    for J, off in ((20, 4.0e4), (30, 9.0e4), (40, 4.0e4)):
        prob = util.solar_problem(J, 1500)
        t = prob["t"] + off
        co = prob["kernel"].get_device_coefficients()
        eng = StreamingBatch([co], t, prob["y"], diag=prob["diag_user"], tile_rows=512)
        eng.generator_period = 1
        assert 1.6e6 < eng._pack[6] * eng._tmax < 1e12
        assert eng._fused_ok() or eng._wide_ok()
        ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], prob["y"])
        ll = float(eng.log_likelihood()[0])
        assert info == 0 and abs(ll - ref) <= RTOL_LL * abs(ref), (J, ll, ref)
        ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=256)[0])
        assert abs(ll_tp - ref) <= RTOL_LL * abs(ref), (J, ll_tp, ref)


def test_streaming_irregular_cadence_fused(hip):
    """Irregular sampling: the fused kernel re-evaluates its cached exp(-c dt) on every row."""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    rng = np.random.default_rng(11)
    prob = util.solar_problem(12, 1500)
    t = np.cumsum(rng.uniform(20e-6, 200e-6, len(prob["t"])))
    co = prob["kernel"].get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], prob["y"])
    for allow_fused in (True, False):
        eng = StreamingBatch([co], t, prob["y"], diag=prob["diag_user"], tile_rows=320,
                             allow_fused=allow_fused)
        ll = float(eng.log_likelihood()[0])
        assert info == 0 and abs(ll - ref) <= RTOL_LL * abs(ref), (allow_fused, ll, ref)


@pytest.mark.parametrize("J,N,tile", [(1, 700, 128), (6, 1500, 256), (30, 2500, 512), (31, 2000, 320)],
                         ids=["W2", "W12", "W60", "W62"])
def test_fused_kernels_by_name(hip, J, N, tile):
    """Both fused kernels by name -- the 2 x 32 lane tiling (k_factor7, what `auto` picks for these
    complex-only kernels; W = 62 fills its 31 term blocks) and one column per lane (k_factor3) --
    against the oracle, streamed over several tiles and in chunk mode (row stores for the stored
    factor)."""
    import torch
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    prob = util.solar_problem(J, N, gaps=(J == 30))
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    assert len(co[0]) == 0 and len(co[2]) == J
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    c, a, U, V = util.oracle_matrices(prob, __import__("oracle.seq", fromlist=["seq"]))
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    Y = np.random.default_rng(J).normal(size=(len(t), 2))
    ref_ai = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    got = {}
    engines = {}
    for mode in (hip.GF_SWEEP_TILED, hip.GF_SWEEP_COLUMN):
        # the variant is a call argument of the C-ABI: both engines stay alive side by side
        eng = engines[mode] = StreamingBatch([co], t, y, diag=prob["diag_user"], tile_rows=tile)
        eng.sweep_variant = mode
        assert eng._fused_ok()
    for mode, eng in engines.items():
        ll = float(eng.log_likelihood()[0])
        ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=128)[0])
        fac = eng.stored_factor(chunk_len=128)
        assert abs(ll - ref) <= RTOL_LL * abs(ref), (mode, ll, ref)
        assert abs(ll_tp - ref) <= RTOL_LL * abs(ref), (mode, ll_tp, ref)
        ai = fac.apply_inverse(torch.as_tensor(Y).cuda().reshape(1, len(t), 2))[0].cpu().numpy()
        assert _relmax(ai, ref_ai) < TOL_VEC, mode
        got[mode] = ll
    assert abs(got[hip.GF_SWEEP_TILED] - got[hip.GF_SWEEP_COLUMN]) <= 1e-11 * abs(ref)


def test_sweep_options_are_checked(hip):
    """gen_period / variant are validated call arguments (no process-wide switches): a tiled sweep
    asked for a kernel with real terms, or a period that is not a power of two, is refused."""
    from gadfly_amd.engine import StreamingBatch
    prob = util.generic_problem("mixed", 300)
    co = prob["kernel"].get_device_coefficients()
    eng = StreamingBatch([co], prob["t"], prob["y"], diag=prob["diag_user"], tile_rows=64)
    eng.sweep_variant = hip.GF_SWEEP_TILED
    with pytest.raises(hip.GadflyHipError, match="GF_SWEEP_TILED"):
        eng.log_likelihood()
    eng.sweep_variant = hip.GF_SWEEP_AUTO
    eng.generator_period = 3
    with pytest.raises(hip.GadflyHipError, match="gen_period"):
        eng.log_likelihood()
    eng.generator_period = 1
    assert np.isfinite(float(eng.log_likelihood()[0]))


TP_CASES = [
    ("solar", dict(J=6, N=3000), 256),
    ("solar", dict(J=30, N=5000), 512),
    ("solar", dict(J=30, N=3000, yerr=0.0), 320),
    ("solar", dict(J=20, N=2500, jitter_t=True), 128),
    ("solar", dict(J=30, N=3000, gaps=True), 1000),      # 1000 -> 1024, ragged last chunk
    ("generic", dict(kind="mixed", N=900), 64),
    ("generic", dict(kind="overdamped", N=700), 192),
]


@pytest.mark.parametrize("case", TP_CASES, ids=lambda c: f"{c[0]}-{c[1]}-L{c[2]}")
def test_time_parallel_loglike(hip, case):
    """Exact chunk-parallel evaluation (nominal pass, Phi/G, LFT combine, final pass) must
    reproduce the sequential result: log-likelihood, every pivot d_n and every z_n."""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref, seq
    kind, kw, L = case
    prob = _make((kind, kw))
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    eng = StreamingBatch([co], t, y, diag=prob["diag_user"])
    ll_seq = float(eng.log_likelihood()[0])
    ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=L)[0])
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    assert abs(ll_tp - ref) <= RTOL_LL * abs(ref), (ll_tp, ref)
    assert abs(ll_tp - ll_seq) <= 1e-10 * abs(ref)
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    z_ref = cref.solve_lower(t, c, U, W_ref, y)
    N = len(t)
    assert _relmax(eng._tp["d"][:N].cpu().numpy(), d_ref) < 1e-9
    assert _relmax(eng._tp["z"][:N].cpu().numpy(), z_ref) < 1e-8


def test_time_parallel_batch_of_two(hip):
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters
    import gadfly_amd
    from oracle import cref
    base = solar_like_hyperparameters(12)
    ks = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 5 + i), texp=60.0)
          for i in range(2)]
    prob = util.solar_problem(12, 2600)
    eng = StreamingBatch([k.get_device_coefficients() for k in ks], prob["t"], prob["y"],
                         diag=prob["diag_user"])
    ll = eng.log_likelihood_time_parallel(chunk_len=384).cpu().numpy()
    for i, k in enumerate(ks):
        co = k.get_device_coefficients()
        ref, _ = cref.loglike(co[:6], prob["t"], prob["diag_user"] + co[6], prob["y"])
        assert abs(ll[i] - ref) <= RTOL_LL * abs(ref)


@pytest.mark.parametrize("J,variant,two_sweep", [(20, "tiled", False), (20, "tiled", True), (6, "column", True),
                                                 (40, "wide", False), (40, "wide", True)])
def test_nominal_passes_need_no_cleared_state(hip, J, variant, two_sweep):
    """GF_SWEEP_ZERO_START: the nominal pass writes its state slots without reading them, so the engine no
    longer clears them -- slots (and their padding rows) poisoned with NaN between two evaluations must not
    change a bit of the result, on both fused kernels, the wide kernel, with and without the final pass."""
    import torch
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    prob = util.solar_problem(J, 3000)
    co = prob["kernel"].get_device_coefficients()
    eng = StreamingBatch([co, co], prob["t"], prob["y"], diag=prob["diag_user"])
    if variant != "wide":
        eng.sweep_variant = hip.GF_SWEEP_TILED if variant == "tiled" else hip.GF_SWEEP_COLUMN
    eng.two_sweep = two_sweep
    first = eng.log_likelihood_time_parallel(chunk_len=256).clone()
    ws = eng._wide_ws_cache if variant == "wide" else eng._tp
    for name in ("S", "F"):
        if name in ws:
            ws[name].fill_(float("nan"))
    again = eng.log_likelihood_time_parallel(chunk_len=256)
    assert torch.equal(first, again)
    ref, _ = cref.loglike(co[:6], prob["t"], prob["diag_user"] + co[6], prob["y"])
    assert abs(float(again[1]) - ref) <= RTOL_LL * abs(ref)


TREE_CASES = [
    ("solar", dict(J=30, N=5000), 512, 1),       # 10 chunks -> P = 16 (identity padding)
    ("solar", dict(J=30, N=4096), 512, 1),       # 8 chunks  -> P = 8 (no padding)
    ("solar", dict(J=6, N=3000), 64, 1),         # 47 chunks -> P = 64
    ("solar", dict(J=20, N=2500, jitter_t=True), 128, 1),
    ("solar", dict(J=30, N=3000, gaps=True), 192, 1),
    ("generic", dict(kind="mixed", N=900), 64, 1),
    ("solar", dict(J=12, N=2600), 384, 2),       # batch of two
]


@pytest.mark.parametrize("case", TREE_CASES, ids=lambda c: f"{c[0]}-{c[1]}-L{c[2]}-B{c[3]}")
def test_time_parallel_tree_combine(hip, case):
    """The log-depth (Blelloch) combine gives the chunk start states of the sequential combine:
    same log-likelihood, d_n and z_n as the sequential evaluation and as the oracle."""
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref
    kind, kw, L, B = case
    prob = _make((kind, kw))
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    co = k.get_device_coefficients()
    eng = StreamingBatch([co] * B, t, y, diag=prob["diag_user"])
    N = len(t)
    eng.tree_min_chunks = 1 << 30                   # sequential combine
    ll_lin = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    S_lin = eng._tp["S"].clone(); F_lin = eng._tp["F"].clone()
    d_lin = eng._tp["d"][:B * N].clone(); z_lin = eng._tp["z"][:B * N].clone()
    eng.tree_min_chunks = 2                         # tree combine
    ll_tree = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
    assert "tree" in eng._tp
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert info == 0
    for b in range(B):
        assert abs(ll_tree[b] - ref) <= RTOL_LL * abs(ref), (ll_tree[b], ref)
        assert abs(ll_tree[b] - ll_lin[b]) <= 1e-10 * abs(ref)
    assert _relmax(eng._tp["d"][:B * N].cpu().numpy(), d_lin.cpu().numpy()) < 1e-9
    zs = float(z_lin.abs().max())
    assert float((eng._tp["z"][:B * N] - z_lin).abs().max()) < 1e-8 * zs


@pytest.mark.parametrize("keep", [True, False], ids=["factor-kept-by-compute", "factor-built-on-demand"])
def test_gaussian_process_fast_path_long_series(hip, keep):
    """compute + log_likelihood of the drop-in class on a series long enough to take the
    time-parallel engine, against the oracle.  compute() keeps the factor of a long series
    (log_likelihood is then one forward sweep on it); beyond STORE_MAX_ROWS it does not, log_likelihood
    is the fused evaluation and the first predict() builds the stored factor."""
    import gadfly_amd
    from oracle import cref
    prob = util.solar_problem(12, 40000)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    gp = gadfly_amd.GaussianProcess(k, mean=1.5)
    if not keep:
        gp.STORE_MAX_ROWS = 1000
    gp.compute(t, yerr=30.0)
    assert gp._fast is not None and gp._fast._tp_used and (gp._factor is not None) == keep
    co = k.get_device_coefficients()
    ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y - 1.5)
    ll = gp.log_likelihood(y)
    assert info == 0 and abs(ll - ref) <= RTOL_LL * abs(ref)
    c, a, U, V = util.oracle_matrices(prob, __import__("oracle.seq", fromlist=["seq"]))
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    assert abs(gp._log_det - np.sum(np.log(d_ref))) <= 1e-10 * abs(gp._log_det)
    mu = gp.predict(y)
    assert gp._factor is not None
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y - 1.5) / d_ref)
    assert _relmax(mu, y - prob["diag_user"] * alpha) < TOL_VEC
    # the stored factor must survive later evaluations that reuse the time-parallel buffers
    # (log_likelihood of another y refills the shared transition / pivot work arrays)
    y2 = y[::-1].copy()
    ref2, _ = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y2 - 1.5)
    assert abs(gp.log_likelihood(y2) - ref2) <= RTOL_LL * abs(ref2)
    assert _relmax(gp.apply_inverse(y - 1.5), alpha) < TOL_VEC


def test_full_size_round_trip(hip):
    """BASELINE.json's full size (N = 1e6, J = 30) through size-independent identities:
    y = L n  =>  y^T K^-1 y = n^T n  (dot_tril, then apply_inverse), and the log-likelihood of that
    y equals -(n^T n + log det K + N log 2 pi) / 2.  Exercises the time-parallel factor (tree
    combine), the stored scaled factor and its chunk-parallel sweeps at the benchmark's shape."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    N, J = 1_000_000, 30
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
    t = uniform_times(N, 60.0)
    gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
    rng = np.random.Generator(np.random.PCG64(2024))
    n = rng.normal(size=N)
    y = gp.dot_tril(n)
    alpha = gp.apply_inverse(y)
    nn = float(n @ n)
    assert abs(float(y @ alpha) - nn) <= 1e-8 * nn
    ll = gp.log_likelihood(y)
    ref = -0.5 * (nn + gp._log_det + N * np.log(2.0 * np.pi))
    assert abs(ll - ref) <= RTOL_LL * abs(ref), (ll, ref)
    # linearity of the solve at full size
    a2 = gp.apply_inverse(2.5 * y)
    assert _relmax(a2, 2.5 * alpha) < 1e-10


@pytest.mark.parametrize("case", [TP_CASES[0], TP_CASES[1], TP_CASES[4], TP_CASES[5],
                                  ("solar", dict(J=12, N=2000), 256), ("solar", dict(J=20, N=2500, jitter_t=True), 128)],
                         ids=["solar6", "solar30", "solar30gaps", "mixed", "solar12", "solar20jitter"])
def test_scaled_factor_sweeps(hip, case):
    """Chunk-parallel triangular sweeps on the stored scaled factor against the oracle:
    solve_lower / solve_upper / apply_inverse / dot_tril, 1 and many right-hand sides."""
    import torch
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref, seq
    kind, kw, L = case
    prob = _make((kind, kw))
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    N = len(t)
    co = k.get_device_coefficients()
    eng = StreamingBatch([co], t, y, diag=prob["diag_user"])
    fac = eng.stored_factor(chunk_len=L)
    assert fac.nch > 1
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0 and _relmax(fac.d[0].cpu().numpy(), d_ref) < 1e-9
    assert float(fac.Ut[:, fac.W:].abs().max()) == 0.0 and float(fac.Wt[:, fac.W:].abs().max()) == 0.0
    rng = np.random.default_rng(3)
    for R in (1, 5, 16, 33, 70):        # (dot_tril: R >= 16 runs on the matrix pipe, k_mmR_mfma)
        Y = rng.normal(size=(N, R))
        Yd = torch.as_tensor(Y).cuda().reshape(1, N, R)
        assert _relmax(fac.solve_lower(Yd)[0].cpu().numpy(), cref.solve_lower(t, c, U, W_ref, Y)) < TOL_VEC
        assert _relmax(fac.solve_upper(Yd)[0].cpu().numpy(), cref.solve_upper(t, c, U, W_ref, Y)) < TOL_VEC
        ref = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
        assert _relmax(fac.apply_inverse(Yd)[0].cpu().numpy(), ref) < TOL_VEC
        ref = cref.matmul_lower(t, c, U, W_ref, Y * np.sqrt(d_ref)[:, None])
        got = fac.dot_tril(Yd)
        assert _relmax(got[0].cpu().numpy(), ref) < TOL_VEC
        # dot_tril's local pass starts from zero by itself: its state slots are not cleared any more, so a
        # freed block full of NaN (what the allocator hands back for them next) must not change a bit
        nchm = fac._mm_chunking(R)[1]
        poison = torch.full((nchm, 64 * R), float("nan"), dtype=torch.float64, device="cuda")
        del poison
        assert torch.equal(fac.dot_tril(Yd), got)
    # single-chunk factor (nch = 1) takes the same code path without a combine
    fac1 = eng.stored_factor(chunk_len=4 * N)
    assert fac1.nch == 1
    Y = rng.normal(size=(N, 3))
    Yd = torch.as_tensor(Y).cuda().reshape(1, N, 3)
    ref = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    assert _relmax(fac1.apply_inverse(Yd)[0].cpu().numpy(), ref) < TOL_VEC


@pytest.mark.parametrize("case,L", [(("solar", dict(J=30, N=9000)), 128), (("generic", dict(kind="mixed", N=2900)), 64),
                                    (("solar", dict(J=12, N=6000, gaps=True)), 64)],
                         ids=["solar30-71chunks", "mixed-46chunks", "solar12gaps-94chunks"])
def test_scaled_factor_two_level_combine(hip, case, L):
    """Long chains of chunks: the solves' combine runs in two levels on the composed segment
    transitions (gf_chunk_segment_transitions / gf_chunk_linear_combine_seg).  Same answers as the
    oracle and as the plain sequential combine, for ragged segment counts, 1 and several RHS."""
    import torch
    from gadfly_amd.engine import StreamingBatch
    from oracle import cref, seq
    prob = _make(case)
    t = prob["t"]
    N = len(t)
    eng = StreamingBatch([prob["kernel"].get_device_coefficients()], t, prob["y"], diag=prob["diag_user"])
    fac = eng.stored_factor(chunk_len=L)
    assert fac.nch >= fac.SEG_MIN_CHUNKS
    c, a, U, V = util.oracle_matrices(prob, seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    rng = np.random.default_rng(5)
    for R in (1, 4):
        Y = rng.normal(size=(N, R))
        Yd = torch.as_tensor(Y).cuda().reshape(1, N, R)
        lo = fac.solve_lower(Yd)[0].cpu().numpy()
        up = fac.solve_upper(Yd)[0].cpu().numpy()
        assert fac._Psi is not None and fac._Psi[1].shape[0] == -(-fac.nch // fac._Psi[0])
        assert _relmax(lo, cref.solve_lower(t, c, U, W_ref, Y)) < TOL_VEC
        assert _relmax(up, cref.solve_upper(t, c, U, W_ref, Y)) < TOL_VEC
        ref = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
        assert _relmax(fac.apply_inverse(Yd)[0].cpu().numpy(), ref) < TOL_VEC
        # the plain combine on the same factor
        fac.SEG_MIN_CHUNKS = 10 ** 9
        assert _relmax(fac.solve_lower(Yd)[0].cpu().numpy(), lo) < 1e-11
        assert _relmax(fac.solve_upper(Yd)[0].cpu().numpy(), up) < 1e-11
        del fac.SEG_MIN_CHUNKS
    # other segment lengths, including one that leaves a single chunk in the last segment
    for seg_len in (2, 5, fac.nch - 1):
        Psi = torch.empty((-(-fac.nch // seg_len), 4096), dtype=torch.float64, device="cuda")
        PsiT = torch.empty_like(Psi)
        rc = hip.load().gf_chunk_segment_transitions(1, fac.nch, seg_len, hip.ptr(fac.Phi), hip.ptr(Psi),
                                                     hip.ptr(fac._PhiT), hip.ptr(PsiT), None)
        hip.check(rc, "gf_chunk_segment_transitions")
        fac._Psi, fac._PsiT = (seg_len, Psi), PsiT
        assert _relmax(fac.solve_lower(Yd)[0].cpu().numpy(), lo) < 1e-11
        assert _relmax(fac.solve_upper(Yd)[0].cpu().numpy(), up) < 1e-11
        # without the transposed copies the backward combine reads the transposes out of Phi / Psi
        keep = fac._PhiT, fac._PsiT
        fac._PhiT = fac._PsiT = None
        assert _relmax(fac.solve_upper(Yd)[0].cpu().numpy(), up) < 1e-11
        fac._PhiT, fac._PsiT = keep


def test_recompute_refactorises_everything(hip):
    """celerite2's recompute() after a kernel change refactorises: apply_inverse / predict / dot_tril
    must use the NEW kernel's factor (a stored factor of the old kernel survived)."""
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters
    from oracle import cref, seq
    prob = util.solar_problem(12, 20000)
    t, y = prob["t"], prob["y"]
    gp = gadfly_amd.GaussianProcess(prob["kernel"], t=t, yerr=30.0)
    old = gp.apply_inverse(y)                       # builds the stored factor of the first kernel
    assert gp._factor is not None
    k2 = gadfly_amd.StellarOscillatorKernel(
        jitter_hyperparameters(solar_like_hyperparameters(12), 77, frac=0.3), texp=60.0)
    gp.kernel = k2
    gp.recompute()
    c, a, U, V = util.oracle_matrices(dict(prob, kernel=k2), seq)
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    alpha = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, y) / d_ref)
    new = gp.apply_inverse(y)
    assert _relmax(new, alpha) < TOL_VEC
    assert _relmax(old, alpha) > 1e-3               # the kernels really differ
    assert _relmax(gp.predict(y), y - prob["diag_user"] * alpha) < TOL_VEC
    n = np.random.default_rng(1).normal(size=len(t))
    assert _relmax(gp.dot_tril(n), cref.matmul_lower(t, c, U, W_ref, n * np.sqrt(d_ref))) < TOL_VEC
    co = k2.get_device_coefficients()
    ref, _ = cref.loglike(co[:6], t, prob["diag_user"] + co[6], y)
    assert abs(gp.log_likelihood(y) - ref) <= RTOL_LL * abs(ref)


def test_time_parallel_reports_first_failing_pivot(hip):
    """A matrix that stops being positive definite part-way: the time-parallel evaluation must name
    the FIRST non-positive pivot, as celerite2 and the sequential sweep do, although
    every chunk after it also fails (and fails first in wall-clock order)."""
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from oracle import seq
    prob = util.solar_problem(6, 12000)
    k, t, y = prob["kernel"], prob["t"], prob["y"]
    bad = prob["diag_user"].copy()
    bad[5000:] = -2.0 * k.get_value(np.zeros(1))[0]
    co = k.get_device_coefficients()
    c, a, U, V = util.oracle_matrices(dict(prob, diag_user=bad), seq)
    _, _, info_ref = seq.factor(t, c, a, U, V)
    assert info_ref == 5001
    eng = StreamingBatch([co], t, y, diag=bad, tile_rows=1024)
    assert eng.log_likelihood().cpu().numpy()[0] == -np.inf and int(eng.info[0]) == info_ref
    for L in (256, 1024, 4096):
        ll = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
        assert ll[0] == -np.inf and int(eng.info[0]) == info_ref, (L, int(eng.info[0]))
    gp = gadfly_amd.GaussianProcess(k)
    with pytest.raises(gadfly_amd.LinAlgError, match="pivot 5001 "):
        gp.compute(t, diag=bad)
