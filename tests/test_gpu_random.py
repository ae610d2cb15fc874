"""GPU: seeded random problems through every engine path against the C oracle -- random numbers
of over/under-damped SHO terms, cadence patterns (uniform, jittered, gaps, clusters, duplicates
of the cadence), noise levels, tile / chunk sizes.  Complements the fixed cases of
test_gpu_parity.py; the seeds are fixed, so a failure is reproducible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL_LL, TOL_VEC = 1e-8, 1e-6


def _relmax(x, ref):
    return float(np.max(np.abs(np.asarray(x) - ref)) / max(float(np.max(np.abs(ref))), 1e-300))


def _problem(seed):
    from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
    rng = np.random.Generator(np.random.PCG64(seed))
    J = int(rng.integers(1, 31))
    n_over = int(rng.integers(0, min(J, 3) + 1)) if rng.random() < 0.4 else 0
    terms = []
    for j in range(J):
        w0 = float(np.exp(rng.uniform(np.log(0.5), np.log(3000.0))))
        Q = float(rng.uniform(0.05, 0.45)) if j < n_over else float(np.exp(rng.uniform(np.log(0.5), np.log(300.0))))
        S0 = float(np.exp(rng.uniform(-2, 4)))
        terms.append(SHOTerm(S0=S0, w0=w0, Q=Q))
    dt = float(np.exp(rng.uniform(np.log(2e-5), np.log(2e-3))))          # cadence in 1e6 s
    N = int(rng.integers(40, 5000))
    kind = rng.choice(["uniform", "jitter", "gaps", "clusters"])
    t = np.arange(N) * dt
    if kind == "jitter":
        t = t + rng.uniform(-0.3, 0.3, N) * dt
    elif kind == "gaps":
        keep = np.ones(N, bool)
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, N - 1)); keep[a:a + int(rng.integers(1, max(2, N // 8)))] = False
        keep[0] = True
        t = t[keep]
    elif kind == "clusters":
        t = np.sort(rng.uniform(0, N * dt, N))
        t = np.unique(np.round(t / (dt * 1e-3)) * (dt * 1e-3))            # no exact duplicates
    t = np.sort(t)
    N = len(t)
    kernel = TermConvolution(TermSum(*terms), float(rng.uniform(0.1, 1.0)) * dt)
    yerr = 0.0 if rng.random() < 0.2 else float(np.exp(rng.uniform(-3, 2)))
    amp = float(np.sqrt(kernel.get_value(np.zeros(1))[0]))
    y = amp * rng.normal(size=N) + np.cumsum(rng.normal(size=N)) * 0.1 * amp
    return dict(kernel=kernel, t=t, y=y, diag_user=np.full(N, yerr ** 2), rng=rng, kind=kind, J=J)


@pytest.mark.parametrize("seed", range(100, 140))
def test_random_problem(hip, seed):
    import torch
    from gadfly_amd.engine import DeviceBatch, StreamingBatch
    from oracle import cref, seq
    prob = _problem(seed)
    k, t, y, du, rng = prob["kernel"], prob["t"], prob["y"], prob["diag_user"], prob["rng"]
    co = k.get_device_coefficients()
    N = len(t)
    ref, info = cref.loglike(co[:6], t, du + co[6], y)
    if info != 0:
        # not positive definite (clustered times closer than the exposure): the fused path and the
        # general path must both stop at the oracle's failing row
        eng = StreamingBatch([co], t, y, diag=du, tile_rows=256)
        ll = eng.log_likelihood()
        gen = DeviceBatch([co], t, diag=du)
        gen.log_likelihood(torch.as_tensor(y).cuda())
        assert int(eng.info[0]) == info and int(gen.info[0]) == info and float(ll[0]) == float("-inf")
        return
    c, a, U, V = seq.celerite_matrices(co[:6], t, du + co[6])
    d_ref, W_ref, _ = cref.factor(t, c, a, U, V)
    # problems too ill-conditioned for the 1e-8 bar in ANY double-precision implementation are
    # skipped: the plain-C recurrence itself must agree with the 80-bit one to 1e-9 (a 2000-seed
    # sweep had two cases with conditions of 4e7 and 1e8 where C was at 1.7e-9 and the GPU, with
    # exact generator rows, at 1.4e-8 / 9e-9: float64's edge, not an implementation defect)
    ld = np.longdouble
    cl, al, Ul, Vl = seq.celerite_matrices(co[:6], t, du + co[6], dtype=ld)
    dl, Wl, _ = seq.factor(t.astype(ld), cl, al, Ul, Vl)
    zl = seq.solve_lower(t.astype(ld), cl, Ul, Wl, y.astype(ld))
    ll80 = float(-0.5 * (np.sum(np.log(dl)) + N * np.log(2 * ld(np.pi))) - 0.5 * np.sum(zl * zl / dl))
    cond = float(al.max() / dl.min())
    if abs(ref - ll80) > 1e-9 * abs(ll80) or cond > 1e7:
        # (cond > 1e7: a 5000-seed sweep had a problem at 2.4e7 where the C recurrence happened to land
        # 4e-10 from the 80-bit result while EVERY GPU path, the unscaled celerite recurrence included,
        # sat at 1.5e-8 -- rounding of the inputs times the condition, not an implementation defect)
        pytest.skip(f"conditioning {cond:.1e}: beyond float64 at 1e-8")
    tile = int(rng.choice([64, 128, 320, 1024, 8192]))
    eng = StreamingBatch([co], t, y, diag=du, tile_rows=tile)
    # exact generator rows (the drop-in class' setting) for half of the cases; for the others the
    # period the product's own rule allows (StreamingBatch.calibrate_generator: rotation error
    # ~1.6e-15 * period * condition kept below 1e-9), capped at 16
    per = 1
    while per < 16 and 1.6e-15 * (2 * per) * cond <= 1e-9:
        per *= 2
    eng.generator_period = 1 if seed % 2 == 0 else per
    tag = (seed, prob["kind"], prob["J"], N, tile, eng._pack[5], eng._fused_ok(), eng.generator_period)
    ll = float(eng.log_likelihood()[0])
    assert int(eng.info[0]) == 0, tag
    assert abs(ll - ref) <= RTOL_LL * abs(ref), (tag, ll, ref)
    Y = rng.normal(size=(N, 3))
    Yd = torch.as_tensor(Y).cuda().reshape(1, N, 3)
    ai_ref = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    if eng._fused_ok():
        L = int(rng.choice([64, 128, 256, 512])) * max(1, eng._pack[5] // 64)
        eng.tree_min_chunks = int(rng.choice([2, 1 << 30]))              # tree or sequential combine
        ll_tp = float(eng.log_likelihood_time_parallel(chunk_len=L)[0])
        assert abs(ll_tp - ref) <= RTOL_LL * abs(ref), (tag, L, ll_tp, ref)
        # the same from two sweeps (nominal sums + per-chunk corrections): these matrices are
        # positive definite (info == 0 above), so the result is finite and the same number
        eng.two_sweep = True
        ll_2 = float(eng.log_likelihood_time_parallel(chunk_len=L)[0])
        eng.two_sweep = False
        assert abs(ll_2 - ref) <= RTOL_LL * abs(ref), (tag, L, ll_2, ref)
        fac = eng.stored_factor(chunk_len=L)
        assert _relmax(fac.apply_inverse(Yd)[0].cpu().numpy(), ai_ref) < TOL_VEC, (tag, L)
        dt_ref = cref.matmul_lower(t, c, U, W_ref, Y * np.sqrt(d_ref)[:, None])
        assert _relmax(fac.dot_tril(Yd)[0].cpu().numpy(), dt_ref) < TOL_VEC, (tag, L)
    gen = DeviceBatch([co], t, diag=du)
    ll_g = float(gen.log_likelihood(torch.as_tensor(y).cuda(), keep_W=True)[0])
    assert abs(ll_g - ref) <= RTOL_LL * abs(ref), (tag, ll_g, ref)
    assert _relmax(gen.apply_inverse(Yd)[0].cpu().numpy(), ai_ref) < TOL_VEC, tag


def test_generator_calibration(hip):
    """The condition estimate (max a / min d from the reduction kernel) and the period it selects:
    a well-conditioned solar-like problem gets the longest period, an ill-conditioned one (seed
    1127: condition 4e5) exact rows; the batched evaluator adapts after its first evaluation."""
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    from oracle import cref, seq
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(12), texp=60.0)
    t = uniform_times(20000, 60.0)
    y = np.random.default_rng(1).normal(size=len(t)) * 80.0
    ev = gadfly_amd.BatchedLogLikelihood([k, k], t, y, yerr=30.0)
    assert ev.engine.generator_period == 4                  # until something has been measured
    ev.two_sweep = False                                    # (an evaluation with a final pass: the true pivots)
    ev.evaluate()
    co = k.get_device_coefficients()
    c, a, U, V = seq.celerite_matrices(co[:6], t, np.full(len(t), 900.0) + co[6])
    d_ref, _, _ = cref.factor(t, c, a, U, V)
    cond_ref = float(a.max() / d_ref.min())
    assert abs(ev.engine.condition_estimate() - cond_ref) <= 1e-6 * cond_ref
    assert ev.engine.generator_period == 64 and cond_ref < 1e3
    # the two-sweep evaluation (the batched default) sees the nominal pass' pivots only -- never smaller than
    # the true ones -- and scales its estimate by a margin: not below the true condition, at most 1.5 x above
    ev.two_sweep = True
    ev.evaluate()
    assert ev.engine._two_sweep_used
    est = ev.engine.condition_estimate()
    assert cond_ref <= est <= 1.5 * cond_ref * (1 + 1e-12) and ev.engine.generator_period == 64

    prob = _problem(1127)
    co = prob["kernel"].get_device_coefficients()
    eng = StreamingBatch([co], prob["t"], prob["y"], diag=prob["diag_user"])
    eng.log_likelihood()
    cond, period = eng.calibrate_generator()
    assert cond > 1e5 and period == 1
    ref, _ = cref.loglike(co[:6], prob["t"], prob["diag_user"] + co[6], prob["y"])
    assert abs(float(eng.log_likelihood()[0]) - ref) <= 2e-9 * abs(ref)


def test_accuracy_guard_reruns_an_ill_conditioned_walker(hip):
    """The generator period is calibrated on well-conditioned walkers (period 64); then ONE walker of
    a later proposal is badly conditioned (amplitudes x 1e6, frequencies x 0.03: a process so smooth
    at this cadence that its condition estimate max(a) / min(d) is 3e5).  The device-side guard
    must flag exactly that evaluation and repeat it with exact rows, so that every entry still
    matches the oracle at 1e-8 -- accuracy does not depend on the sampler standing still."""
    import gadfly_amd
    from gadfly_amd.core import Hyperparameters
    from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters
    from oracle import cref
    N, J, B = 6000, 12, 6
    base = solar_like_hyperparameters(J)
    t = np.arange(N) * 60e-6
    rng = np.random.default_rng(9)
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 50 + i), texp=60.0)
               for i in range(B)]
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0, tile_rows=1024)
    ev.engine.force_streaming = True
    ev.evaluate()
    assert ev.engine.generator_period == 64 and ev.guard_reruns == 0
    # next proposal: walker 3 jumps to a region with huge amplitudes and slow oscillations
    hot = Hyperparameters([dict(hyperparameters=dict(S0=p["hyperparameters"]["S0"] * 1e6,
                                                     w0=p["hyperparameters"]["w0"] * 0.03,
                                                     Q=p["hyperparameters"]["Q"]),
                                metadata=dict(p["metadata"])) for p in base], name="hot")
    kernels2 = list(kernels)
    kernels2[3] = gadfly_amd.StellarOscillatorKernel(hot, texp=60.0)
    ev.auto_generator_period = False                  # the period stays at 64: only the guard protects
    out = ev.evaluate_device(ev.pack(kernels2))       # asynchronous path
    unguarded = out.clone()
    assert ev.resolve() == 1 and ev.guard_reruns == 1
    got = out.cpu().numpy()
    for i, k in enumerate(kernels2):
        co = k.get_device_coefficients()
        ref, info = cref.loglike(co[:6], t, np.full(N, 900.0) + co[6], y)
        assert info == 0 and abs(got[i] - ref) <= 1e-8 * abs(ref), (i, got[i], ref)
    # the other entries were not touched; the flagged one was replaced by the exact-row result
    same = np.arange(B) != 3
    assert np.array_equal(unguarded.cpu().numpy()[same], got[same])
    ev.engine.generator_period = 1
    exact = ev.evaluate_device(ev.pack(kernels2)).cpu().numpy()
    assert exact[3] == got[3]
