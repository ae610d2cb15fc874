#!/usr/bin/env python
"""
BASELINE.json's configurations beyond the headline, measured on ONE GPU (whole batch on this GPU):

  cfg2  N=1e6, J=30 single series through the drop-in class: compute, log_likelihood,
        predict(y) [conditional mean at the observed times], predict(y, t*=1000 new times)
  cfg3  256 Kepler-cadence light curves (N=65,000, J=20), own t, y, yerr and kernel each
  cfg4  512 MCMC walkers on one N=200,000, J=40 series (shared t, y)
  cfg5  dot_tril of 64 normal vectors (GP.sample's arithmetic) at N=500,000, J=30

Each function returns a dict of measured numbers (ms, units/s, algorithmic GB/s per SURVEY.md 8d and
its fraction of the 8 TB/s HBM peak) plus a ``_sample``: the inputs and the GPU's answer for ONE
entry, which bench.py checks against the oracle (this module never touches oracle/).

Usage: python tools/configs.py           (prints the table; bench.py embeds the same numbers)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0


def _clock(torch, fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    out = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def _guarded(ev):
    """One evaluation as a caller sees it: enqueue, then the accuracy guard (reruns of flagged entries
    with exact generator rows) before the values are read -- inside the timing."""
    out = ev.evaluate_device()
    ev.resolve()
    return out


def _path(eng):
    if eng._fused_ok():
        return "time-parallel fused" if getattr(eng, "_tp_used", False) else "fused"
    if getattr(eng, "_wide_ok", lambda: False)():
        if getattr(eng, "_last_wide_tp", False):
            return f"time-parallel fused-wide ({eng._wide_tp['nch']} chunks of {eng._wide_tp['chunk_len']} rows)"
        return "fused-wide"
    return "scaled" if eng.scaled else ("scaled-wide" if eng.scaled_wide else "v1")


def measure_cfg3(frac=1.0, jitter=False):
    import torch
    import gadfly_amd
    from gadfly_amd.synth import cfg3_light_curves
    B, N, J = max(1, int(256 * frac)), 65_000, 20
    hps, t, y, yerr, texp = cfg3_light_curves(B, N, J, jitter=jitter)
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr)
    ev.evaluate()                                   # warm-up + generator calibration
    dt, out = _clock(torch, lambda: _guarded(ev), reps=5, warm=1)
    ll = out.cpu().numpy()
    W = 2 * J
    gb = 8.0 * N * (3 * W + 4) * B / 1e9
    i = min(100, B - 1)
    path, period, reruns = _path(ev.engine), int(ev.engine.generator_period), int(ev.guard_reruns)
    extra = {}
    if not jitter:      # the same batch with +-0.2 s jitter on every other star's time stamps (exact rows)
        del ev
        torch.cuda.empty_cache()
        extra["jittered_stamps_ms"] = measure_cfg3(frac, jitter=True)["ms"]
    return {**extra, "workload": f"cfg3: {B} light curves x N={N}, J={J} (W={W}), own t/y/yerr/kernel, "
                        + ("every other star with jittered time stamps" if jitter else "uniform 58.85 s cadence"),
            "value": B / dt, "unit": "evals/s", "ms": 1e3 * dt, "algorithmic_GB": gb,
            "algorithmic_GBs": gb / dt, "frac": gb / dt / HBM_PEAK_GBS, "path": path,
            "generator_period": period, "guard_reruns": reruns,
            "all_finite": bool(np.all(np.isfinite(ll))),
            "_sample": dict(kind="loglike", index=i, coeffs=kernels[i].get_device_coefficients(),
                            t=t[i], diag=yerr[i] ** 2, y=y[i], got=float(ll[i]))}


def measure_cfg4(frac=1.0):
    import torch
    import gadfly_amd
    from gadfly_amd.synth import cfg4_walkers
    B, N, J = max(1, int(512 * frac)), 200_000, 40
    hps, t, y, texp = cfg4_walkers(B, N, J)
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
    ev.evaluate()
    dt, out = _clock(torch, lambda: _guarded(ev), reps=3, warm=0)
    ll = out.cpu().numpy()
    W = 2 * J
    gb = 8.0 * N * (3 * W + 4) * B / 1e9
    i = B - 1
    return {"workload": f"cfg4: {B} walkers x N={N}, J={J} (W={W}), shared t, y",
            "value": B / dt, "unit": "evals/s", "ms": 1e3 * dt, "algorithmic_GB": gb,
            "algorithmic_GBs": gb / dt, "frac": gb / dt / HBM_PEAK_GBS, "path": _path(ev.engine),
            "generator_period": int(ev.engine.generator_period), "guard_reruns": int(ev.guard_reruns),
            "all_finite": bool(np.all(np.isfinite(ll))),
            "_sample": dict(kind="loglike", index=i, coeffs=kernels[i].get_device_coefficients(),
                            t=t, diag=np.full(N, 900.0), y=y, got=float(ll[i]))}


def measure_cfg3_shard(full_ms=None):
    """What ONE of 8 GPUs runs of cfg3 (static block partition): 32 of the 256 light curves."""
    r = measure_cfg3(frac=1.0 / 8.0)
    r.pop("jittered_stamps_ms", None)
    r["workload"] = "cfg3 shard (1 of 8 GPUs): " + r["workload"][6:]
    if full_ms:
        r["predicted_speedup_8gpu"] = full_ms / r["ms"]
        r["predicted_speedup_note"] = "t(256 light curves on one GPU) / t(this 32-curve shard): strong scaling of cfg3 over 8 GPUs"
    return r


def measure_cfg4_shard(full_ms=None):
    """What ONE of 8 GPUs runs of cfg4: 64 of the 512 walkers (shared t, y)."""
    r = measure_cfg4(frac=1.0 / 8.0)
    r["workload"] = "cfg4 shard (1 of 8 GPUs): " + r["workload"][6:]
    if full_ms:
        r["predicted_speedup_8gpu"] = full_ms / r["ms"]
        r["predicted_speedup_note"] = "t(512 walkers on one GPU) / t(this 64-walker shard): strong scaling of cfg4 over 8 GPUs"
    return r


def measure_strong(which, dist, device, rank, world, backend, rows_scale=1.0):
    """cfg3 / cfg4 STRONG scaling: the whole batch (256 light curves / 512 walkers) partitioned over the ranks
    (static block partition, gadfly_amd.dist.shard_bounds), every rank evaluates its block on its own GPU
    and the B scalars are collected with one all_gather (RCCL over xGMI) -- the collective is inside the
    timed region.  Every rank calls this; returns the dict on every rank (timings MAX-reduced)."""
    import torch
    import gadfly_amd
    from gadfly_amd.dist import shard_bounds, gather_results
    from gadfly_amd.synth import cfg3_light_curves, cfg4_walkers
    if which == "cfg3":
        B, N, J = 256, max(1024, int(65_000 * rows_scale)), 20       # (rows_scale < 1: rehearsals only)
        hps, t, y, yerr, texp = cfg3_light_curves(B, N, J, jitter=False)
    else:
        B, N, J = 512, max(1024, int(200_000 * rows_scale)), 40
        hps, t, y, texp = cfg4_walkers(B, N, J)
        yerr = 30.0
    lo, hi = shard_bounds(B, world, rank)
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps[lo:hi]]
    if which == "cfg3":
        ev = gadfly_amd.BatchedLogLikelihood(kernels, t[lo:hi], y[lo:hi], yerr=yerr[lo:hi], device=device)
    else:
        ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr, device=device)
    ev.evaluate()                                   # warm-up + generator calibration

    def once():
        out = _guarded(ev)
        return gather_results(out.cpu().numpy(), B, device=device if backend == "nccl" else "cpu")

    once()
    times, full = [], None
    for _ in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = once()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                          device=device if backend == "nccl" else "cpu")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        times.append(float(dt.item()))
    dt = float(np.median(times))
    W = 2 * J
    gb = 8.0 * N * (3 * W + 4) * B / 1e9
    i = 0
    return {"workload": f"{which}: {B} x N={N}, J={J} (W={W}) partitioned over {world} GPUs ({hi - lo} on rank {rank}), "
                        "results all_gathered",
            "scaling": "strong", "n_gpus": world, "value": B / dt, "unit": "evals/s", "ms": 1e3 * dt,
            "algorithmic_GBs": gb / dt, "frac_of_aggregate_hbm": gb / dt / (HBM_PEAK_GBS * world),
            "path": _path(ev.engine), "all_finite": bool(np.all(np.isfinite(full))),
            "_sample": dict(kind="loglike", index=i, coeffs=gadfly_amd.StellarOscillatorKernel(
                                hps[i], texp=texp).get_device_coefficients(),
                            t=t[i] if which == "cfg3" else t, diag=(yerr[i] ** 2 if which == "cfg3"
                                                                    else np.full(N, 900.0)),
                            y=y[i] if which == "cfg3" else y, got=float(full[i]))}


def measure_cfg5():
    import torch
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    N, J, R = 500_000, 30, 64
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
    t = uniform_times(N, 60.0)
    gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
    eng = gp._engine                                # stored factor (time-parallel factorisation)
    np.random.seed(42)
    n = np.random.randn(N, R)
    nd = gp._to_device(n).reshape(1, N, R)
    dt, Z = _clock(torch, lambda: eng.dot_tril(nd), reps=5, warm=1)
    np.random.seed(42)
    t0 = time.perf_counter()
    draws = gp.sample(size=R)
    t_api = time.perf_counter() - t0
    W = 2 * J
    gb = 8.0 * N * (2 * W + 2 + 2 * R) / 1e9
    col = 17
    return {"workload": f"cfg5: dot_tril of {R} normal vectors, N={N}, J={J} (W={W}); vectors resident in HBM",
            "value": R / dt, "unit": "draws/s", "ms": 1e3 * dt, "algorithmic_GB": gb,
            "algorithmic_GBs": gb / dt, "frac": gb / dt / HBM_PEAK_GBS,
            "sample_api_ms": 1e3 * t_api,
            "sample_api_note": "gp.sample(size=64) end to end: numpy legacy randn on the host (the reference's "
                               "RNG contract) + PCIe both ways + the device work above",
            "draws_shape": list(draws.shape),
            "_sample": dict(kind="dot_tril", column=col, coeffs=k.get_device_coefficients(), t=t,
                            diag=np.full(N, 900.0), n=n[:, col].copy(),
                            got=Z[0, :, col].cpu().numpy())}


#: Time(0, format='bkjd') as the reference's GaussianProcess sees it: jd * day in units of 1e6 s
#: (/root/reference/gadfly/gp.py:79-80) -- the time axis of a lightkurve light curve or of the runtime-speed notebook
BKJD0 = 2454833.0 * 0.0864


def measure_cfg2_api(jd=False):
    import torch
    import gadfly_amd
    from gadfly_amd.synth import solar_like_hyperparameters, uniform_times
    N, J = 1_000_000, 30
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
    t = uniform_times(N, 60.0)
    if jd:
        # the same series on a JD-based axis (what `GaussianProcess(kernel, light_curve=lc)` hands over,
        # docs/gadfly/synth.rst:73-91): phases d t of 5e9 rad, still on the fused / time-parallel kernels
        t = BKJD0 + t
    rng = np.random.Generator(np.random.PCG64(12345))
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    gp = gadfly_amd.GaussianProcess(k)
    c_ms, _ = _clock(torch, lambda: gp.compute(t, yerr=30.0), reps=7, warm=2)
    l_ms, ll = _clock(torch, lambda: gp.log_likelihood(y), reps=7, warm=1)
    r_ms, _ = _clock(torch, lambda: (gp.recompute(), gp.log_likelihood(y)), reps=7, warm=1)
    _ = gp._engine
    p_ms, mu = _clock(torch, lambda: gp.predict(y), reps=7, warm=1)
    ts = np.sort(rng.uniform(t[0], t[-1], 1000))
    q_ms, mus = _clock(torch, lambda: gp.predict(y, t=ts), reps=5, warm=1)
    W = 2 * J
    gb_ll = 8.0 * N * (3 * W + 4) / 1e9
    gb_ai = 8.0 * N * (4 * W + 7) / 1e9
    idx = np.linspace(0, N - 1, 64).astype(int)
    # ONE chain through the batched evaluator with B = 1 (what a sampler that only needs log-likelihoods would call):
    # fresh hyperparameters per step, coefficient algebra + upload + the two-sweep time-parallel evaluation + readback
    from gadfly_amd.synth import jitter_hyperparameters
    ev1 = gadfly_amd.BatchedLogLikelihood([k], t, y, yerr=30.0)
    props = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 5000 + i),
                                                texp=60.0) for i in range(6)]
    ev1.evaluate([props[0]])
    e_ts = []
    for kk in props[1:]:
        t0 = time.perf_counter()
        ev1.evaluate([kk])
        e_ts.append(time.perf_counter() - t0)
    e_ms = float(np.median(e_ts))
    del ev1
    torch.cuda.empty_cache()
    first = None
    if not jd:
        # the FIRST compute() of a process, reported apart from the warm latency (a child process: code-object load,
        # LDS opt-ins, first allocations)
        import json
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "first_call.py")],
                               capture_output=True, text=True, timeout=300)
            first = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:      # noqa: BLE001  (a diagnostic leg: never fails the run)
            first = {"error": repr(e)[:200]}
    return {"first_call": first,
            "workload": f"cfg2 through the drop-in GaussianProcess: N={N}, J={J} (W={W}), ONE series "
                        "(latency path: exact time-parallel factorisation / sweeps); host arrays in, "
                        "host arrays out (PCIe included)"
                        + ("; time axis = BKJD 0 + n minutes as jd * day (t ~ 2.12e5, phases to 5e9 rad)" if jd else ""),
            "route": type(gp._factor).__name__,
            "evaluator_b1_ms": 1e3 * e_ms,
            "compute_ms": 1e3 * c_ms, "log_likelihood_ms": 1e3 * l_ms,
            "recompute_plus_log_likelihood_ms": 1e3 * r_ms,
            "predict_mean_ms": 1e3 * p_ms, "predict_1000_new_times_ms": 1e3 * q_ms,
            "compute_plus_loglike_algorithmic_GBs": gb_ll / (c_ms + l_ms),
            "predict_mean_algorithmic_GBs": gb_ai / p_ms,
            "predict_mean_frac": gb_ai / p_ms / HBM_PEAK_GBS,
            "_sample": dict(kind="predict", coeffs=k.get_device_coefficients(), t=t, diag=np.full(N, 900.0),
                            y=y, ts=ts, idx=idx, got_ll=float(ll), got_mu=mu[idx].copy(), got_mus=mus.copy())}


def measure_jd_batch():
    """The headline's workload on a JD-based time axis (BKJD 0 + n minutes, phases of 5e9 rad), at a length that
    keeps the leg short: 2048 walkers x N = 131 072, J = 30, streamed fused sweep, at the generator period the
    evaluator calibrates itself to and with exact rows -- next to the same batch on the zero-based axis.  Until
    round 4 such an axis ran on the materialised-row kernels (phases beyond the in-kernel sincos) and, once fused,
    needed exact rows (the phase quantum)."""
    import torch
    import gadfly_amd
    from gadfly_amd.synth import jitter_hyperparameters, solar_like_hyperparameters, uniform_times
    B, N, J = 2048, 131_072, 30
    base = solar_like_hyperparameters(J)
    kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 1000 + i), texp=60.0) for i in range(B)]
    rng = np.random.Generator(np.random.PCG64(12345))
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    out = {}
    for name, t0 in (("jd_axis", BKJD0), ("zero_based", 0.0)):
        t = t0 + uniform_times(N, 60.0)
        ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
        ev.engine.force_streaming = True
        ev.evaluate()                               # warm-up + generator calibration
        dt, res = _clock(torch, lambda: _guarded(ev), reps=3, warm=1)
        period = int(ev.engine.generator_period)
        ev.auto_generator_period = False
        ev.engine.generator_period = 1
        dt1, _ = _clock(torch, lambda: _guarded(ev), reps=2, warm=1)
        out[name] = {"ms": 1e3 * dt, "generator_period": period, "row_evaluations_per_s": B * N / dt,
                     "exact_rows_ms": 1e3 * dt1, "path": _path(ev.engine)}
        if name == "jd_axis":
            ll = res.cpu().numpy()
            i = B - 1
            sample = dict(kind="loglike", index=i, coeffs=kernels[i].get_device_coefficients(), t=t,
                          diag=np.full(N, 900.0), y=y, got=float(ll[i]))
        del ev
        torch.cuda.empty_cache()
    return {"workload": f"{B} walkers x N={N}, J={J} (W={2 * J}) streamed, on t = BKJD 0 + n minutes as jd * day "
                        "(phases to 5e9 rad) and on the zero-based axis",
            **out, "jd_over_zero_based": out["jd_axis"]["ms"] / out["zero_based"]["ms"], "_sample": sample}


def measure_runtime_speed():
    """The reference's own performance harness (/root/reference/notebooks/paper/runtime-speed.ipynb:40-42, :77):
    ``SolarOscillatorKernel()`` (86 terms, W = 172), eight durations 0.1 ... 1000 d at one-minute cadence on
    ``linspace(0, D, n) d + Time(0, format='bkjd')``, ``gp = GaussianProcess(kernel, t=new_times)``, then
    ``%timeit gp.sample()`` -- the construction is outside the timing there and here -- and "years of observations
    per second" = interp(1 s, runtimes, durations).  astropy / tynt are not installed: the kernel takes the bolometric
    SOHO VIRGO bandpass (alpha = 1) instead of the default Kepler one, the axis is formed as jd * 0.0864 directly.
    ``_samples``: inputs and the GPU's draw (seed 42) of the sizes an oracle run finishes in seconds; bench.py times
    the same call on a host core and checks the draw there (this module never touches oracle/)."""
    import warnings
    import torch
    import gadfly_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
    co = k.get_device_coefficients()
    durations = np.logspace(-1, 3, 8)               # days
    rows, samples = [], []
    for D in durations:
        n = int(D * 1440.0)
        t = (2454833.0 + np.linspace(0.0, D, n)) * 0.0864
        c0 = time.perf_counter()
        gp = gadfly_amd.GaussianProcess(k, t=t)
        _ = gp._engine
        torch.cuda.synchronize()
        construct = time.perf_counter() - c0

        def once():
            np.random.seed(42)                      # (so that the draw handed to the check is randn(n) after seed 42)
            return gp.sample()

        dt, draw = _clock(torch, once, reps=3, warm=1)
        rows.append({"duration_d": float(D), "N": n, "sample_ms": 1e3 * dt, "construct_ms": 1e3 * construct,
                     "route": type(gp._factor).__name__
                              + (f" ({gp._factor.nch} chunks)" if getattr(gp._factor, "nch", 1) > 1 else "")})
        if n <= 110_000:
            samples.append(dict(N=n, t=t, coeffs=co, draw=draw))
        del gp
        torch.cuda.empty_cache()
    secs = np.array([r["sample_ms"] for r in rows]) * 1e-3
    days = np.array([r["duration_d"] for r in rows])
    per_s = float(np.interp(1.0, secs, days)) / 365.25         # the notebook's statement (clamps at 1000 d)
    slope = (days[-1] - days[-2]) / (secs[-1] - secs[-2])       # days of observations per second of runtime
    return {"workload": "runtime-speed.ipynb: gp.sample() of SolarOscillatorKernel (86 terms, W = 172) on a BKJD "
                        "axis, eight durations at one-minute cadence; construction outside the timing as in the "
                        "notebook; host randn (the reference's RNG contract) and PCIe both ways inside it",
            "durations": rows,
            "years_of_observations_per_second": per_s,
            "years_per_second_note": "np.interp(1 s, runtimes, durations) as the notebook prints it; the longest run "
                                     f"(1000 d = 2.74 yr) takes {secs[-1] * 1e3:.1f} ms, so the statement clamps "
                                     "there; extrapolated with the last two sizes' slope: "
                                     f"{(days[-1] + slope * (1.0 - secs[-1])) / 365.25:.0f} yr",
            "largest_ms": float(secs[-1] * 1e3),
            "_samples": samples}


def main():
    import json
    which = [a for a in sys.argv[1:] if not a.startswith("--so=")]
    for a in sys.argv[1:]:
        if a.startswith("--so="):               # A/B builds of the library (development)
            from gadfly_amd import _lib
            _lib.SO_PATH = os.path.abspath(a[5:])
    fns = dict(cfg2=measure_cfg2_api, cfg2jd=lambda: measure_cfg2_api(jd=True), cfg3=measure_cfg3, cfg4=measure_cfg4,
               cfg5=measure_cfg5, cfg3s=measure_cfg3_shard, cfg4s=measure_cfg4_shard, runtime=measure_runtime_speed,
               jdbatch=measure_jd_batch)
    for fn in ([fns[w] for w in which] if which else fns.values()):
        r = fn()
        r.pop("_sample", None)
        r.pop("_samples", None)
        print(json.dumps(r))


if __name__ == "__main__":
    main()
