#!/usr/bin/env python
"""Two more randomized sweeps (development, next to tools/random_sweep.py):
  jd    FIRST COUNT   tests/test_gpu_random.py::test_random_problem with every time axis moved to t + 2.12e5 (a JD-based
                      axis: phases far beyond 4e6 rad, the row generator's rounded-phase steps -- RowGen::qmode -- on)
  wide  FIRST COUNT   random WIDE kernels (33 ... 88 complex terms, W = 66 ... 176), two series each: streamed sweep,
                      three-sweep and two-sweep time-parallel evaluation (the latter with the pivot-sign check of every
                      chunk: a false alarm shows up as a non-finite value) against the oracle at 1e-8"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random as T  # noqa: E402
from gadfly_amd import _lib as hip  # noqa: E402

what, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
hip.require_device()
bad, skipped = [], 0

if what == "jd":
    base = T._problem

    def moved(seed):
        p = base(seed)
        p["t"] = p["t"] + 2.12e5
        return p

    T._problem = moved
    for seed in range(first, first + count):
        try:
            T.test_random_problem(hip, seed)
        except pytest.skip.Exception:
            skipped += 1
        except Exception as e:      # noqa: BLE001
            bad.append(seed)
            print("FAIL", seed, repr(e)[:300], flush=True)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} seeds, {len(bad)} failures, {skipped} skipped", flush=True)
else:
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
    from oracle import cref
    for seed in range(first, first + count):
        rng = np.random.Generator(np.random.PCG64(seed))
        J = int(rng.integers(33, 89))
        terms = [SHOTerm(S0=float(np.exp(rng.uniform(-2, 4))), w0=float(np.exp(rng.uniform(np.log(0.5), np.log(3000.0)))),
                         Q=float(np.exp(rng.uniform(np.log(0.5), np.log(300.0))))) for _ in range(J)]
        dt = float(np.exp(rng.uniform(np.log(2e-5), np.log(2e-3))))
        N = int(rng.integers(1500, 5000))
        t = np.arange(N) * dt
        kind = rng.choice(["uniform", "jitter", "gaps"])
        if kind == "jitter":
            t = np.sort(t + rng.uniform(-0.3, 0.3, N) * dt)
        elif kind == "gaps":
            keep = np.ones(N, bool)
            a = int(rng.integers(0, N - 1)); keep[a:a + int(rng.integers(1, N // 8))] = False
            keep[0] = True
            t = t[keep]
        N = len(t)
        k = TermConvolution(TermSum(*terms), float(rng.uniform(0.1, 1.0)) * dt)
        yerr = 0.0 if rng.random() < 0.2 else float(np.exp(rng.uniform(-3, 2)))
        amp = float(np.sqrt(k.get_value(np.zeros(1))[0]))
        y = amp * rng.normal(size=N) + np.cumsum(rng.normal(size=N)) * 0.1 * amp
        co = k.get_device_coefficients()
        du = np.full(N, yerr ** 2)
        ref, info = cref.loglike(co[:6], t, du + co[6], y)
        if info != 0:
            skipped += 1
            continue
        eng = StreamingBatch([co, co], t, y, diag=du, tile_rows=1024)
        eng.generator_period = 1
        L = int(rng.choice([192, 256, 640, 1024]))
        try:
            res = {"streamed": eng.log_likelihood().cpu().numpy()}
            cond = eng.condition_estimate()
            if cond > 1e7:
                skipped += 1
                continue
            for two in (False, True):
                eng.two_sweep = two
                res["two-sweep" if two else "three-sweep"] = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
            for name, v in res.items():
                rel = float(np.max(np.abs(v - ref)) / abs(ref))
                if not rel <= 1e-8:
                    raise AssertionError((seed, name, J, N, L, kind, rel, cond))
        except Exception as e:      # noqa: BLE001
            bad.append(seed)
            print("FAIL", seed, repr(e)[:300], flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} seeds, {len(bad)} failures, {skipped} skipped", flush=True)
print(f"{what}: {count} seeds from {first}: {len(bad)} failures, {skipped} skipped")
sys.exit(1 if bad else 0)
