#!/usr/bin/env python
"""Latency of the drop-in GaussianProcess API on BASELINE.json's single-series configs:
cfg2 (N=1e6, J=30: compute, log_likelihood, predict) and cfg5 (N=5e5, J=30: sample(size=64)).
(The CPU side of the comparison is bench.py's cpu_baseline leg.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402


def clock(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


def main():
    J = 30
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
    rng = np.random.Generator(np.random.PCG64(12345))
    # ---- cfg2 ----
    N = 1_000_000
    t = uniform_times(N, 60.0)
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    gp = gadfly_amd.GaussianProcess(k)
    ms, _ = clock(lambda: gp.compute(t, yerr=30.0), reps=2)
    print(f"cfg2 N={N} J={J}: compute              {ms:9.1f} ms")
    ms, ll = clock(lambda: gp.log_likelihood(y))
    print(f"cfg2 N={N} J={J}: log_likelihood       {ms:9.1f} ms   ll={ll:.10e}")
    ms0, _ = clock(lambda: gp._engine, reps=1)
    ms, mu = clock(lambda: gp.predict(y))
    print(f"cfg2 N={N} J={J}: predict(y)           {ms:9.1f} ms   (first call also builds the stored factor)")
    ts = np.sort(rng.uniform(t[0], t[-1], 1000))
    ms, mus = clock(lambda: gp.predict(y, t=ts), reps=2)
    print(f"cfg2 N={N} J={J}: predict(y, t*=1000)  {ms:9.1f} ms")
    del gp
    torch.cuda.empty_cache()
    # ---- cfg5 ----
    N = 500_000
    t = uniform_times(N, 60.0)
    gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
    _ = gp._engine
    np.random.seed(42)
    ms, s = clock(lambda: gp.sample(size=64), reps=2)
    print(f"cfg5 N={N} J={J}: sample(size=64)      {ms:9.1f} ms   (includes host randn + transfers)")
    n = torch.randn((1, N, 64), dtype=torch.float64, device="cuda")
    ms, _ = clock(lambda: gp._engine.dot_tril(n))
    print(f"cfg5 N={N} J={J}: dot_tril on device   {ms:9.1f} ms   "
          f"({8*N*(2*2*J+2+2*64)/ (ms*1e-3)/1e9:7.1f} GB/s algorithmic)")


if __name__ == "__main__":
    main()
