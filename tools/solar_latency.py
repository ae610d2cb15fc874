#!/usr/bin/env python
"""Latency of the drop-in GaussianProcess with gadfly's default kernel (SolarOscillatorKernel: 86 SHO
terms, W = 172) -- what /root/reference/gadfly/tests/test_core.py:23-29 and
notebooks/paper/runtime-speed.ipynb:40-42 run.  Usage: python tools/solar_latency.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402


def clock(fn, reps=2):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out


def main():
    k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
    W = len(k.get_device_coefficients()[2]) * 2
    for N in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]:
        t = np.linspace(0, 100, N) * 0.0864 * (N / 1e5)        # 100 d per 1e5 points, in 1e6 s
        y = np.random.default_rng(0).normal(size=N) * 50
        gp = gadfly_amd.GaussianProcess(k)
        ms, _ = clock(lambda: gp.compute(t), reps=2)
        print(f"SolarOscillatorKernel W={W} N={N}: compute          {ms:9.1f} ms")
        ms, _ = clock(lambda: gp.log_likelihood(y))
        print(f"SolarOscillatorKernel W={W} N={N}: log_likelihood   {ms:9.1f} ms")
        ms, _ = clock(lambda: gp.predict(y))
        print(f"SolarOscillatorKernel W={W} N={N}: predict(y)       {ms:9.1f} ms")
        np.random.seed(1)
        ms, _ = clock(lambda: gp.sample())
        print(f"SolarOscillatorKernel W={W} N={N}: sample()         {ms:9.1f} ms")
        del gp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
