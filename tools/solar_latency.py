#!/usr/bin/env python
"""The drop-in classes with gadfly's own default kernel: SolarOscillatorKernel (86 SHO terms, celerite
width 172) -- the general-width kernels (k_build + k_factor, k_solve_*).  compute / log_likelihood /
predict(y) / sample.  Usage: python tools/solar_latency.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
kernel = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
t = np.arange(N) * 60e-6
rng = np.random.default_rng(0)
y = 100 * rng.normal(size=N) + np.cumsum(rng.normal(size=N))


def timed(label, fn, reps=2):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"SolarOscillatorKernel W={len(kernel)} N={N}: {label:28s} {best*1e3:9.1f} ms", flush=True)
    return out


gp = gadfly_amd.GaussianProcess(kernel, t=t, yerr=30.0)
timed("compute", lambda: gp.compute(t, yerr=30.0))
timed("log_likelihood", lambda: gp.log_likelihood(y))
timed("predict(y)", lambda: gp.predict(y))
np.random.seed(1)
timed("sample()", lambda: gp.sample())
