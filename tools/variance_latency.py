#!/usr/bin/env python
"""predict(y, t*, return_var=True) at the benchmark's size: N = 1e6, J = 30, M = 300 new times
(docs/gadfly/synth.rst:193-201 uses M = 300 for gap filling).  Usage: python tools/variance_latency.py [N] [M]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.Generator(np.random.PCG64(7))
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(30), texp=60.0)
t = uniform_times(N, 60.0)
y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
ts = np.sort(rng.uniform(t[N // 3], t[N // 3] + 300 * 60e-6, M))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu, var = gp.predict(y, t=ts, return_var=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N} M={M}: predict(y, t*, return_var=True) {dt*1e3:8.1f} ms   "
          f"var in [{var.min():.4g}, {var.max():.4g}] (prior {k.get_value(np.zeros(1))[0]:.4g})")
