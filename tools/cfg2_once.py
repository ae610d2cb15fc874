#!/usr/bin/env python
"""cfg2 through the drop-in class, one phase at a time, for kernel profiles (development):
python tools/cfg2_once.py compute|predict|loglike [N] [J] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "compute"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
J = int(sys.argv[3]) if len(sys.argv) > 3 else 30
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
t = uniform_times(N, 60.0)
rng = np.random.Generator(np.random.PCG64(12345))
y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
_ = gp._engine
fn = {"compute": lambda: gp.compute(t, yerr=30.0), "predict": lambda: gp.predict(y),
      "loglike": lambda: gp.log_likelihood(y), "recompute": lambda: (gp.recompute(), gp.log_likelihood(y))}[what]
fn()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"{what} N={N} J={J}: median {1e3 * np.median(ts):.3f} ms  min {1e3 * min(ts):.3f} ms  ({reps} reps + 1 warm-up + construction)")
