#!/usr/bin/env python
"""Throughput of the batched log-likelihood path on BASELINE.json's batched configs (per GPU share):
cfg3: 256 Kepler-cadence light curves (N=65,000, J=20)    -- own t, y and kernel per light curve
cfg4: 512 walkers on one N=200,000, J=40 series           -- shared t, y
Usage: python tools/configs.py [fraction]   (fraction of the full batch evaluated on this GPU, default 1)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import (solar_like_hyperparameters, jitter_hyperparameters,  # noqa: E402
                              scale_hyperparameters)

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rng = np.random.Generator(np.random.PCG64(12345))


def run(name, ev, B):
    ev.evaluate_device(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = ev.evaluate_device(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    print(f"{name}: B={B} in {dt*1e3:8.1f} ms -> {B/dt:9.1f} evals/s "
          f"(kernels: {'fused' if ev.engine._fused_ok() else ('scaled' if ev.engine.scaled else ('scaled-wide' if ev.engine.scaled_wide else 'v1'))})")


# cfg3
B, N, J = max(1, int(256 * frac)), 65_000, 20
base = solar_like_hyperparameters(J)
kernels = [gadfly_amd.StellarOscillatorKernel(scale_hyperparameters(base, f), texp=58.85)
           for f in np.geomspace(0.3, 1.0, B)]
t = np.tile(np.arange(N) * 58.85e-6, (B, 1))
y = rng.normal(size=(B, N)) * 50.0
run("cfg3 (N=65000, J=20)", gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0), B)
# cfg4
B, N, J = max(1, int(512 * frac)), 200_000, 40
base = solar_like_hyperparameters(J)
kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 1000 + i), texp=60.0)
           for i in range(B)]
t = np.arange(N) * 60e-6
y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
run("cfg4 (N=200000, J=40)", gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0), B)
