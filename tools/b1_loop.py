"""One chain of single evaluations through BatchedLogLikelihood (B = 1, N = 1e6, J = 30): the process profiled for
profiles/r04_b1_evaluator_kernel_stats.csv (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gadfly_amd
from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters, uniform_times
N, J = 1_000_000, 30
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
t = uniform_times(N, 60.0)
y = np.random.default_rng(1).normal(size=N) * 50.0
ev = gadfly_amd.BatchedLogLikelihood([k], t, y, yerr=30.0)
props = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 5000 + i), texp=60.0) for i in range(20)]
for kk in props:
    ev.evaluate([kk])
print("period", ev.engine.generator_period, "chunk", getattr(ev.engine, "_tp_chunk_len", None), "two", ev.engine._two_sweep_used)
