#!/usr/bin/env python
"""Single-evaluation latency of the log-likelihood path (B = 1): sequential streamed sweep vs the
exact time-parallel evaluation, for several chunk lengths.  Usage: python tools/latency.py [N] [J]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.engine import StreamingBatch  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
J = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.Generator(np.random.PCG64(12345))
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
t = uniform_times(N, 60.0)
y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
eng = StreamingBatch([k.get_device_coefficients()] * B, t, y, diag=np.full(N, 900.0))


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


dt, ll0 = timeit(eng.log_likelihood, reps=1)
print(f"N={N} J={J} B={B} sequential streamed sweep: {dt*1e3:9.2f} ms  ll={float(ll0[0]):.12e}")
dt, ll = timeit(lambda: eng.log_likelihood_time_parallel())
print(f"  time-parallel, default chunking ({eng._tp_chunking(None)[0]} rows x {eng._tp_chunking(None)[1]}): "
      f"{dt*1e3:9.2f} ms  ({B/dt:8.1f} evals/s)")
for L in [int(x) for x in os.environ.get("TP_CHUNKS", "32768,16384,8192,4096,2048,1024,512").split(",")]:
    if L >= N:
        continue
    dt, ll = timeit(lambda: eng.log_likelihood_time_parallel(chunk_len=L))
    rel = abs(float(ll[0]) - float(ll0[0])) / abs(float(ll0[0]))
    print(f"  time-parallel chunk_len={L:6d} nch={-(-N//L):5d}: {dt*1e3:9.2f} ms  "
          f"({B/dt:8.1f} evals/s)  rel diff vs sequential {rel:.1e}")
