#!/usr/bin/env python
"""cfg3's batch (256 light curves x 65 000, J = 20) through the time-parallel route at given chunk lengths:
python tools/cfg3_chunks.py 8128 7232 ...   (development: where the chunk count should sit)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import cfg3_light_curves  # noqa: E402

B, N, J = int(os.environ.get("B", 256)), 65_000, 20
hps, t, y, yerr, texp = cfg3_light_curves(B, N, J, jitter=False)
kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=yerr)
ev.evaluate()
eng = ev.engine
# two sweeps (nominal pass over all chunks + transitions of chunks 1 ... nch - 1) as
# BatchedLogLikelihood runs it, or TWO_SWEEP=0: three sweeps with the final pass
eng.two_sweep = os.environ.get("TWO_SWEEP", "1") != "0"
print("period", eng.generator_period, "two_sweep", eng.two_sweep)
for L in [int(a) for a in sys.argv[1:]]:
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.log_likelihood_time_parallel(chunk_len=L)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"chunk_len={L} nch={eng._tp_key[1]}: {1e3 * dt:.2f} ms")
