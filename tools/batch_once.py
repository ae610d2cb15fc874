#!/usr/bin/env python
"""One streamed batched log-likelihood evaluation of B walkers (shared t, y) -- the process profiled by
the rocprofv3 counter passes of the wide kernels.  Usage: python tools/batch_once.py N J B [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import cfg4_walkers  # noqa: E402

N, J, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
hps, t, y, texp = cfg4_walkers(B, N, J)
kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
ev.engine.force_streaming = os.environ.get("ROUTE", "stream") == "stream"     # ROUTE=auto: the product's choice
if os.environ.get("CHUNK_LEN"):                 # time-parallel route with a given chunk length
    ev.engine.force_streaming = False
    ev.engine.wide_tp_min_rows = 0
    ev.engine._wide_tp_ok = lambda: True        # (whatever the batch size)
    _len = int(os.environ["CHUNK_LEN"])
    ev.engine._wide_chunking = (lambda chunk_len, _orig=ev.engine._wide_chunking: _orig(_len))
ev.engine.generator_period = int(os.environ.get("GEN_PERIOD", "64"))
ev.auto_generator_period = False
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ev.evaluate_device()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"N={N} J={J} B={B}: {dt*1e3:.2f} ms  {dt/N*1e6:.3f} us/row  kernel={ev.engine.kernel_used if hasattr(ev.engine, 'kernel_used') else '?'}  "
      f"finite={bool(torch.isfinite(out).all())}")
