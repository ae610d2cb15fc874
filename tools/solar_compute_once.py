#!/usr/bin/env python
"""`compute` of the drop-in GaussianProcess with the reference's default kernel (SolarOscillatorKernel,
W = 172), a few times, for profiles: python tools/solar_compute_once.py [N] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
t = np.linspace(0, 100, N) * 0.0864 * (N / 1e5)
if os.environ.get("WIDE_TP_COEF"):              # chunk count of the time-parallel factorisation ~ sqrt(coef N)
    from gadfly_amd.engine import StreamingBatch
    StreamingBatch.WIDE_TP_COEF = float(os.environ["WIDE_TP_COEF"])
gp = gadfly_amd.GaussianProcess(k)
for i in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.compute(t)
    torch.cuda.synchronize()
    print(f"compute N={N}: {1e3 * (time.perf_counter() - t0):.2f} ms")
