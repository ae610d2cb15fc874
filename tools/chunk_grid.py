#!/usr/bin/env python
"""Two-sweep time-parallel log-likelihood in ms over a grid of batch sizes, lengths, widths and chunk counts
(B * nch ~ 512 / 1024 / 2048 / 4096 waves): the grid behind engine._tp_chunking's rule.  Development:
python tools/chunk_grid.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gadfly_amd
from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters, uniform_times
def run(B, N, J, waves):
    ks = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 100 + i), texp=60.0) for i in range(B)]
    t = uniform_times(N, 60.0)
    y = np.random.default_rng(1).normal(size=N) * 50.0
    ev = gadfly_amd.BatchedLogLikelihood(ks, t, y, yerr=30.0)
    for _ in range(3):
        ev.evaluate()
    eng = ev.engine
    eng.two_sweep = True
    res = []
    for wv in waves:
        nch = max(1, wv // B)
        ch = max(256, -(-N // nch))
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = eng.log_likelihood_time_parallel(chunk_len=ch); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        nch = eng._tp["info"].numel() // B
        res.append("%d:%dx%d %.2f" % (wv, nch, -(-ch // 64) * 64, np.median(ts[2:])))
    print("B=%d N=%d W=%d  " % (B, N, 2 * J) + " | ".join(res), flush=True)
for J in (30, 20, 10):
    for B, N in ((1, 250_000), (1, 500_000), (1, 1_000_000), (1, 2_000_000), (1, 4_000_000), (8, 65_000), (16, 65_000), (32, 65_000), (64, 65_000)):
        run(B, N, J, [512, 1024, 2048, 4096])
