import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gadfly_amd
from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters, uniform_times
def run(B, N, J):
    ks = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 100 + i), texp=60.0) for i in range(B)]
    t = uniform_times(N, 60.0)
    y = np.random.default_rng(1).normal(size=N) * 50.0
    ev = gadfly_amd.BatchedLogLikelihood(ks, t, y, yerr=30.0)
    for _ in range(3):
        ev.evaluate()
    eng = ev.engine
    w = eng._tp
    nch = w["info"].numel() // B
    corr = w["corr"].cpu().numpy()
    nblk = B * (nch - 1)
    # maps are [B*nch] (not padded P) in the engine's slots
    n = B * nch
    dbg = corr[2 * n: 2 * n + 16 * nblk].reshape(nblk, 16)
    d = np.diff(np.concatenate([np.zeros((nblk, 1)), dbg[:, :9]], axis=1), axis=1)
    names = ["load", "e/diag", "pchol", "G,T=GR", "M", "chol", "reload", "A=I-XG", "LU", ]
    print("B=%d N=%d J=%d nch=%d maps=%d rank median %.0f min %.0f max %.0f" % (B, N, J, nch, nblk, np.median(dbg[:, 9]), dbg[:, 9].min(), dbg[:, 9].max()))
    names = ["load X", "e", "pchol", "G,w1,T", "M", "chol", "reload+A=I-XG", "GJ", "finish"]
    tot = dbg[:, 8]
    print("  total cycles median %.0f  (first-round blocks %.0f, last %.0f)" % (np.median(tot), np.median(tot[:256]), np.median(tot[-256:])))
    for k, nm in enumerate(names):
        print("  %-9s %9.0f  %5.1f%%" % (nm, np.median(d[:, k]), 100 * np.median(d[:, k]) / np.median(tot)))
run(1, 1_000_000, 30)
run(32, 65_000, 20)
