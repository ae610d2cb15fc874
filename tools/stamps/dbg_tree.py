import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gadfly_amd
from gadfly_amd import _lib
from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters, uniform_times
def run(B, N, J):
    ks = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 100 + i), texp=60.0) for i in range(B)]
    t = uniform_times(N, 60.0)
    y = np.random.default_rng(1).normal(size=N) * 50.0
    ev = gadfly_amd.BatchedLogLikelihood(ks, t, y, yerr=30.0)
    for _ in range(3):
        ev.evaluate()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.SO_PATH)
    out = (ctypes.c_double * 16)()
    lib.gf_debug_read(out, 16)
    ts = np.array(out[:13])
    names = ["loads a", "product a", "GJ", "g2v", "product c", "load Phi2", "2 products d", "S rmw", "load Phi1", "2 products e+store", "2 products f+m", "G store"]
    print("B=%d N=%d W=%d: top-level compose %d cycles" % (B, N, 2 * J, ts[12]))
    for k, nm in enumerate(names):
        print("   %-20s %8.0f  %5.1f%%" % (nm, ts[k + 1] - ts[k], 100 * (ts[k + 1] - ts[k]) / ts[12]))
run(1, 1_000_000, 30)
run(32, 65_000, 20)
