import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gadfly_amd
from gadfly_amd import _lib
from gadfly_amd.synth import cfg4_walkers
N, J = 65536, 40
for B in (256, 512):
    hps, t, y, texp = cfg4_walkers(B, N, J)
    kernels = [gadfly_amd.StellarOscillatorKernel(hp, texp=texp) for hp in hps]
    ev = gadfly_amd.BatchedLogLikelihood(kernels, t, y, yerr=30.0)
    ev.engine.force_streaming = True
    ev.engine.generator_period = 64
    ev.auto_generator_period = False
    for _ in range(2):
        out = ev.evaluate_device()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.SO_PATH)
    o = (ctypes.c_double * 8)()
    lib.gf_debug_read_w(o, 8)
    ts = np.array(o[:8])
    rows = ts[5]
    names = ["sweep (setprio 0)", "prefetch + reductions + r", "LDS writes drained", "barrier wait", "partials, rcp, q, wb loads"]
    print("   resets %.0f, cycles in reset blocks %.0f" % ((ts[5] % 1) * 1e6, ts[7]))
    print("B=%d: last tile: %d rows, %.0f cycles per row in the loop (%.0f incl. resets)" % (B, rows, ts[:5].sum() / rows, ts[6] / rows))
    for k in range(5):
        print("   %-28s %7.0f cycles  %5.1f%%" % (names[k], ts[k] / rows, 100 * ts[k] / ts[:5].sum()))
