#!/usr/bin/env python
"""How the generator period interacts with the phase quantum of a time axis far from zero (development):
celerite2 (and the oracle) form theta = fl(d t), which carries half an ulp of the PHASE as rounding -- 1e-6 rad at
the 5e9 rad of a JD-based axis -- while rotation steps between anchors follow the true phase.  Prints the
relative log-likelihood error against the oracle for periods 1 / 4 / 64 over a ladder of time offsets, with the
model terms GEN_ERR * period * cond and quantum * cond / sqrt(N).   python tools/phase_quantum.py [J]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from gadfly_amd.engine import StreamingBatch  # noqa: E402
from oracle import cref  # noqa: E402

J = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for N in (16384, 131072):
    for yerr in (30.0, 0.0):
        prob = util.solar_problem(J, N, yerr=yerr)
        co = prob["kernel"].get_device_coefficients()
        for off in (0.0, 1.0e3, 1.0e4, 1.0e5, 2.12e5, 2.0e6):
            t = prob["t"] + off
            ref, info = cref.loglike(co[:6], t, prob["diag_user"] + co[6], prob["y"])
            eng = StreamingBatch([co], t, prob["y"], diag=prob["diag_user"])
            errs = []
            for period in (1, 4, 64):
                eng.generator_period = period
                ll = float(eng.log_likelihood()[0])
                errs.append(abs(ll - ref) / abs(ref))
            cond = eng.condition_estimate()
            q = eng._pack[6] * eng._tmax * 2.0 ** -53
            print(f"J={J} N={N:7d} yerr={yerr:4.0f} off={off:9.3g} phase={eng._pack[6] * eng._tmax:9.3g} cond={cond:9.3g} "
                  f"err p1={errs[0]:.1e} p4={errs[1]:.1e} p64={errs[2]:.1e} | q*cond/sqrtN={q * cond / np.sqrt(N):.1e} "
                  f"q*cond={q * cond:.1e} gen64={1.6e-15 * 64 * cond:.1e}", flush=True)
