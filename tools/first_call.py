#!/usr/bin/env python
"""What the FIRST compute() of a process costs next to a warm one (code-object load, LDS opt-ins, the allocator's
first blocks, page-locked staging buffers): run in a fresh process, prints one JSON line.
python tools/first_call.py [N] [J]"""
import json
import os
import sys
import time

t0 = time.perf_counter()
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
J = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t_import = time.perf_counter() - t0
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(J), texp=60.0)
t = uniform_times(N, 60.0)
gp = gadfly_amd.GaussianProcess(k)
out = []
for _ in range(3):
    c0 = time.perf_counter()
    gp.compute(t, yerr=30.0)
    torch.cuda.synchronize()
    out.append(1e3 * (time.perf_counter() - c0))
print(json.dumps({"N": N, "J": J, "imports_and_device_init_ms": 1e3 * t_import, "first_compute_ms": out[0],
                  "second_compute_ms": out[1], "third_compute_ms": out[2]}))
