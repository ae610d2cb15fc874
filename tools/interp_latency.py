#!/usr/bin/env python
"""interpolate_missing_data on a Kepler-like series (2e6 short cadences, 10 % missing, cadence numbers
given): device kernels vs the numpy formulation of the reference (gadfly/interp.py:6-60) on the host.
Usage: python tools/interp_latency.py [n_full]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402


def host_interpolate(times, fluxes, cadences):
    """The reference's numpy formulation (gadfly/interp.py:6-60), timed as the CPU side."""
    t0 = times[0]
    dt = np.median(np.diff(times) / np.diff(cadences))
    index = cadences - cadences[0]
    missing = np.setdiff1d(np.arange(index.min(), index.max()), index)
    t_new = t0 + missing * dt
    f_new = np.interp(t_new, times, fluxes)
    t_all, f_all = np.concatenate([times, t_new]), np.concatenate([fluxes, f_new])
    order = np.argsort(t_all, kind="stable")
    return t_all[order], f_all[order]


n_full = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(1)
keep = rng.uniform(size=n_full) > 0.1
keep[[0, -1]] = True
cad = np.flatnonzero(keep) + 7000
t = 2454833.0 + cad * (58.85 / 86400.0) + 0.5 * rng.uniform(-1, 1, cad.size) / 86400.0
f = 1e4 + 50 * np.sin(cad * 0.01) + rng.normal(size=cad.size)

best = 1e30
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    td, fd = gadfly_amd.interpolate_missing_data(t, f, cadences=cad, return_device=True)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
# kernels only (inputs resident): HIP events around plan + fill
lib, p = gadfly_amd._lib.load(), gadfly_amd._lib.ptr
t_d, f_d, c_d = (torch.as_tensor(a, device="cuda") for a in (t, f, cad))
n = len(t)
dt = float(np.median(np.diff(t) / np.diff(cad)))
off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
work = torch.empty(int(lib.gf_interp_work(n)), dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    lib.gf_interp_plan(n, p(t_d), p(c_d), dt, p(off), p(work), st)
    lib.gf_interp_fill(n, p(t_d), p(f_d), p(c_d), dt, p(off), p(td), p(fd), st)
ev[1].record(); torch.cuda.synchronize()
k_ms = ev[0].elapsed_time(ev[1]) / 5
h0 = time.perf_counter(); rt, rf = host_interpolate(t, f, cad); host = time.perf_counter() - h0
same = bool(np.array_equal(td.cpu().numpy(), rt) and np.array_equal(fd.cpu().numpy(), rf))
print(json.dumps({"workload": f"{n} of {n_full} cadences present, cadence numbers given",
                  "device_end_to_end_ms": best * 1e3, "device_kernels_ms": k_ms,
                  "algorithmic_GBps": (24.0 * n + 16.0 * len(rt)) / (k_ms * 1e-3) / 1e9,
                  "host_numpy_ms": host * 1e3, "bit_identical": same}))
