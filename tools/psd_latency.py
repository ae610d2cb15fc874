#!/usr/bin/env python
"""Power spectra of prior draws at cfg5's size (64 draws of N = 5e5, J = 30): sample -> FFT power ->
log bins, draws left on the GPU, next to the numpy/scipy formulation of the reference
(gadfly/psd.py:186-300, :566-587) on the host.  Usage: python tools/psd_latency.py [N] [R] [bins]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gadfly_amd  # noqa: E402
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 25
k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(30), texp=60.0)
t = uniform_times(N, 60.0)
gp = gadfly_amd.GaussianProcess(k, t=t, yerr=30.0)
np.random.seed(1)


def timed(fn, reps=3):
    best, out = 1e30, None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


ms_draw, draws = timed(lambda: gp.sample_device(size=R), reps=2)
ms_psd, ps = timed(lambda: gadfly_amd.PowerSpectrum.from_flux(draws, 60e-6))
ms_bin, binned = timed(lambda: ps.bin(NB))

# kernel-only durations with HIP events on the current stream
lib, p = gadfly_amd._lib.load(), gadfly_amd._lib.ptr
spec = torch.view_as_real(torch.fft.rfft(draws, dim=-1)).contiguous()
M = spec.shape[1]
power = torch.empty((R, M - 1), dtype=torch.float64, device=draws.device)
st = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
k_ms = {}
for name, call in (
        ("rfft (hipFFT)", lambda: torch.fft.rfft(draws, dim=-1)),
        ("k_psd_power", lambda: lib.gf_psd_power(R, M, 1, ps.norm, p(spec), p(power), st))):
    call(); torch.cuda.synchronize()
    ev[0].record(); [call() for _ in range(10)]; ev[1].record(); torch.cuda.synchronize()
    k_ms[name] = ev[0].elapsed_time(ev[1]) / 10
alg = 24.0 * R * (M - 1)            # read one complex, write one double per frequency and series

# host formulation of the reference for ONE series (numpy FFT + scipy.binned_statistic with per-bin
# callables that look the bin's end points up by value, gadfly/psd.py:186-227, :566-587), timed here
# as the CPU side of the comparison
from scipy.stats import binned_statistic  # noqa: E402

_trapz = getattr(np, "trapezoid", None) or np.trapz


def host_fft_power(flux, d):
    n = len(flux)
    spec = np.fft.rfft(flux)
    return np.fft.rfftfreq(n, d)[1:], (np.real(spec * np.conj(spec)) * (d / (2 * np.pi) ** 0.5 / n))[1:]


def host_binned(freq, power, bins):
    axis = np.log10(freq)

    def stat(yv):
        lo = np.argwhere(power == yv[0])[0, 0]
        hi = np.argwhere(power == yv[-1])[0, 0]
        if hi > lo and axis[hi] - axis[lo] > 0:
            return _trapz(yv, axis[lo:hi + 1]) / (axis[hi] - axis[lo])
        return yv[0]

    def stat_err(yv):
        lo = np.argwhere(power == yv[0])[0, 0]
        hi = np.argwhere(power == yv[-1])[0, 0]
        if hi > lo and axis[hi] - axis[lo] > 0:
            return np.nanstd(yv) / len(yv) ** 0.5 * np.nanmean(axis[lo:hi + 1]) / (axis[hi] - axis[lo])
        return yv[0]

    out = binned_statistic(axis, power, statistic=stat, bins=bins).statistic
    binned_statistic(axis, power, statistic=stat_err, bins=bins)       # the reference's second pass
    return out


one = draws[0].cpu().numpy()
t0 = time.perf_counter(); f, pw = host_fft_power(one, 60e-6); h_fft = time.perf_counter() - t0
t0 = time.perf_counter(); s = host_binned(f, pw, NB); h_bin = time.perf_counter() - t0
dev = float(np.nanmax(np.abs(binned.power[0] - s) / np.abs(s)))

print(json.dumps({
    "workload": f"{R} draws x N={N}, J=30, {NB} log bins",
    "sample_device_ms": ms_draw, "from_flux_ms": ms_psd, "bin_ms": ms_bin,
    "kernel_ms": k_ms,
    "k_psd_power_GBps": alg / (k_ms["k_psd_power"] * 1e-3) / 1e9,
    "host_numpy_ms_per_series": {"fft_power": h_fft * 1e3, "binned_statistic": h_bin * 1e3},
    "host_numpy_ms_all_series": (h_fft + h_bin) * 1e3 * R,
    "binned_power_rel_dev_vs_host": dev,
}))
