"""CPU: the C-ABI library loads and exports every declared symbol; host-side logic of the
GaussianProcess class; loud failure without a GPU; gloo sharding."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import gadfly_amd
from gadfly_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.load()                      # binds every name in _lib.SIGNATURES
    header = open(os.path.join(ROOT, "include", "gadfly_hip.h")).read()
    declared = set(re.findall(r"\b(gf_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gf_version() >= 100
    # argument-free host queries only (no compute without a GPU)
    assert lib.gf_leading_dim(60) == 64 and lib.gf_leading_dim(40) == 48
    assert lib.gf_leading_dim(0) < 0 and lib.gf_leading_dim(257) < 0
    assert lib.gf_state_size(60) == 64 * 64 and lib.gf_state_cols(80) == 128
    assert lib.gf_scaled_supported(64) == 1 and lib.gf_scaled_supported(65) == 0
    assert lib.gf_reduce_work(1_000_000) > 0


def test_library_has_gfx950_code_object(tmp_path):
    # (llvm-objdump --offloading drops the unbundled code objects into its working directory: a scratch one)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.SO_PATH],
                         capture_output=True, text=True, cwd=tmp_path)
    assert "gfx950" in out.stdout + out.stderr


def test_bad_arguments_are_status_codes_not_crashes():
    lib = _lib.load()
    st = lib.gf_factor(1, 10, 0, 300, 304, None, None, None, None, None, 0, None, None, None,
                       None, None, None, None)
    assert st < 0 and b"width" in lib.gf_last_error()
    st = lib.gf_solve(7, 1, 10, 4, 16, 1, None, None, None, None, None, None, None)
    assert st < 0 and b"mode" in lib.gf_last_error()
    st = lib.gf_build_scaled(1, 16, 3, 0, 2, 16, *([None] * 8), 8, None, 0, None, 0,
                             *([None] * 4), None)
    assert st < 0 and b"n_first" in lib.gf_last_error()
    st = lib.gf_loglike_fused(1, 16, 0, 1, 32, 8, 1, 0, *([None] * 8), None, 0, None, 0, None, 0,
                              *([None] * 5), None)
    assert st < 0 and b"width" in lib.gf_last_error()        # W = 65 with a real term: no fused sweep
    assert lib.gf_fused_supported(0, 30) == 1 and lib.gf_fused_supported(0, 40) == 1
    assert lib.gf_fused_supported(0, 88) == 1 and lib.gf_fused_supported(0, 89) == 0
    assert lib.gf_fused_supported(2, 31) == 0 and lib.gf_fused_supported(3, 30) == 1
    assert lib.gf_fused_state_size(0, 30) == 64 * 64
    assert lib.gf_fused_state_size(0, 40) == 96 * 80 and lib.gf_fused_state_size(0, 86) == 192 * 176
    # the product ABI has no process-wide switches and ships no experimental sweeps
    for gone in ("gf_set_generator_period", "gf_set_pipelined", "gf_loglike_blocked"):
        assert not hasattr(lib, gone)


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="CPU-only behaviour")
def test_no_gpu_fails_loudly():
    from gadfly_amd.synth import solar_like_hyperparameters
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(6), texp=60.0)
    with pytest.raises(_lib.GadflyHipError, match="no CPU fallback|no HIP device"):
        gadfly_amd.GaussianProcess(k, t=np.arange(10.0))
    with pytest.raises(_lib.GadflyHipError):
        gadfly_amd.log_likelihood_batch([k], np.arange(10.0), np.zeros(10))


def test_host_side_errors_need_no_gpu():
    from gadfly_amd.synth import solar_like_hyperparameters
    k = gadfly_amd.StellarOscillatorKernel(solar_like_hyperparameters(6), texp=60.0)
    gp = gadfly_amd.GaussianProcess(k)               # no t: nothing computed yet
    with pytest.raises(RuntimeError):
        gp.log_likelihood(np.zeros(4))
    with pytest.raises(RuntimeError):
        gp.sample()
    with pytest.raises(RuntimeError):
        gp.recompute()
    with pytest.raises(RuntimeError):
        gp.mean_value
    with pytest.raises(ValueError, match="sorted"):
        gp.compute(np.array([0.0, 2.0, 1.0]))
    with pytest.raises(ValueError, match="only one"):
        gp.compute(np.arange(3.0), yerr=1.0, diag=np.ones(3))
    with pytest.raises(ValueError, match="dimension"):
        gp.compute(np.zeros((3, 2)))
    # ndarrays pass through the unit shims untouched (reference gp.py:82-84, :111-113)
    x = np.arange(5.0)
    assert gadfly_amd.GaussianProcess._time_to_freq(x) is x
    assert gp._flux_to_ppm(x) is x
    assert callable(gp.mean) and np.all(gp.mean(x) == 0.0)
    gp.mean = lambda t: 2.0 * t
    assert np.all(gp.mean(x) == 2.0 * x)


def test_shard_bounds_partition():
    from gadfly_amd.dist import shard_bounds
    for B in (1, 7, 8, 256, 513):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(B, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
import gadfly_amd
from gadfly_amd.dist import sharded_log_likelihood
from gadfly_amd.synth import solar_like_hyperparameters, jitter_hyperparameters
from oracle import cref

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
N, J, B = 600, 6, 5
base = solar_like_hyperparameters(J)
kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 1000 + i), texp=60.0)
           for i in range(B)]
rng = np.random.default_rng(0)
t = np.arange(N) * 60e-6
y = rng.normal(size=N) * 50.0
calls = []

def checker(ks, tt, yy, yerr=None, diag=None, mean=0.0):
    # stand-in evaluator for the CPU test: the oracle (tests may use it), never the product path
    calls.append(len(ks))
    out = []
    for k in ks:
        co = k.get_device_coefficients()
        out.append(cref.loglike(co[:6], tt, np.full(len(tt), yerr ** 2) + co[6], yy - mean)[0])
    return np.array(out)

res = sharded_log_likelihood(kernels, t, y, yerr=30.0, evaluate=checker)
ref = checker(kernels, t, y, yerr=30.0)
assert res.shape == (B,) and np.array_equal(res, ref), (res, ref)
assert calls[0] == (3 if dist.get_rank() == 0 else 2)
# per-problem t, y are sliced with the kernels
ts = np.stack([t * (1 + 0.01 * i) for i in range(B)]); ys = np.stack([y + i for i in range(B)])
def checker2(ks, tt, yy, yerr=None, diag=None, mean=0.0):
    assert tt.shape[0] == len(ks) and yy.shape[0] == len(ks)
    return np.array([cref.loglike(k.get_device_coefficients()[:6], tt[i],
                     np.full(N, yerr ** 2) + k.get_device_coefficients()[6], yy[i])[0]
                     for i, k in enumerate(ks)])
res2 = sharded_log_likelihood(kernels, ts, ys, yerr=30.0, evaluate=checker2)
assert np.all(np.isfinite(res2)) and res2.shape == (B,)
# ragged batches: lists of series of different lengths are sliced like the kernels (never stacked)
lens = [300, 420, 350, 600, 512]
tr = [np.arange(n) * 60e-6 * (1 + 0.02 * i) for i, n in enumerate(lens)]
yr = [rng.normal(size=n) * 40.0 for n in lens]
er = [np.full(n, 20.0 + i) for i, n in enumerate(lens)]
seen = []
def checker3(ks, tt, yy, yerr=None, diag=None, mean=0.0):
    assert isinstance(tt, list) and isinstance(yy, list) and isinstance(yerr, list)
    assert len(tt) == len(yy) == len(yerr) == len(ks)
    seen.append([len(x) for x in tt])
    return np.array([cref.loglike(k.get_device_coefficients()[:6], tt[i],
                     yerr[i] ** 2 + k.get_device_coefficients()[6], yy[i] - mean)[0] for i, k in enumerate(ks)])
res3 = sharded_log_likelihood(kernels, tr, yr, yerr=er, evaluate=checker3)
lo, hi = (0, 3) if dist.get_rank() == 0 else (3, 5)
assert seen[0] == lens[lo:hi]
ref3 = checker3(kernels, tr, yr, yerr=er)
assert res3.shape == (B,) and np.array_equal(res3, ref3)
dist.destroy_process_group()
print("rank", os.environ["RANK"], "ok")
"""


def test_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
