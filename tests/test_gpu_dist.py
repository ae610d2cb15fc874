"""GPU: the multi-rank code path of bench.py (RCCL through torch.distributed: init with device_id,
barrier, MAX-reduce of the elapsed time) and of gadfly_amd.dist (all_gather of the results), forced
at world size 1 on the one GPU of the test box.  The partition logic for world size > 1 is covered
on CPU with gloo (tests/test_abi_and_host.py); the 8-GPU run itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update(GADFLY_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_bench_distributed_path_rccl(hip):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "1", "--warmup", "1", "--rows", "16384", "--evals", "64",
           "--no-cpu-baseline", "--no-configs"]
    out = subprocess.run(cmd, cwd=ROOT, env=_env(29541), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["roofline"]["frac"] > 0


def test_sharded_log_likelihood_rccl(hip):
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GADFLY_ROOT"])
import gadfly_amd
from gadfly_amd.dist import sharded_log_likelihood, gather_results
from gadfly_amd.synth import solar_like_hyperparameters, uniform_times, jitter_hyperparameters
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
base = solar_like_hyperparameters(6)
ks = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(base, 10 + i), texp=60.0) for i in range(5)]
t = uniform_times(4096, 60.0)
y = np.random.default_rng(0).normal(size=4096) * 50
ll = sharded_log_likelihood(ks, t, y, yerr=30.0)
# the gather helper itself over RCCL (sharded_log_likelihood returns before it at world size 1)
full = gather_results(ll, len(ks), device="cuda:0")
assert full.shape == (5,) and np.array_equal(full, ll)
buf = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(buf, op=dist.ReduceOp.MAX)
out = [torch.empty_like(buf)]
dist.all_gather(out, buf)
dist.barrier()
assert ll.shape == (5,) and np.all(np.isfinite(ll)) and float(out[0]) == 1.5
dist.destroy_process_group()
print("RCCL_OK")
'''
    env = _env(29542)
    env["GADFLY_ROOT"] = ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_bench_self_launch_two_ranks(hip):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two rank processes
    itself (the parent never touches the GPU).  Both ranks share the box's one GPU here, so the
    process group is gloo; the launch, rendezvous, barrier and MAX-reduce are the real code path."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GADFLY_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--rows", "16384", "--evals", "64", "--no-cpu-baseline", "--no-configs"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "weak"


def test_bench_strong_scaling_legs_two_ranks(hip):
    """`bench.py --gpus 2`: after the weak-scaling headline every rank takes its block of cfg3's 256 light
    curves and of cfg4's 512 walkers (static partition), evaluates it, and the results are all_gathered --
    the strong-scaling legs the driver's multi-GPU run records.  Two ranks share the box's one GPU here
    (gloo, short series); entry 0 of each batch is checked against the oracle by rank 0."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GADFLY_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--rows", "16384", "--evals", "64", "--strong-rows-scale", "0.05"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    for name, B in (("cfg3_strong", 256), ("cfg4_strong", 512)):
        leg = res["configs"][name]
        assert leg["scaling"] == "strong" and leg["n_gpus"] == 2 and leg["all_finite"]
        assert leg["value"] > 0 and abs(leg["value"] * leg["ms"] * 1e-3 - B) < 1e-6 * B
        assert leg["parity"]["ok"] and leg["parity"]["rel_err"] <= 1e-8
