#!/usr/bin/env python
"""
bench.py -- GP log-likelihood evaluations/s on synthetic solar-like light curves.

Metric (BASELINE.json): "GP log-likelihood evals/sec (N=1e6, J=30)".  One evaluation =
matrix build + semiseparable factor (gadfly ``compute``, /root/reference/gadfly/gp.py:202) +
forward solve + reductions (``log_likelihood``, gp.py:350) for FRESH hyperparameters
(SURVEY.md 8d).  Inputs (t, y, yerr) are resident in HBM before the timed region; the celerite
coefficient algebra of every proposal and its upload are INSIDE it.

A step = every rank evaluates ``--evals`` independent log-likelihoods (MCMC-walker style:
shared t, y; jittered hyperparameters, seed 1000 + id) of the N=1e6, J=30 (W=60) problem.
Ranks are independent (no data-path collective): weak scaling; value = total evals/s.

Launch:  python bench.py [--gpus N --steps K --warmup W]        (N > 1: spawns its N ranks itself)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0: the headline metric with ``roofline`` and ``cpu_baseline``; at
N = 1 also ``exact_rows`` (the same workload with exact generator rows), ``parity`` (a gate over
several timed evaluations) and ``configs`` (BASELINE.json's other configurations, each with a
sampled oracle check) -- all measured by this run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # vector FP64 peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", "--n", dest="n", type=int, default=1_000_000,
                    help="cadences per light curve")
    ap.add_argument("--terms", "--j", dest="j", type=int, default=30,
                    help="SHO terms (celerite width = 2J)")
    ap.add_argument("--evals", type=int, default=2048,
                    help="independent evaluations (walkers) per rank per step")
    ap.add_argument("--tile-rows", type=int, default=8192, help="rows per streamed tile")
    ap.add_argument("--overlap", action="store_true",
                    help="build tile k+1 on a side stream during the sweep of tile k (fall-back "
                         "path only; slower: the build displaces a sweep wave)")
    ap.add_argument("--kernel", choices=["auto", "column", "tiled"], default="auto",
                    help="fused sweep variant (C-ABI argument): tiled = k_factor7 (2 x 32 lane "
                         "tiling), column = k_factor3 (one column per lane); auto picks tiled for "
                         "kernels of complex terms only")
    ap.add_argument("--generator-period", type=int, default=0,
                    help="rows between exact re-anchorings of the in-kernel row generator "
                         "(1 = exact rows ... 64 = at the scaling-block resets only); 0 = chosen "
                         "after the warm-up from the measured conditioning so that the generator's "
                         "share of the log-likelihood error stays below 1e-9 (DESIGN.md 2.1a)")
    ap.add_argument("--gate", type=int, default=8,
                    help="timed evaluations checked against the oracle (N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the oracle legs (cpu_baseline, parity gate, config checks)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip BASELINE.json's other configurations (cfg2 predict, cfg3, cfg4, cfg5)")
    ap.add_argument("--no-exact-rows", action="store_true",
                    help="skip the extra step with exact generator rows (period 1)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the all-cores CPU baseline (0 = every core this process may use, "
                         "at most 64: an evaluation holds 1.5 GB)")
    ap.add_argument("--strong-rows-scale", type=float, default=1.0,
                    help="scale the series length of the strong-scaling legs (rehearsals on one GPU; the "
                         "record uses 1.0 = BASELINE.json's sizes)")
    ap.add_argument("--no-strong", action="store_true",
                    help="--gpus N > 1: skip the strong-scaling legs (cfg3 / cfg4 partitioned over the ranks)")
    return ap.parse_args()


# -------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` starts its N ranks itself.  The parent makes NO GPU call
# (it never imports torch) and never re-execs: plain child processes, rank 0's stdout is relayed.
# -------------------------------------------------------------------------------------------------
def self_launch(args):
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(codes, key=abs)


# -------------------------------------------------------------------------------------------------
# oracle legs (CPU): parity gate + cpu_baseline.  bench.py's cpu leg is one of the three places that
# may touch oracle/ (tests, smoke, here) -- as the checker and the timed baseline, never the product.
# -------------------------------------------------------------------------------------------------
def oracle_loglikes(jobs, threads):
    """jobs: list of (coeffs7, t, diag_user, y).  Runs the C restatement on `threads` host threads
    (ctypes releases the GIL; celerite2 itself is single-threaded, one problem per core is how a
    batch is run on a CPU).  Returns (values, infos, wall seconds, per-job seconds)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref
    cref.lib()

    def one(job):
        co, t, diag, y = job
        t0 = time.perf_counter()
        v, info = cref.loglike(co[:6], t, diag + co[6], y)
        return v, info, time.perf_counter() - t0

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        res = list(ex.map(one, jobs))
    wall = time.perf_counter() - t0
    return [r[0] for r in res], [r[1] for r in res], wall, [r[2] for r in res]


def host_cores():
    """(cores this process may use, how that was found): the scheduler affinity, capped by the cgroup CPU
    quota when there is one (a GPU box hands a job a share of the host's cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = f"sched_getaffinity={n}"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(float(quota) / float(period)))
            how += f", cgroup cpu.max={q}"
            n = min(n, q)
    except (OSError, ValueError):
        pass
    return n, how


def celerite2_loglike(hp, delta, t, yerr, y):
    """The reference's own CPU path (celerite2 is what gadfly.GaussianProcess calls,
    /root/reference/gadfly/gp.py:202, :350) when the package happens to be importable on this host;
    returns None otherwise (it is not installed in the build image and cannot be fetched)."""
    try:
        import celerite2
        from celerite2 import terms
    except ImportError:
        return None
    sho = [terms.SHOTerm(S0=p["hyperparameters"]["S0"], w0=p["hyperparameters"]["w0"],
                         Q=p["hyperparameters"]["Q"]) for p in hp]
    t0 = time.perf_counter()
    gp = celerite2.GaussianProcess(terms.TermConvolution(terms.TermSum(*sho), delta), mean=0.0)
    gp.compute(t, yerr=yerr)
    v = gp.log_likelihood(y)
    return float(v), time.perf_counter() - t0


def check_sample(sample):
    """Oracle check of ONE entry of a configuration measured by tools/configs.py."""
    from oracle import cref
    co = sample["coeffs"]
    t, diag = sample["t"], sample["diag"] + co[6]
    if sample["kind"] == "loglike":
        ref, info = cref.loglike(co[:6], t, diag, sample["y"])
        rel = abs(sample["got"] - ref) / abs(ref)
        return {"what": f"log-likelihood of entry {sample['index']} vs oracle", "rel_err": rel,
                "gate": 1e-8, "ok": bool(info == 0 and rel <= 1e-8)}
    c, a, U, V = cref.get_matrices(co[:6], t, diag)
    d, Wm, info = cref.factor(t, c, a, U, V)
    if sample["kind"] == "dot_tril":
        ref = cref.matmul_lower(t, c, U, Wm, sample["n"] * np.sqrt(d))
        err = float(np.max(np.abs(sample["got"] - ref)) / np.max(np.abs(ref)))
        return {"what": f"draw column {sample['column']} vs oracle matmul_lower", "rel_err": err,
                "gate": 1e-6, "ok": bool(info == 0 and err <= 1e-6)}
    # predict: alpha = K^-1 y; mean at the observed times y - diag_user alpha (sampled), and at t*
    y = sample["y"]
    z = cref.solve_lower(t, c, U, Wm, y)
    ll_ref = -0.5 * (np.sum(np.log(d)) + len(t) * np.log(2 * np.pi)) - 0.5 * np.sum(z * z / d)
    alpha = cref.solve_upper(t, c, U, Wm, z / d)
    idx = sample["idx"]
    mu_ref = y[idx] - sample["diag"][idx] * alpha[idx]
    _, _, Us, Vs = cref.get_matrices(co[:6], sample["ts"], 0.0)
    mus_ref = cref.general_matmul(sample["ts"], t, c, Us, Vs, U, V, alpha)
    e_ll = abs(sample["got_ll"] - ll_ref) / abs(ll_ref)
    e_mu = float(np.max(np.abs(sample["got_mu"] - mu_ref)) / np.max(np.abs(mu_ref)))
    e_ms = float(np.max(np.abs(sample["got_mus"] - mus_ref)) / np.max(np.abs(mus_ref)))
    return {"what": "log_likelihood, predict(y) at 64 sampled rows, predict(y, t*) at all 1000 times vs oracle",
            "loglike_rel_err": e_ll, "predict_mean_rel_err": e_mu, "predict_new_times_rel_err": e_ms,
            "gate": "1e-8 / 1e-6 / 1e-6",
            "ok": bool(info == 0 and e_ll <= 1e-8 and e_mu <= 1e-6 and e_ms <= 1e-6)}


def runtime_speed_cpu(r):
    """CPU side of the runtime-speed leg: the same `gp.sample()` on ONE host core through the oracle's C restatement
    (randn + sqrt(d) scaling + matmul_lower + the mean subtraction, on a factor made beforehand -- the notebook
    times the call, not the construction), and the GPU's draw checked against it (1e-6)."""
    from oracle import cref
    cpu = {}
    for smp in r.pop("_samples"):
        n, t, co = smp["N"], smp["t"], smp["coeffs"]
        c, a, U, V = cref.get_matrices(co[:6], t, np.zeros(n) + co[6])
        d, Wm, info = cref.factor(t, c, a, U, V)
        best, want = None, None
        for _ in range(2):
            np.random.seed(42)
            t0 = time.perf_counter()
            nn = np.random.randn(n)
            want = cref.matmul_lower(t, c, U, Wm, (nn * np.sqrt(d))[:, None])[:, 0]
            want = want - want.mean()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        err = float(np.max(np.abs(smp["draw"] - want)) / np.max(np.abs(want)))
        cpu[n] = (best, err, info)
    worst = 0.0
    for row in r["durations"]:
        if row["N"] in cpu:
            row["cpu_sample_ms"] = 1e3 * cpu[row["N"]][0]
            row["draw_rel_err_vs_oracle"] = cpu[row["N"]][1]
            worst = max(worst, cpu[row["N"]][1])
        else:
            row["cpu_sample_ms"] = None
    nmax = max(cpu)
    r["cpu_port_us_per_row"] = 1e6 * cpu[nmax][0] / nmax
    r["cpu_port_years_per_second"] = (nmax / cpu[nmax][0]) / 1440.0 / 365.25
    r["cpu_note"] = ("oracle/celerite_ref.c on one core, sizes up to 1.1e5 rows (the port is linear in N beyond a few "
                     "thousand rows: years per second from its time per row at the largest size run)")
    r["parity"] = {"what": "gp.sample() (np.random.seed(42)) vs oracle matmul_lower at every size checked",
                   "rel_err": worst, "gate": 1e-6, "ok": bool(worst <= 1e-6 and not any(v[2] for v in cpu.values()))}
    return r


def other_configs(check):
    """BASELINE.json configs 2 (API legs), 3, 4, 5 on this GPU: measured by tools/configs.py, one
    sampled entry of each checked against the oracle here (a failed check fails the run)."""
    from tools import configs
    import torch
    out = {}
    for name, fn in (("cfg2_api", configs.measure_cfg2_api),
                     ("cfg2_api_jd", lambda: configs.measure_cfg2_api(jd=True)),
                     ("jd_batch", configs.measure_jd_batch),
                     ("runtime_speed", configs.measure_runtime_speed),
                     ("cfg3", configs.measure_cfg3),
                     ("cfg3_shard", lambda: configs.measure_cfg3_shard(out["cfg3"]["ms"])),
                     ("cfg4", configs.measure_cfg4),
                     ("cfg4_shard", lambda: configs.measure_cfg4_shard(out["cfg4"]["ms"])),
                     ("cfg5", configs.measure_cfg5)):
        r = fn()
        if name == "runtime_speed":
            if check:
                r = runtime_speed_cpu(r)
            r.pop("_samples", None)
        else:
            sample = r.pop("_sample")
            if check:
                r["parity"] = check_sample(sample)
        if check and not r["parity"]["ok"]:
            raise SystemExit(f"parity gate failed in {name}: {r['parity']}")
        out[name] = r
        torch.cuda.empty_cache()
    if "cfg2_api_jd" in out:
        a, b = out["cfg2_api"], out["cfg2_api_jd"]
        b["vs_cfg2_api"] = {k: b[k] / a[k] for k in ("compute_ms", "log_likelihood_ms", "predict_mean_ms",
                                                    "predict_1000_new_times_ms")}
    return out


def strong_configs(dist, device, rank, world, backend, check, rows_scale=1.0):
    """--gpus N > 1: BASELINE.json's sharded configurations (cfg3: 256 light curves, cfg4: 512 walkers) with
    the WHOLE batch partitioned over the N ranks -- strong scaling, next to the weak-scaling headline."""
    from tools import configs
    import torch
    out = {}
    for which in ("cfg3", "cfg4"):
        r = configs.measure_strong(which, dist, device, rank, world, backend, rows_scale)
        sample = r.pop("_sample")
        if rank == 0 and check:
            r["parity"] = check_sample(sample)
            if not r["parity"]["ok"]:
                raise SystemExit(f"parity gate failed in strong-scaling {which}: {r['parity']}")
        out[which + "_strong"] = r
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import gadfly_amd
    from gadfly_amd import _lib
    from gadfly_amd.synth import (solar_like_hyperparameters, uniform_times,
                                  jitter_hyperparameters)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # one rank per GPU; (local % device_count only matters when rehearsing N ranks on fewer GPUs)
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    dist = None
    # "nccl" is RCCL on ROCm; GADFLY_BENCH_BACKEND=gloo is for rehearsals on a single GPU
    backend = os.environ.get("GADFLY_BENCH_BACKEND", "nccl")
    # GADFLY_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, MAX-reduce of
    # the time) even at world size 1 -- how the N > 1 path is smoke-tested on a one-GPU box
    if world > 1 or os.environ.get("GADFLY_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))            # (a forced one-rank group started without a launcher)
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    N, J, E = args.n, args.j, args.evals
    W = 2 * J
    cadence = 60.0
    base = solar_like_hyperparameters(J)
    t = uniform_times(N, cadence)
    rng = np.random.Generator(np.random.PCG64(12345))
    # data: red noise + white noise at the yerr level.  (SURVEY.md 8d asks for a prior draw; that
    # needs a factorisation first, and the cost of an evaluation does not depend on the values.)
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    yerr = 30.0

    def walker_ids(step):
        return [(rank * 1000003 + step * E + e) for e in range(E)]

    def walker_kernel(step, e):
        return gadfly_amd.StellarOscillatorKernel(
            jitter_hyperparameters(base, 1000 + walker_ids(step)[e]), texp=cadence)

    nsteps = args.warmup + args.steps
    first = [walker_kernel(0, e) for e in range(E)]
    ev = gadfly_amd.BatchedLogLikelihood(first, t, y, yerr=yerr, device=device,
                                         tile_rows=args.tile_rows, overlap_build=args.overlap)
    # what a sampler hands over per step: (E, J) arrays of proposed hyperparameters.  The celerite
    # coefficient algebra of the E kernels and its upload are part of every
    # evaluation and run INSIDE the timed region (vectorised: gadfly_amd.batch.sho_coefficient_pack)
    delta = first[0].delta                           # the kernels' exposure in 1/uHz

    def proposals(step):
        hps = [jitter_hyperparameters(base, 1000 + i) for i in walker_ids(step)]
        return tuple(np.array([[p["hyperparameters"][k] for p in hp] for hp in hps])
                     for k in ("S0", "w0", "Q"))

    extra = 0 if (args.no_exact_rows or world > 1) else 1      # one more step with exact rows
    params = [proposals(s) for s in range(nsteps + extra)]
    eng = ev.engine
    ref_pack, vec_pack = ev.pack(first), ev.pack_parameters(*params[0], delta)
    for a, b in zip(ref_pack[:5], vec_pack[:5]):        # same numbers as the per-object path
        a = a if isinstance(a, (tuple, list)) else (a,)
        b = b if isinstance(b, (tuple, list)) else (b,)
        if not all(bool(torch.equal(x, y_)) for x, y_ in zip(a, b)):
            raise SystemExit("vectorised coefficient pack differs from the per-kernel path")
    del first
    eng.sweep_variant = {"auto": _lib.GF_SWEEP_AUTO, "column": _lib.GF_SWEEP_COLUMN,
                         "tiled": _lib.GF_SWEEP_TILED}[args.kernel]
    eng.time_factor = True
    eng.force_streaming = True      # the metric is the streamed sweep of independent evaluations
    if args.generator_period > 0:
        eng.generator_period = args.generator_period
    torch.cuda.synchronize()

    outs, mind = [], []

    def step(s):
        outs.append(ev.evaluate_device(ev.pack_parameters(*params[s], delta)))
        mind.append(eng.acc[:, 2].clone())              # min pivot per walker (condition estimate)

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    gen_cond = None
    if args.generator_period == 0 and args.warmup > 0:
        gen_cond, _ = ev.calibrate()        # what a sampler does after its first steps
    eng.factor_events = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    guard_reruns = ev.resolve()     # accuracy guard of the asynchronous evaluations (inside the timing)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    timed_events = list(eng.factor_events)
    timed_period = int(eng.generator_period)

    # the same workload with exact generator rows (period 1), one step, for the record
    exact = None
    if extra:
        eng.generator_period = 1
        eng.factor_events = []
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        step(nsteps)
        torch.cuda.synchronize()
        edt = time.perf_counter() - e0
        ems = [a.elapsed_time(b) for a, b, _ in eng.factor_events]
        exact = {"value": E / edt, "unit": "evals/s", "generator_period": 1, "steps": 1,
                 "ms_per_step": 1e3 * edt, "kernel_ms": float(np.mean(ems)) if ems else None}
        eng.generator_period = timed_period

    lls = torch.stack(outs).cpu().numpy()          # (nsteps [+1], E)
    dmin = torch.stack(mind).cpu().numpy()
    if not np.all(np.isfinite(lls)):
        raise SystemExit("non-finite log-likelihood in the benchmark")
    # dominant kernel: the factor(+solve) sweep, HIP events on its own stream
    fac_ms = [a.elapsed_time(b) for a, b, _ in timed_events]
    fac_rows = [r for _, _, r in timed_events]
    fac_avg_ms = float(np.mean(fac_ms)) if fac_ms else float("nan")
    fac_avg_rows = float(np.mean(fac_rows)) if fac_rows else float("nan")

    total_evals = world * E * args.steps
    value = total_evals / elapsed
    result = {
        "metric": "GP log-likelihood evals/sec (N=1e6, J=30)" if (N, J) == (1_000_000, 30)
                  else f"GP log-likelihood evals/sec (N={N}, J={J})",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"cfg2: single solar-like light curve N={N}, J={J} SHO terms (celerite "
                        f"width {W}), 60 s cadence, yerr=30 ppm, y = red + white noise (not a prior "
                        f"draw: cost-neutral); {E} independent evaluation(s) "
                        "(MCMC-walker style, fresh hyperparameters each) per rank per step, "
                        "each = coefficient algebra + build + factor + solve + reduce; time axis "
                        f"streamed in tiles of {args.tile_rows} rows",
            "N": N, "J": J, "W": W, "evals_per_rank_per_step": E,
            "tile_rows": args.tile_rows, "generator_period": timed_period,
            "condition_estimate": gen_cond, "accuracy_guard_reruns": guard_reruns,
            "parallelism": f"independent evaluations x{world}" if world > 1 else "1 GPU",
        },
    }

    if rank == 0:
        # 8 (3W + 4) algorithmic bytes per row and evaluation; one launch
        # advances all E evaluations by one tile
        alg_bytes = 8.0 * fac_avg_rows * (3 * W + 4) * E
        alg_flops = 5.0 * W * W * fac_avg_rows * E
        ach = alg_bytes / (fac_avg_ms * 1e-3) / 1e9
        fused = eng._fused_ok()
        tiled = fused and eng.sweep_variant != _lib.GF_SWEEP_COLUMN and eng.Jr == 0 and eng.Jc <= 31
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE per the
        # guide's gfx950 correction; profiles/r03_traffic.json, else earlier rounds), scaled to this launch
        # shape: a constant measured on this kernel, not a counter of this run
        traffic, traffic_src = None, None
        for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", name)))
                if tr.get("W") == W and fused:
                    traffic = tr["hbm_bytes_per_row_eval"] * fac_avg_rows * E
                    traffic_src = f"profiles/{name} (PMC passes of this kernel, scaled)"
                    break
            except (OSError, ValueError, KeyError):
                pass
        valu_tf = alg_flops / (fac_avg_ms * 1e-3) / 1e12
        result["roofline"] = {
            # what binds this kernel: the fused sweep moves ~1.6 % of the algorithmic bytes
            # through HBM; it is FP64-vector-issue- and dependent-chain-bound.  achieved = ALGORITHMIC flops
            # (5 W^2 per row and evaluation) per launch / the launch's HIP-event time
            "bound": "fp64_valu",
            "kernel": ("k_factor7 (fused build + factor + forward solve, 2 x 32 lane tiling)" if tiled else
                       "k_factor3 (fused build + factor + forward solve, one column per lane)" if fused
                       else "materialised rows: k_build2 + k_factor2 / k_factor2w / k_factor"),
            "achieved": valu_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": valu_tf / FP64_VALU_PEAK_TF,
            "algorithmic_flops_per_launch": alg_flops,
            # the contract's HBM-algorithmic figure: algorithmic bytes 8 (3 W + 4) per row and evaluation
            # per launch / the same time, against the HBM peak
            "hbm_algorithmic": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": alg_bytes},
            "traffic": traffic, "traffic_source": traffic_src,
            "hbm_measured_GBs": (traffic / (fac_avg_ms * 1e-3) / 1e9) if traffic else None,
            "kernel_ms": fac_avg_ms, "launches_timed": len(fac_ms),
            "rows_per_launch": fac_avg_rows,
        }
        if exact is not None:
            result["exact_rows"] = exact
        oracle_ok = not args.no_cpu_baseline and world == 1
        if oracle_ok:
            usable, how = host_cores()
            threads = args.cpu_threads or min(usable, 64)
            diag = np.full(N, yerr ** 2)
            # ---- cpu_baseline: ONE evaluation on ONE core (celerite2 is single-threaded) ----------
            k0 = walker_kernel(args.warmup, 0)              # first timed evaluation
            ref1, info1, wall1, _ = oracle_loglikes([(k0.get_device_coefficients(), t, diag, y)], 1)
            result["cpu_baseline"] = {
                "value": 1.0 / wall1, "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"1 evaluation of the first timed walker on all {N} rows ({wall1:.2f} s; "
                          "oracle/celerite_ref.c, gcc -O3 -march=x86-64-v3 restatement of the "
                          "celerite2 algorithm, single thread like celerite2)",
                "host_cpu_count": os.cpu_count(), "usable_cores": usable, "usable_cores_from": how,
            }
            # the reference's own CPU path, should celerite2 be importable on this host
            c2 = celerite2_loglike(jitter_hyperparameters(base, 1000 + walker_ids(args.warmup)[0]),
                                   k0.delta, t, yerr, y)
            if c2 is not None:
                v2, wall2 = c2
                result["cpu_baseline"].update(
                    value=1.0 / wall2, kind="celerite2",
                    sample=f"1 evaluation of the first timed walker through celerite2.GaussianProcess "
                           f"(compute + log_likelihood, {wall2:.2f} s, one core)",
                    port={"value": 1.0 / wall1, "unit": "evals/s", "cores": 1},
                    celerite2_vs_gpu_rel_err=abs(float(lls[args.warmup, 0]) - v2) / abs(v2))
            # ---- parity gate: >= 8 timed evaluations spread over the steps, including the one with
            # the largest device condition estimate max(a)/min(d); run on all host cores, which is
            # also the all-cores CPU baseline (one problem per thread) ------------------------------
            timed = range(args.warmup, nsteps + extra)
            worst = np.unravel_index(np.argmin(dmin[args.warmup:]), dmin[args.warmup:].shape)
            picks = [(args.warmup + int(worst[0]), int(worst[1]))]
            # (at least one evaluation per host thread, so that the gate doubles as the all-cores baseline)
            ng = max(args.gate, threads, 1)
            for i in range(ng - 1):
                s = list(timed)[i % len(timed)]
                e = (i * 977 + 1) % E
                if (s, e) not in picks:
                    picks.append((s, e))
            jobs = [(walker_kernel(s, e).get_device_coefficients(), t, diag, y) for s, e in picks]
            refs, infos, wall, each = oracle_loglikes(jobs, threads)
            rels = [abs(float(lls[s, e]) - r) / abs(r) for (s, e), r in zip(picks, refs)]
            result["parity"] = {
                "gate": 1e-8, "checked": len(picks), "loglike_rel_err_max": max(rels),
                "loglike_rel_err_first_timed": abs(float(lls[args.warmup, 0]) - ref1[0]) / abs(ref1[0]),
                "worst_conditioned": {"step": picks[0][0], "walker": picks[0][1],
                                      "min_pivot": float(dmin[picks[0]]), "rel_err": rels[0]},
                "includes_exact_rows_step": bool(extra),
            }
            result["cpu_baseline"]["all_cores"] = {
                "value": len(jobs) / wall, "unit": "evals/s", "cores": min(threads, len(jobs)),
                "kind": "port",
                "sample": f"{len(jobs)} evaluations, one per thread on every core this process may use "
                          f"({how}; at most 64: an evaluation holds 1.5 GB) -- {wall:.2f} s wall, "
                          f"{np.mean(each):.2f} s each",
            }
            if any(infos) or info1[0] or max(rels) > 1e-8 or \
                    result["parity"]["loglike_rel_err_first_timed"] > 1e-8:
                raise SystemExit(f"parity gate failed: {result['parity']}")
        if world == 1 and not args.no_configs:
            del ev, eng, outs, mind
            torch.cuda.empty_cache()
            result["configs"] = other_configs(check=oracle_ok)
            api = result["configs"]["cfg2_api"]
            # ONE MCMC chain (one proposal at a time) through the drop-in class: fresh hyperparameters,
            # factorisation, log-likelihood -- the latency path next to the batched headline
            result["single_chain"] = {
                "value": 1e3 / api["recompute_plus_log_likelihood_ms"], "unit": "evals/s",
                "ms_per_eval": api["recompute_plus_log_likelihood_ms"],
                "what": "GaussianProcess.recompute() + log_likelihood(y) at N=1e6, J=30: one evaluation at a "
                        "time (exact time-parallel factorisation), host arrays in and out",
                # the same chain through the batched evaluator with B = 1 (log-likelihoods only: two sweeps, no
                # stored factor): fresh hyperparameters, coefficient algebra, upload, evaluation, readback per step
                "evaluator_b1": {"value": 1e3 / api["evaluator_b1_ms"], "unit": "evals/s",
                                 "ms_per_eval": api["evaluator_b1_ms"]},
            }
    # ---- strong scaling of the sharded configurations (every rank takes part: collectives) ----------
    if dist is not None and world > 1 and not args.no_strong and not args.no_configs:
        try:
            del ev, eng, outs, mind
        except NameError:
            pass
        torch.cuda.empty_cache()
        strong = strong_configs(dist, device, rank, world, backend, check=not args.no_cpu_baseline,
                                rows_scale=args.strong_rows_scale)
        if rank == 0:
            result["configs"] = strong
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
