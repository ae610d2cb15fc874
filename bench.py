#!/usr/bin/env python
"""
bench.py -- GP log-likelihood evaluations/s on synthetic solar-like light curves.

Metric (BASELINE.json): "GP log-likelihood evals/sec (N=1e6, J=30)".  One evaluation =
matrix build + semiseparable factor (gadfly ``compute``, /root/reference/gadfly/gp.py:202) +
forward solve + reductions (``log_likelihood``, gp.py:350) for FRESH hyperparameters
(SURVEY.md 8d).  Inputs (t, y, yerr and the pre-packed coefficient sets) are resident in HBM
before the timed region.

A step = every rank evaluates ``--evals`` independent log-likelihoods (MCMC-walker style:
shared t, y; jittered hyperparameters, seed 1000 + id) of the N=1e6, J=30 (W=60) problem.
Ranks are independent (no data-path collective): weak scaling; value = total evals/s.

Launch:  python bench.py [--gpus 1 --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
                    help="build tile k+1 on a side stream during the sweep of tile k (slower: "
                         "the build waves displace one of the two sweep waves per SIMD)")
    ap.add_argument("--kernel", choices=["auto", "blocked", "fused", "pipelined", "split", "tiled"], default="auto",
                    help="sweep kernel: blocked = k_factor4 (FP64 MFMA, rank-16 blocks), "
                         "fused = k_factor3 (vector FMA); auto takes blocked when supported")
    ap.add_argument("--generator-period", type=int, default=0,
                    help="rows between exact re-anchorings of the in-kernel row generator "
                         "(1 = exact rows ... 64 = at the scaling-block resets only); 0 = chosen "
                         "after the warm-up from the measured conditioning so that the generator's "
                         "share of the log-likelihood error stays below 1e-9 (DESIGN.md 2.1a)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0,
                    help="rows of the CPU-baseline sample (0 = full N, one evaluation)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import gadfly_amd
    from gadfly_amd.synth import (solar_like_hyperparameters, uniform_times,
                                  jitter_hyperparameters)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
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
    # data: smooth red-noise + white noise at the yerr level (a prior draw needs a factor
    # first; the likelihood cost does not depend on the values)
    y = np.cumsum(rng.normal(size=N)) * 5.0 + 30.0 * rng.normal(size=N)
    yerr = 30.0

    def walkers(step):
        ids = [(rank * 1000003 + step * E + e) for e in range(E)]
        return [gadfly_amd.StellarOscillatorKernel(
            jitter_hyperparameters(base, 1000 + i), texp=cadence) for i in ids]

    nsteps = args.warmup + args.steps
    ev = gadfly_amd.BatchedLogLikelihood(walkers(0), t, y, yerr=yerr, device=device,
                                         tile_rows=args.tile_rows,
                                         overlap_build=args.overlap)
    # what a sampler hands over per step: (E, J) arrays of proposed hyperparameters.  The celerite
    # coefficient algebra of the E kernels and its upload are part of every
    # evaluation and run INSIDE the timed region (vectorised: gadfly_amd.batch.sho_coefficient_pack)
    delta = walkers(0)[0].delta                     # the kernels' exposure in 1/uHz

    def proposals(step):
        ids = [(rank * 1000003 + step * E + e) for e in range(E)]
        hps = [jitter_hyperparameters(base, 1000 + i) for i in ids]
        return tuple(np.array([[p["hyperparameters"][k] for p in hp] for hp in hps])
                     for k in ("S0", "w0", "Q"))

    params = [proposals(s) for s in range(nsteps)]
    eng = ev.engine
    ref_pack, vec_pack = ev.pack(walkers(0)), ev.pack_parameters(*params[0], delta)
    for a, b in zip(ref_pack[:5], vec_pack[:5]):        # same numbers as the per-object path
        a = a if isinstance(a, (tuple, list)) else (a,)
        b = b if isinstance(b, (tuple, list)) else (b,)
        if not all(bool(torch.equal(x, y)) for x, y in zip(a, b)):
            raise SystemExit("vectorised coefficient pack differs from the per-kernel path")
    if args.kernel == "fused":
        eng.lib.gf_set_pipelined(4)
    elif args.kernel == "pipelined":
        eng.lib.gf_set_pipelined(1)
    elif args.kernel == "split":
        eng.lib.gf_set_pipelined(2)
    elif args.kernel == "tiled":
        eng.lib.gf_set_pipelined(3)
    if args.kernel in ("fused", "pipelined", "split", "tiled"):
        eng.allow_blocked = False
    elif args.kernel == "blocked" and not eng._blocked_ok():
        raise SystemExit("--kernel blocked: not supported for this term structure / cadence")
    eng.time_factor = True
    eng.force_streaming = True      # the metric is the streamed sweep of independent evaluations
    if args.generator_period > 0:
        eng.generator_period = args.generator_period
    torch.cuda.synchronize()

    outs = []
    for s in range(args.warmup):
        outs.append(ev.evaluate_device(ev.pack_parameters(*params[s], delta)))
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
        outs.append(ev.evaluate_device(ev.pack_parameters(*params[s], delta)))
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

    lls = torch.stack(outs).cpu().numpy()          # (nsteps, E)
    if not np.all(np.isfinite(lls)):
        raise SystemExit("non-finite log-likelihood in the benchmark")
    # dominant kernel: the factor(+solve) sweep, HIP events on its own stream
    fac_ms = [a.elapsed_time(b) for a, b, _ in eng.factor_events]
    fac_rows = [r for _, _, r in eng.factor_events]
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
            "workload": f"single solar-like light curve N={N}, J={J} SHO terms (celerite "
                        f"width {W}), 60 s cadence, yerr=30 ppm; {E} independent evaluation(s) "
                        "(MCMC-walker style, fresh hyperparameters each) per rank per step, "
                        "each = coefficient algebra + build + factor + solve + reduce; time axis streamed in tiles of "
                        f"{args.tile_rows} rows",
            "N": N, "J": J, "W": W, "evals_per_rank_per_step": E,
            "tile_rows": args.tile_rows, "generator_period": int(eng.generator_period),
            "condition_estimate": gen_cond,
            "parallelism": f"independent evaluations x{world}" if world > 1 else "1 GPU",
        },
    }

    if rank == 0:
        # 8 (3W + 4) algorithmic bytes per row and evaluation; one launch
        # advances all E evaluations by one tile
        alg_bytes = 8.0 * fac_avg_rows * (3 * W + 4) * E
        alg_flops = 5.0 * W * W * fac_avg_rows * E
        ach = alg_bytes / (fac_avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # profiles/r01_traffic.json), scaled to this launch shape; null if not applicable
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tr.get("W") == W and getattr(eng, "_fused_ok", lambda: False)():
                traffic = tr["hbm_bytes_per_row_eval"] * fac_avg_rows * E
        except (OSError, ValueError, KeyError):
            pass
        co0 = walkers(0)[0].get_device_coefficients()    # term structure (real / complex counts)
        result["roofline"] = {
            "bound": "hbm",
            "kernel": {"blocked": "k_factor4 (fused build + factor + forward solve, rank-16 blocks "
                                  "on v_mfma_f64_16x16x4)",
                       "fused": {"pipelined": "k_factor5 (fused build + factor + forward solve, post work "
                                              "of row n-1 interleaved with sweep n)",
                                 "split": "k_factor6 (fused build + factor + forward solve, split sweep: "
                                          "the fold runs under the row's chain)",
                                 "fused": "k_factor3 (fused build + factor + forward solve, one column "
                                          "per lane)"}.get(
                                     args.kernel,
                                     "k_factor7 (fused build + factor + forward solve, 2 x 32 lane tiling)"
                                     if len(co0[0]) == 0 and len(co0[2]) <= 31 else
                                     "k_factor3 (fused build + factor + forward solve, one column per lane)")}.get(
                           getattr(eng, "kernel_used", ""), "k_factor"),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": fac_avg_ms, "launches_timed": len(fac_ms),
            "rows_per_launch": fac_avg_rows,
            "algorithmic_bytes_per_launch": alg_bytes,
            "fp64_valu_frac": alg_flops / (fac_avg_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import cref
            cn = args.cpu_n if args.cpu_n > 0 else N
            k0 = walkers(args.warmup)[0]            # first timed evaluation
            co = k0.get_device_coefficients()
            work = np.empty(cn * (3 * W + 3) + W)
            c0 = time.perf_counter()
            ref, info = cref.loglike(co[:6], t[:cn], np.full(cn, yerr ** 2) + co[6],
                                     y[:cn], work=work)
            cdt = time.perf_counter() - c0
            result["cpu_baseline"] = {
                "value": (cn / N) / cdt if cn != N else 1.0 / cdt,
                "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"1 evaluation of the first timed walker on {cn} of {N} rows "
                          f"({cdt:.2f} s, gcc -O3 -march=x86-64-v3 restatement of the "
                          "celerite2 algorithm, single thread like celerite2)",
            }
            if cn == N:
                rel = abs(float(lls[args.warmup, 0]) - ref) / abs(ref)
                result["parity"] = {"loglike_rel_err_vs_oracle": rel, "gate": 1e-8}
                if not (info == 0 and rel <= 1e-8):
                    raise SystemExit(f"parity gate failed: rel={rel:.3e}")
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
