"""
Multi-GPU sharding of independent evaluations.

The path shards over independent units -- light curves or MCMC walkers -- with NO data-path
collective: one process per GPU, a static contiguous block partition of the B problems over
the ranks, every rank evaluates its block on its own device, and the B scalars are gathered
at the end (``all_gather`` of ceil(B/world) float64 per rank: RCCL over xGMI on a GPU node,
gloo in the CPU tests).  The gather is the only communication and carries a few KB.
"""
import numpy as np

__all__ = ["shard_bounds", "sharded_log_likelihood", "gather_results"]


def shard_bounds(B, world, rank):
    """Contiguous block partition: ranks 0..r-1 get one extra item when B % world = r."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_results(local, B, group=None, device=None):
    """All ranks' blocks of results -> the full (B,) array on every rank: ONE ``all_gather`` of
    ceil(B / world) float64 per rank (RCCL over xGMI with the nccl backend, gloo on CPU).  ``local`` is
    this rank's block of the static partition (:func:`shard_bounds`).  Runs the collective at any
    world size, 1 included (the GPU test drives it that way on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(B, world, rank)
    local = np.asarray(local, dtype=np.float64)
    if local.shape != (hi - lo,):
        raise ValueError("dimension mismatch")
    width = -(-B // world)
    backend = dist.get_backend(group)
    dev = torch.device(device if device is not None else
                       (f"cuda:{torch.cuda.current_device()}" if backend == "nccl" else "cpu"))
    buf = torch.full((width,), float("nan"), dtype=torch.float64, device=dev)
    buf[:hi - lo] = torch.as_tensor(local, dtype=torch.float64, device=dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    res = np.empty(B)
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        res[a:b] = out[r][:b - a].cpu().numpy()
    return res


def sharded_log_likelihood(kernels, t, y, yerr=None, diag=None, mean=0.0, evaluate=None,
                           group=None, device=None):
    """
    log-likelihoods of B problems evaluated across the ranks of ``torch.distributed``.

    ``kernels`` is the full list of B kernels on every rank; ``t`` / ``y`` are either shared
    ((N,)), per problem ((B, N)), or ragged (lists of B series of different lengths; ``yerr`` / ``diag`` then a
    scalar or a list of per-series arrays).  ``evaluate(kernels, t, y, yerr=..., diag=..., mean=...)
    -> (b,) array`` defaults to :func:`gadfly_amd.log_likelihood_batch` on this rank's GPU
    (the tests inject a checker so the partition/gather logic runs under gloo without a GPU).
    Returns the full (B,) numpy array on every rank.
    """
    import torch.distributed as dist

    if evaluate is None:
        from .batch import log_likelihood_batch

        def evaluate(ks, tt, yy, **kw):
            return log_likelihood_batch(ks, tt, yy, device=device, **kw)

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = len(kernels)
    lo, hi = shard_bounds(B, world, rank)

    def part(x):
        if x is None:
            return None
        if isinstance(x, (list, tuple)) and len(x) == B and all(np.ndim(v) >= 1 for v in x) \
                and len({np.shape(v) for v in x}) > 1:
            return list(x[lo:hi])           # ragged: one series per problem, different lengths
        x = np.asarray(x)
        if x.ndim == 1 and x.dtype == object and x.shape[0] == B:
            return list(x[lo:hi])
        return x[lo:hi] if (x.ndim == 2 and x.shape[0] == B) else x

    local = np.zeros(0)
    if hi > lo:
        local = np.asarray(evaluate(kernels[lo:hi], part(t), part(y), yerr=part(yerr),
                                    diag=part(diag), mean=mean), dtype=np.float64)
    if world == 1:
        return local
    return gather_results(local, B, group=group, device=device)
