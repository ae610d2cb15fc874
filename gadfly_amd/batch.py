"""
Batched log-likelihood front-end: B independent evaluations on one GPU.

One "evaluation" is what BASELINE.json's metric counts: for fresh
hyperparameters, ``compute`` (matrix build + factor, /root/reference/gadfly/gp.py:202) followed
by ``log_likelihood`` (forward solve + reductions, gp.py:350).  Here the factor and the
forward solve share one sweep and nothing returns to the host until the B scalars are read.

Two batch shapes:
  * walkers      : shared t, y; one kernel (hyperparameter set) per walker;
  * light curves : own t, y (and kernel) per problem, common N.
"""
import numpy as np

from .engine import StreamingBatch

__all__ = ["BatchedLogLikelihood", "log_likelihood_batch"]


class BatchedLogLikelihood:
    """Reusable evaluator: device buffers are allocated once; each :meth:`evaluate` with new
    kernels costs an O(B J) coefficient upload plus the device work."""

    def __init__(self, kernels, t, y, yerr=None, diag=None, mean=0.0, device=None,
                 tile_rows=8192, overlap_build=False):
        if yerr is not None and diag is not None:
            raise ValueError("only one of 'diag' and 'yerr' can be provided")
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if np.any(np.diff(t, axis=-1) < 0.0):
            raise ValueError("The input coordinates must be sorted")
        if y.shape[-1] != t.shape[-1]:
            raise ValueError("dimension mismatch")
        d = None
        if yerr is not None:
            d = np.asarray(yerr, dtype=np.float64) ** 2
        elif diag is not None:
            d = np.asarray(diag, dtype=np.float64)
        if d is not None and d.ndim == 0:
            d = np.full(t.shape[-1], float(d))
        self.engine = StreamingBatch([k.get_device_coefficients() for k in kernels], t,
                                     y - mean, diag=d, tile_rows=tile_rows, device=device,
                                     overlap_build=overlap_build)
        #: keep the row generator's share of the relative log-likelihood error below this by
        #: choosing its re-anchoring period from the measured conditioning
        self.generator_target = 1e-9
        self.auto_generator_period = True

    @property
    def B(self):
        return self.engine.B

    def pack(self, kernels):
        return self.engine.pack_coefficients(
            [k.get_device_coefficients() for k in kernels])

    def evaluate_device(self, pack=None):
        """Enqueue one evaluation per problem; returns the (B,) device tensor."""
        eng = self.engine
        if pack is not None:
            eng.use_coefficients(pack)
        # small batches of long series are chunked in time as well (exact, see engine.evaluate)
        return eng.evaluate()[0]

    def evaluate(self, kernels=None):
        out = self.evaluate_device(None if kernels is None else self.pack(kernels))
        res = out.cpu().numpy()
        if self.auto_generator_period:
            # the result copy synchronised anyway: adapt the generator period of the NEXT
            # evaluation to the conditioning just seen (hyperparameters move slowly in a sampler)
            self.engine.calibrate_generator(self.generator_target)
        return res

    def calibrate(self):
        """After asynchronous evaluations (:meth:`evaluate_device`): set the generator period
        from the condition estimate of the last one.  Returns (condition, period)."""
        return self.engine.calibrate_generator(self.generator_target)


def log_likelihood_batch(kernels, t, y, yerr=None, diag=None, mean=0.0, device=None):
    """log-likelihoods of B problems (numpy array of shape (B,)); -inf where K is not
    positive definite (celerite2's ``quiet=True`` convention)."""
    return BatchedLogLikelihood(kernels, t, y, yerr=yerr, diag=diag, mean=mean,
                                device=device).evaluate()
