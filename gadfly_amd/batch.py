"""
Batched log-likelihood front-end: B independent evaluations on one GPU.

One "evaluation" is what BASELINE.json's metric counts: for fresh
hyperparameters, ``compute`` (matrix build + factor, /root/reference/gadfly/gp.py:202) followed
by ``log_likelihood`` (forward solve + reductions, gp.py:350).  Here the factor and the
forward solve share one sweep and nothing returns to the host until the B scalars are read.

Batch shapes:
  * walkers      : shared t, y; one kernel (hyperparameter set) per walker;
  * light curves : own t, y (and kernel) per problem -- a (B, N) array when the series share N, or a LIST of B
                   series of different lengths (real Kepler quarters, /root/reference/gadfly/core.py:509-512,
                   psd.py:483-531): every series is then extended to the longest one with MISSING-DATA rows -- its
                   own cadence continued, y = 0 and the diagonal PAD_DIAG = 2^1000.  Such a row has the pivot 2^1000
                   exactly, the gain 2^-1000 (no trace in the state at double precision) and z^2 / d of order 2^-1000: the sums
                   over the real rows are untouched, and what a pad row does add -- log 2^1000 to sum log d, one more
                   row to N log 2 pi -- is a known constant taken off again per problem.  No kernel knows about it.
"""
import numpy as np

from .engine import StreamingBatch

__all__ = ["BatchedLogLikelihood", "log_likelihood_batch", "sho_coefficient_pack"]


def sho_coefficient_pack(S0, w0, Q, delta, eps=1e-5):
    """
    Celerite coefficients of B exposure-integrated sums of J SHO terms at once: the term algebra of
    ``TermConvolution(TermSum(SHOTerm x J), delta)`` (what ``StellarOscillatorKernel`` is,
    /root/reference/gadfly/core.py:371-394) vectorised over the batch,
    so that a sampler proposing hyperparameter arrays does not build B x J Python objects per step
    (1.5 s for 2048 walkers of 30 terms against a millisecond here).

    ``S0, w0, Q``: arrays of shape (B, J); ``delta``: exposure in 1/uHz (scalar or (B,)).  Terms with
    Q < 1/2 become two real exponentials each, so the overdamped pattern must be the same for every
    problem of the batch.  Returns ``(Jr, Jc, real, comp, diag_add, c)`` in the layout of the
    per-object path (:func:`gadfly_amd.engine._coeff_pack`), with identical values.
    """
    S0, w0, Q = (np.atleast_2d(np.asarray(v, dtype=np.float64)) for v in (S0, w0, Q))
    if not (S0.shape == w0.shape == Q.shape):
        raise ValueError("dimension mismatch")
    B = S0.shape[0]
    over = Q < 0.5
    if np.any(over != over[0]):
        raise ValueError("all problems of a batch must share the term structure "
                         "(the same terms overdamped, Q < 1/2, in every problem)")
    over = over[0]
    und = ~over
    delta = np.broadcast_to(np.asarray(delta, dtype=np.float64), (B,))[:, None]
    # overdamped terms: two real exponentials each, in term order (SHOTerm.get_coefficients)
    So, wo, Qo = S0[:, over], w0[:, over], Q[:, over]
    f = np.sqrt(np.maximum(1.0 - 4.0 * Qo * Qo, eps))
    amp = 0.5 * So * wo * Qo
    ar = np.stack([amp * (1.0 + 1.0 / f), amp * (1.0 - 1.0 / f)], axis=-1).reshape(B, -1)
    cr = np.stack([0.5 * wo / Qo * (1.0 - f), 0.5 * wo / Qo * (1.0 + f)], axis=-1).reshape(B, -1)
    # underdamped terms (Q == 1/2 lands here with f = sqrt(eps))
    Su, wu, Qu = S0[:, und], w0[:, und], Q[:, und]
    f = np.sqrt(np.maximum(4.0 * Qu * Qu - 1.0, eps))
    a = Su * wu * Qu
    cc = 0.5 * wu / Qu
    # exposure integration (TermConvolution.get_coefficients / get_diag_shift), complex form
    A = np.concatenate([ar.astype(np.complex128), a - 1j * (a / f)], axis=1)
    z = np.concatenate([cr.astype(np.complex128), cc - 1j * (cc * f)], axis=1)
    zd = z * delta
    with np.errstate(over="ignore", invalid="ignore"):
        Ap = 2.0 * A * (np.cosh(zd) - 1.0) / zd ** 2
        shift = np.sum((2.0 * A * (zd - np.sinh(zd)) / zd ** 2).real, axis=1)
    Jr, Jc = ar.shape[1], a.shape[1]
    real = np.zeros((2, B, max(Jr, 1)))
    comp = np.zeros((4, B, max(Jc, 1)))
    real[0, :, :Jr], real[1, :, :Jr] = Ap[:, :Jr].real, z[:, :Jr].real
    comp[0, :, :Jc], comp[1, :, :Jc] = Ap[:, Jr:].real, -Ap[:, Jr:].imag
    comp[2, :, :Jc], comp[3, :, :Jc] = z[:, Jr:].real, -z[:, Jr:].imag
    diag_add = np.sum(real[0, :, :Jr], axis=1) + np.sum(comp[0, :, :Jc], axis=1) + shift
    c = np.zeros((B, Jr + 2 * Jc))
    c[:, :Jr] = real[1, :, :Jr]
    c[:, Jr::2] = comp[2, :, :Jc]
    c[:, Jr + 1::2] = comp[2, :, :Jc]
    return Jr, Jc, real, comp, diag_add, c


class _Pack(tuple):
    """A coefficient pack of the engine that remembers whether its kernels are positive semi-definite by
    construction (sums of SHO terms with positive S0, w0, Q, possibly exposure-integrated)."""
    psd_safe = False


def _sho_only(kernel, dt_min=None, t_abs_max=0.0):
    """True when K + diag(>= 0) is positive semi-definite by construction, or so close to it that only rounding
    or cadence jitter stands in the way: a sum of SHO terms with S0, w0, Q > 0 is a covariance function whatever the
    hyperparameters; its exposure-integrated form (``TermConvolution``, every kernel gadfly builds) is one only
    where the celerite coefficients represent it exactly, i.e. for lags >= delta -- closer time stamps get the
    un-integrated form's continuation, which need not be a covariance (``TermConvolution(SHOTerm(S0=1, w0=2,
    Q=0.7), 1)`` on ``arange(200) * 0.2`` has 33 negative eigenvalues).  ``dt_min``: the smallest spacing of the
    time axis the kernel is evaluated on (None: unknown -> False for an integrated kernel); see EXPOSURE_SLACK for
    stamps marginally closer than delta."""
    from .terms import SHOTerm, TermConvolution
    base = kernel
    if isinstance(kernel, TermConvolution):
        base = kernel.term
        if not _exposure_resolved(float(kernel.delta), dt_min, t_abs_max):
            return False
    terms = getattr(base, "terms", None)
    if terms is None:
        terms = (base,)
    return len(terms) > 0 and all(type(t) is SHOTerm and t.S0 > 0.0 and t.w0 > 0.0 and t.Q > 0.0 for t in terms)


#: how far below the exposure time the closest pair of stamps may sit for the two-sweep route to be taken: real
#: cadences jitter (barycentric corrections, +-0.2 s in cfg3's jittered variant), and for lags this close to delta the
#: celerite form departs from the integrated kernel only to second order in the shortfall.  Such a matrix is no longer
#: a covariance BY CONSTRUCTION -- what guarantees the result there is the corrections' check of every pivot's sign
#: (gadfly_dense.hip: k_corr_small / k_pchol + k_spd_check), which sends anything indefinite to the final pass.
#: Grossly unresolved exposures (the advisor's example: stamps 0.2 apart, delta = 1) are indefinite for sure: not taken.
EXPOSURE_SLACK = 0.1


def _exposure_resolved(delta, dt_min, t_abs_max=0.0):
    """No two time stamps closer than (1 - EXPOSURE_SLACK) of the exposure time ``delta`` (and the rounding of the
    stamps themselves: a one-minute cadence on a JD-based axis jitters by 5e-7 of the spacing)."""
    if delta <= 0.0:
        return True
    if dt_min is None:
        return False
    tol = max(1e-9 * delta, 4.0 * float(np.spacing(abs(t_abs_max))))
    return dt_min >= (1.0 - EXPOSURE_SLACK) * delta - tol


#: diagonal of a missing-data row (ragged batches): a power of two, so that a + 2^1000 and the pivot are 2^1000 exactly
PAD_DIAG = 2.0 ** 1000


def _is_ragged(t):
    """A list / tuple / object array of 1-D series (not one rectangular array)."""
    if isinstance(t, np.ndarray):
        return t.dtype == object
    if isinstance(t, (list, tuple)) and len(t) and all(np.ndim(x) == 1 for x in t):
        return len({len(x) for x in t}) > 1         # (equal lengths: an ordinary (B, N) array)
    return False


def _pad_ragged(t, y, d, mean):
    """B series of different lengths -> rectangular (B, Nmax) arrays with missing-data rows at the end.
    Returns (t, y - mean, diag, rows per problem, largest real diagonal per problem, smallest real spacing,
    largest real |t|)."""
    B = len(t)
    ts = [np.ascontiguousarray(x, dtype=np.float64) for x in t]
    if not isinstance(y, (list, tuple)) and not (isinstance(y, np.ndarray) and y.dtype == object):
        raise ValueError("dimension mismatch")
    ys = [np.ascontiguousarray(x, dtype=np.float64) for x in y]
    if len(ys) != B or any(a.ndim != 1 or a.shape != b.shape for a, b in zip(ts, ys)):
        raise ValueError("dimension mismatch")
    rows = np.array([len(x) for x in ts], dtype=np.int64)
    if np.any(rows < 1):
        raise ValueError("dimension mismatch")
    if any(np.any(np.diff(x) < 0.0) for x in ts):
        raise ValueError("The input coordinates must be sorted")
    if d is None:
        ds = [np.zeros(n) for n in rows]
    elif isinstance(d, (list, tuple)) or (isinstance(d, np.ndarray) and d.dtype == object):
        ds = [np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)) for x, n in zip(d, rows)]
        if len(ds) != B:
            raise ValueError("dimension mismatch")
    else:
        d = np.asarray(d, dtype=np.float64)
        if d.ndim > 1 or (d.ndim == 1 and d.shape[0] != B):
            raise ValueError("dimension mismatch")          # a scalar, or one value per problem
        ds = [np.full(n, float(d if d.ndim == 0 else d[i])) for i, n in enumerate(rows)]
    means = np.broadcast_to(np.asarray(mean, dtype=np.float64), (B,)) if np.ndim(mean) <= 1 else None
    if means is None:
        raise ValueError("dimension mismatch")
    Nmax = int(rows.max())
    T = np.empty((B, Nmax))
    Y = np.zeros((B, Nmax))
    D = np.full((B, Nmax), PAD_DIAG)
    for i, n in enumerate(rows):
        T[i, :n], Y[i, :n], D[i, :n] = ts[i], ys[i] - means[i], ds[i]
        if n < Nmax:
            # the series' own cadence continued: the row generator keeps stepping, no gap, no new phase range
            dt = float(np.median(np.diff(ts[i]))) if n > 1 else 1.0
            if not dt > 0.0:
                dt = 1.0
            T[i, n:] = ts[i][-1] + dt * np.arange(1, Nmax - n + 1)
    dmax = np.array([float(np.max(x)) if len(x) else 0.0 for x in ds])
    dmin_real = min(float(np.min(x)) for x in ds)
    dt_min = min((float(np.min(np.diff(x))) for x in ts if len(x) > 1), default=None)
    tabs = max(float(np.max(np.abs(x))) for x in ts)
    return T, Y, D, rows, dmax, dmin_real, dt_min, tabs


class BatchedLogLikelihood:
    """Reusable evaluator: device buffers are allocated once; each :meth:`evaluate` with new
    kernels costs an O(B J) coefficient upload plus the device work.

    ``t`` / ``y``: shared ((N,)), per problem ((B, N)), or -- ragged -- lists of B series of different lengths
    (``yerr`` / ``diag`` then a scalar, one value per problem, or a list of per-series arrays; ``mean`` a scalar or
    one value per problem)."""

    def __init__(self, kernels, t, y, yerr=None, diag=None, mean=0.0, device=None,
                 tile_rows=8192, overlap_build=False):
        if yerr is not None and diag is not None:
            raise ValueError("only one of 'diag' and 'yerr' can be provided")
        self.rows = None                    # ragged batches: real rows per problem
        self._pad_corr = None
        if _is_ragged(t):
            if len(t) != len(kernels):
                raise ValueError("dimension mismatch")
            dd = diag
            if yerr is not None:
                dd = ([np.asarray(e, dtype=np.float64) ** 2 for e in yerr]
                      if isinstance(yerr, (list, tuple)) or (isinstance(yerr, np.ndarray) and yerr.dtype == object)
                      else np.asarray(yerr, dtype=np.float64) ** 2)
            t, ym, d, self.rows, dmax, dmin_real, dt_min, tabs = _pad_ragged(t, y, dd, mean)
            self.engine = StreamingBatch([k.get_device_coefficients() for k in kernels], t, ym, diag=d,
                                         tile_rows=tile_rows, device=device, overlap_build=overlap_build)
            eng = self.engine
            torch = eng.torch
            # the condition estimates look at the REAL rows' diagonal, and the results lose the pad rows' constants
            eng._diag_amax = torch.as_tensor(dmax, dtype=torch.float64, device=eng.device)
            pad = (t.shape[1] - self.rows).astype(np.float64)
            self._pad_corr = torch.as_tensor(0.5 * pad * (np.log(PAD_DIAG) + np.log(2.0 * np.pi)),
                                             dtype=torch.float64, device=eng.device)
            self._diag_nonneg = dmin_real >= 0.0
            self._dt_min, self._t_abs_max = dt_min, tabs
            self._finish_init(kernels)
            return
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
        self._diag_nonneg = d is None or bool(np.all(np.asarray(d) >= 0.0))
        #: smallest spacing of the time axes and their largest |t|: an exposure-integrated kernel is a covariance
        #: only while no two stamps are closer than its exposure time (see _sho_only)
        self._dt_min = float(np.min(np.diff(t, axis=-1))) if t.shape[-1] > 1 else None
        self._t_abs_max = float(np.max(np.abs(t))) if t.size else 0.0
        self._finish_init(kernels)

    def _finish_init(self, kernels):
        #: the time-parallel route may drop its final pass (engine.two_sweep) when the matrix is positive
        #: semi-definite by construction: SHO kernels, a non-negative diagonal, no two time stamps closer than
        #: the exposure time.  Rounding can still break a pivot: the corrections check the sign of EVERY pivot
        #: of a chunk (a Cholesky attempt on (I - X G) X, gadfly_dense.hip: k_spd_check), not only the parity
        #: det(I - X G) gives, and a chunk that fails it leaves a non-finite value, which :meth:`resolve` repeats
        #: with the final pass
        self._init_safe = self._diag_nonneg and all(_sho_only(k, self._dt_min, self._t_abs_max) for k in kernels)
        self.two_sweep = True
        #: keep the row generator's share of the relative log-likelihood error below this by
        #: choosing its re-anchoring period from the measured conditioning
        self.generator_target = 1e-9
        self.auto_generator_period = True
        #: evaluations whose accuracy guard has not been looked at yet: (out, flag, pack, period)
        self._unresolved = []
        #: evaluations repeated with exact generator rows by the guard so far
        self.guard_reruns = 0
        self._last_cond = None

    @property
    def B(self):
        return self.engine.B

    def pack(self, kernels):
        pk = _Pack(self.engine.pack_coefficients([k.get_device_coefficients() for k in kernels]))
        pk.psd_safe = all(_sho_only(k, self._dt_min, self._t_abs_max) for k in kernels)
        return pk

    def pack_parameters(self, S0, w0, Q, delta):
        """Coefficient pack straight from (B, J) hyperparameter arrays (:func:`sho_coefficient_pack`):
        the vectorised form of :meth:`pack` for ``StellarOscillatorKernel``-type kernels."""
        pk = _Pack(self.engine.pack_arrays(*sho_coefficient_pack(S0, w0, Q, delta)))
        pk.psd_safe = bool(np.all(np.asarray(S0) > 0.0) and np.all(np.asarray(w0) > 0.0)
                           and np.all(np.asarray(Q) > 0.0)
                           and all(_exposure_resolved(float(dl), self._dt_min, self._t_abs_max)
                                   for dl in np.atleast_1d(np.asarray(delta, dtype=np.float64))))
        return pk

    def evaluate_device(self, pack=None):
        """Enqueue one evaluation per problem; returns the (B,) device tensor (no host sync).

        Accuracy guard (device side): with a generator period > 1 the rows between anchors carry a
        rotation error that the log-likelihood amplifies by the problem's condition number, and the
        period was chosen from the conditioning of EARLIER evaluations.  Every evaluation therefore
        also leaves a flag per problem on the device -- GEN_ERR * period * max(a) / min(d) above the
        target, from the min pivot its own reduction returns -- and :meth:`resolve` (called by
        :meth:`evaluate`, :meth:`calibrate`, or by the caller before reading asynchronous results)
        repeats flagged evaluations with exact rows, writing the corrected values into the tensor
        returned here.  A walker that wanders into a badly conditioned region is thus never silently
        evaluated at a period its conditioning does not allow.
        """
        eng = self.engine
        if pack is not None:
            eng.use_coefficients(pack)
            self._safe = self._diag_nonneg and bool(getattr(pack, "psd_safe", False))
        elif not hasattr(self, "_safe"):
            self._safe = self._init_safe
        eng.two_sweep = bool(self.two_sweep and self._safe)
        # small batches of long series are chunked in time as well (exact, see engine.evaluate)
        out = eng.evaluate()[0]
        if self._pad_corr is not None:
            out = out + self._pad_corr      # ragged batch: the missing-data rows' constants (-inf / NaN stay)
        period = int(eng.generator_period)
        if eng._fused_ok() or eng._wide_ok():
            torch = eng.torch
            acc = eng.last_acc()
            amax = eng._pack[2] if eng.diag is None else eng._pack[2] + eng._diag_amax
            # min pivot and largest diagonal of THIS evaluation (what calibrate() looks at; the
            # engine's own state may belong to a guard rerun of an older pack by then); a two-sweep
            # evaluation has the nominal pass' pivots only: margin (engine.TWO_SWEEP_MARGIN)
            dmin = acc[:, 2].clone()
            if getattr(eng, "_two_sweep_used", False):
                dmin = dmin / eng.TWO_SWEEP_MARGIN
            self._last_cond = (dmin, amax)
            flag = None
            if period > 1:
                # a non-positive pivot (failed factorisation: -inf either way) is not an accuracy case
                # (GEN_ERR * period + the phase-quantum term of a time axis far from zero: engine.PHASE_ERR)
                flag = (eng.generator_error_coefficient(period) * amax > self.generator_target * dmin) & (dmin > 0)
            if getattr(eng, "_two_sweep_used", False):
                # no final pass ran: a pivot that rounding pushed below zero inside a chunk shows up as a
                # non-finite value (det(I - X G) <= 0) -- repeated with the final pass by resolve()
                bad = ~torch.isfinite(out)
                flag = bad if flag is None else (flag | bad)
            if flag is not None:
                self._unresolved.append((out, flag, eng._pack, period))
                if len(self._unresolved) > 64:      # bound the backlog of a caller that never resolves
                    self.resolve()
        return out

    def resolve(self):
        """Look at the accuracy guards of all evaluations enqueued since the last call (one host
        sync) and repeat the flagged ones with exact generator rows, in place.  Returns the number
        of evaluations repeated."""
        if not self._unresolved:
            return 0
        eng = self.engine
        torch = eng.torch
        pending, self._unresolved = self._unresolved, []
        flags = torch.stack([f for _, f, _, _ in pending]).cpu().numpy()      # the sync
        redone = 0
        keep_pack, keep_period, keep_two = eng._pack, eng.generator_period, eng.two_sweep
        for (out, flag, pack, _), hit in zip(pending, flags):
            if not hit.any():
                continue
            # (the pack is switched directly: use_coefficients() would also drop the engine's
            # construction-time coefficient list, which the stored-factor classes still need)
            eng._pack = pack
            eng.generator_period = 1
            eng.two_sweep = False           # the repeat is the reference formulation: exact rows, final pass
            exact = eng.evaluate()[0]
            if self._pad_corr is not None:
                exact = exact + self._pad_corr
            out.copy_(torch.where(flag, exact, out))
            redone += int(hit.sum())
        eng._pack, eng.generator_period, eng.two_sweep = keep_pack, keep_period, keep_two
        self.guard_reruns += redone
        return redone

    def evaluate(self, kernels=None):
        out = self.evaluate_device(None if kernels is None else self.pack(kernels))
        if (self.auto_generator_period and self._last_cond is not None and len(self._unresolved) <= 1
                and (not self._unresolved or self._unresolved[0][0] is out)):
            # the common case -- nothing else pending: the values, the guard's flags and the two numbers the
            # period calibration reads come back in ONE copy (a chain of single evaluations paid four round trips)
            torch = self.engine.torch
            dmin, amax = self._last_cond
            parts = [out, dmin.min().reshape(1), amax.max().reshape(1)]
            if self._unresolved:
                parts.append(self._unresolved[0][1].to(out.dtype))
            host = torch.cat(parts).cpu().numpy()
            B = out.shape[0]
            if not self._unresolved or not host[B + 2:].any():
                self._unresolved = []
                eng = self.engine
                cond = host[B + 1] / host[B] if host[B] > 0.0 else float("inf")
                eng.generator_period = eng.period_for_condition(float(cond), self.generator_target)
                return host[:B].copy()
        self.resolve()
        res = out.cpu().numpy()
        if self.auto_generator_period:
            # the result copy synchronised anyway: adapt the generator period of the NEXT
            # evaluation to the conditioning just seen (hyperparameters move slowly in a sampler)
            self._calibrate_last()
        return res

    def _calibrate_last(self):
        """(condition, period) from the min pivots of the last evaluation enqueued HERE."""
        eng = self.engine
        if self._last_cond is None:
            return eng.calibrate_generator(self.generator_target)
        dmin, amax = self._last_cond
        dmin = float(dmin.min().item())
        cond = float(amax.max().item()) / dmin if dmin > 0.0 else float("inf")
        eng.generator_period = eng.period_for_condition(cond, self.generator_target)
        return cond, eng.generator_period

    def calibrate(self):
        """After asynchronous evaluations (:meth:`evaluate_device`): set the generator period
        from the condition estimate of the last one.  Returns (condition, period)."""
        self.resolve()
        return self._calibrate_last()


def log_likelihood_batch(kernels, t, y, yerr=None, diag=None, mean=0.0, device=None):
    """log-likelihoods of B problems (numpy array of shape (B,)); -inf where K is not
    positive definite (celerite2's ``quiet=True`` convention)."""
    return BatchedLogLikelihood(kernels, t, y, yerr=yerr, diag=diag, mean=mean,
                                device=device).evaluate()
