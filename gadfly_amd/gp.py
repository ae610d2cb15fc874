"""
``GaussianProcess``: gadfly's units-aware GP interface, computed on an MI355X.

API mirror of /root/reference/gadfly/gp.py:13-395 (``GaussianProcess`` subclassing
``celerite2.GaussianProcess``).  The reference's methods strip astropy units and call
``super()``; here the ``super()`` part -- celerite2's ``BaseGaussianProcess`` state
machine (compute / recompute / log_likelihood / apply_inverse / dot_tril / sample /
condition / predict) -- is restated in this class and every O(N)
operation runs through the HIP C-ABI (:mod:`gadfly_amd.engine`).  There is no CPU
compute path.

Error behaviour follows celerite2: unsorted ``t`` -> ``ValueError``; both ``yerr`` and
``diag`` -> ``ValueError``; a method before ``compute`` -> ``RuntimeError``; a
non-positive pivot -> ``LinAlgError`` unless ``quiet=True``, in which case the
log-determinant is ``-inf`` and the log-likelihood ``-inf`` (gp.py:188-192).
"""
import numpy as np

from . import _lib
from . import units as _units
from .engine import DeviceBatch, StreamingBatch, LOG_2PI

__all__ = ["GaussianProcess", "ConditionalDistribution", "LinAlgError"]


class LinAlgError(np.linalg.LinAlgError):
    """celerite2.driver.LinAlgError equivalent: 'failed to factorize or solve matrix'."""


class _ConstantMean:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, x):
        return np.full(np.shape(x), self.value, dtype=np.float64)


class ConditionalDistribution:
    """Conditional (predictive) distribution (celerite2 ``ConditionalDistribution``;
    built at /root/reference/gadfly/gp.py:232).  Everything runs on the device: ``mean`` through the
    chunk-parallel conditional-mean sweeps, ``variance`` / ``covariance`` through the K(t, t*)
    construction celerite2 uses, built by ``gf_cross_covariance`` in blocks of 64 query times and
    solved with the chunk-parallel multi-right-hand-side sweeps."""

    def __init__(self, gp, y, t=None, *, include_mean=True, kernel=None):
        self.gp = gp
        self.y = gp._process_input(y, require_vector=True)
        self.t = None if t is None else np.ascontiguousarray(t, dtype=np.float64)
        if self.t is not None and self.t.ndim != 1:
            raise ValueError("dimension mismatch")
        self.include_mean = include_mean
        self.kernel = kernel
        self._alpha = None
        self._resid_dev = None

    def _get_alpha(self):
        if self._alpha is None:
            gp = self.gp
            resid = gp._resid_to_device(self.y)
            self._resid_dev = resid
            self._alpha = gp._engine.apply_inverse(resid.reshape(1, -1, 1))
        return self._alpha

    @property
    def mean(self):
        gp = self.gp
        alpha = self._get_alpha()
        if self.t is None and self.kernel is None:
            if gp._diag_dev is not None and self._resid_dev is not None:
                # y - diag alpha = (y - mean) - diag alpha + mean, formed where alpha is: one fused pass on the
                # device and one download instead of a download and three passes over host arrays
                torch = gp._engine.torch
                mu_d = torch.addcmul(self._resid_dev.reshape(-1), gp._diag_dev.reshape(-1), alpha.reshape(-1),
                                     value=-1.0)
                const = isinstance(gp._mean, _ConstantMean)
                if self.include_mean and const:
                    mu_d += float(gp._mean.value)
                mu = gp._to_host(mu_d)
                if self.include_mean and not const:
                    mu += gp._mean_value
                return mu
            mu = self.y - gp._diag * gp._to_host(alpha.reshape(-1))
            if not self.include_mean:
                mu = mu - gp._mean_value
            return mu
        xs = gp._t if self.t is None else self.t
        if np.any(np.diff(xs) < 0.0):
            raise ValueError("The input coordinates must be sorted")
        if self.kernel is None:
            eng, other = gp._engine, None
        else:
            # different kernel: its generator matrices at the observed times and at t*
            other = DeviceBatch([self.kernel.get_device_coefficients()], gp._t,
                                device=gp._engine.device)
            eng = other
        ts, Us, Vs = eng.matrices_at(xs)
        mu = gp._engine.predict_at(alpha.reshape(1, -1), ts, Us, Vs, other=other)
        mu = gp._to_host(mu.reshape(-1))
        if self.include_mean:
            mu = mu + gp._mean(xs)
        return mu

    #: query times per cross-covariance block (= right-hand sides per chunk-parallel sweep)
    _RB = 64

    def _cross_blocks(self):
        """K(t, t*) and K^-1 K(t, t*) on the device, in blocks of ``_RB`` query columns.

        celerite2 builds the dense N x M block with numpy on the host (O(N M J) transcendentals and
        an N x M x J temporary); here ``gf_cross_covariance`` writes it straight into the layout the
        multi-right-hand-side sweeps take.  The few entries closer than the exposure time (where
        an exposure-integrated kernel departs from its celerite coefficients) are patched from
        :meth:`Term.get_value`.  Yields (column slice, K block (N, R), K^-1 K block (N, R)).
        """
        gp = self.gp
        kernel = gp.kernel if self.kernel is None else self.kernel
        xs = gp._t if self.t is None else self.t
        torch = gp._engine.torch
        lib, p = _lib.load(), _lib.ptr
        dev = gp._engine.device
        co = kernel.get_device_coefficients()
        cod = [torch.as_tensor(np.ascontiguousarray(v, dtype=np.float64), device=dev) for v in co[:6]]
        Jr, Jc = len(co[0]), len(co[2])
        t_d = gp._to_device(gp._t)
        N = gp._size
        delta = float(getattr(kernel, "delta", 0.0) or 0.0)
        st = torch.cuda.current_stream(dev).cuda_stream
        for j0 in range(0, len(xs), self._RB):
            with torch.cuda.device(dev):        # launches below go to the engine's GPU
                xb = np.ascontiguousarray(xs[j0:j0 + self._RB])
                R = len(xb)
                xb_d = torch.as_tensor(xb, device=dev)
                K = torch.empty((1, N, R), dtype=torch.float64, device=dev)
                rc = lib.gf_cross_covariance(1, N, R, Jr, Jc, *[p(v) if v.numel() else None for v in cod],
                                             p(t_d), 0, p(xb_d), 0, p(K), st)
                _lib.check(rc, "gf_cross_covariance")
                if delta > 0.0:
                    lo = np.searchsorted(gp._t, xb - delta, side="right")
                    hi = np.searchsorted(gp._t, xb + delta, side="left")
                    rows = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)]) if np.any(hi > lo) else np.empty(0, int)
                    if rows.size:
                        cols = np.concatenate([np.full(b - a, r) for r, (a, b) in enumerate(zip(lo, hi))])
                        vals = kernel.get_value(gp._t[rows] - xb[cols])
                        K[0, torch.as_tensor(rows, device=dev), torch.as_tensor(cols, device=dev)] = \
                            torch.as_tensor(vals, device=dev)
                sol = gp._engine.apply_inverse(K)
            yield slice(j0, j0 + R), K[0], sol[0]

    @property
    def variance(self):
        gp = self.gp
        kernel = gp.kernel if self.kernel is None else self.kernel
        xs = gp._t if self.t is None else self.t
        out = np.empty(len(xs))
        k0 = kernel.get_value(np.zeros(1))[0]
        for sl, K, sol in self._cross_blocks():
            out[sl] = k0 - (K * sol).sum(dim=0).cpu().numpy()
        return out

    @property
    def covariance(self):
        gp = self.gp
        kernel = gp.kernel if self.kernel is None else self.kernel
        xs = gp._t if self.t is None else self.t
        blocks = list(self._cross_blocks())
        torch = gp._engine.torch
        Kall = torch.cat([b[1] for b in blocks], dim=1).contiguous()            # (N, M)
        Sall = torch.cat([b[2] for b in blocks], dim=1).contiguous()
        N, M = Kall.shape
        # K(t, t*)^T K^-1 K(t, t*) on the library's FP64-MFMA GEMM tiles (gf_bgemm, A transposed)
        lib, p = _lib.load(), _lib.ptr
        with torch.cuda.device(Kall.device):
            # the long dimension N is cut into S slabs (one batch entry each: a tile's K-loop is sequential),
            # the slabs' partial products are added afterwards
            # (at most N M / 4 doubles of partial products: a quarter of one of the two N x M blocks held anyway;
            # 64 slabs of M x M were 8.6 GB at M = 4096)
            S = int(min(64, max(1, N // 4096), max(1, N // (4 * M))))
            Ks = N // S
            part = torch.empty((S, M, M), dtype=torch.float64, device=Kall.device)
            st = torch.cuda.current_stream(Kall.device).cuda_stream
            _lib.check(lib.gf_bgemm(S, 1, 0, M, M, Ks, p(Kall), M, Ks * M, p(Sall), M, Ks * M, None, 0, 0,
                                    p(part), M, M * M, st), "gf_bgemm")
            C = part.sum(dim=0)
            if S * Ks < N:                      # the last N - S Ks rows
                tail = torch.empty((M, M), dtype=torch.float64, device=Kall.device)
                _lib.check(lib.gf_bgemm(1, 1, 0, M, M, N - S * Ks, p(Kall[S * Ks:]), M, 0, p(Sall[S * Ks:]), M, 0,
                                        p(C), M, 0, p(tail), M, 0, st), "gf_bgemm")
                C = tail
        return kernel.get_value(xs[:, None] - xs[None, :]) - C.cpu().numpy()

    def sample(self, *, size=None, regularize=None):
        mu = self.mean
        cov = self.covariance
        if regularize is not None:
            cov[np.diag_indices_from(cov)] += regularize
        return np.random.multivariate_normal(mu, cov, size=size)


class GaussianProcess:
    """
    The ``gadfly`` interface to the GP solver, on the GPU.

    Parameters follow /root/reference/gadfly/gp.py:22-59: ``kernel`` (a
    :class:`~gadfly_amd.terms.Term`, e.g. ``StellarOscillatorKernel``), ``t`` (astropy
    Quantity/Time, or an ndarray already in 1/uHz), ``mean`` (scalar or callable),
    ``light_curve`` (lightkurve-like object with ``time``, ``flux``, ``flux_err``) and
    ``**kwargs`` forwarded to :meth:`compute`.  ``device`` selects the GPU.
    """

    conditional_distribution = ConditionalDistribution

    def __init__(self, kernel, t=None, mean=0.0, light_curve=None, device=None,
                 **kwargs):
        self._original_flux_median = None
        self.kernel = kernel
        self.mean = mean
        self._device = device
        #: rows between exact re-anchorings of the device-side row generator (1 = exact every row)
        self.generator_period = 1
        self._factor = None
        self._fast = None
        self._t = None
        self._t_dev = self._diag_dev = None
        self._mean_value = None
        self._diag = None
        self._size = None
        self._log_det = -np.inf
        self._norm = np.inf

        if t is not None:
            t = self._time_to_freq(t)

        if light_curve is not None:
            t = self._time_to_freq(light_curve.time)
            flux = light_curve.flux
            if hasattr(flux, "unmasked"):
                median_flux = np.nanmedian(flux.unmasked)
            else:
                median_flux = np.median(flux)
            self._original_flux_median = median_flux
            kwargs["yerr"] = self._flux_to_ppm(light_curve.flux_err, is_error=True)

        if t is not None:
            self.compute(t, **kwargs)

    # ---- mean function (celerite2 BaseGaussianProcess.mean) ---------------
    @property
    def mean(self):
        return self._mean

    @mean.setter
    def mean(self, mean):
        self._mean = mean if callable(mean) else _ConstantMean(mean)

    @property
    def mean_value(self):
        if self._mean_value is None:
            raise RuntimeError(
                "'compute' must be executed before accessing mean_value")
        # (a constant mean is kept as the scalar it is -- every use broadcasts -- and expanded for this accessor)
        if np.ndim(self._mean_value) == 0:
            return np.full(self._size, float(self._mean_value), dtype=np.float64)
        return self._mean_value

    # the user diagonal as an array (celerite2's `_diag`): a scalar yerr / diag is expanded on first use -- the
    # device gets it as a fill, and compute() on 1e6 cadences does not spend half a millisecond writing it out
    @property
    def _diag(self):
        if self._diag_arr is None and getattr(self, "_diag_const", None) is not None and self._size is not None:
            self._diag_arr = np.full(self._size, self._diag_const, dtype=np.float64)
        return self._diag_arr

    @_diag.setter
    def _diag(self, value):
        self._diag_arr = value

    # ---- unit handling (reference gp.py:61-165) ----------------------------
    @staticmethod
    def _time_to_freq(time, freq_unit=None):
        """Times -> 1/uHz.  ndarrays pass through untouched (gp.py:82-84)."""
        if _units.is_time(time):
            time = time.jd * _units.u.day
        if not _units.has_unit(time):
            return time
        _units.require_astropy("a Quantity time axis")
        unit = _units.u.uHz if freq_unit is None else freq_unit
        return time.to(1 / unit).value

    def _flux_to_ppm(self, flux, flux_unit=None, is_error=False):
        """Fluxes -> ppm.  ndarrays pass through untouched (gp.py:111-113)."""
        if isinstance(flux, np.ndarray) and not _units.has_unit(flux):
            return flux
        if not _units.has_unit(flux):
            return np.asarray(flux, dtype=np.float64)
        _units.require_astropy("a Quantity flux")
        u = _units.u
        if flux.unit.is_equivalent(u.electron / u.s):
            # lightkurve fluxes: normalise by the cached median (gp.py:115-124)
            if is_error:
                return 1e6 * (flux / self._original_flux_median).value
            return 1e6 * (flux / self._original_flux_median - 1).value
        return flux.to(u.cds.ppm if flux_unit is None else flux_unit).value

    def _ppm_to_flux(self, value_in_ppm, power=1):
        """ppm -> the light curve's original units (gp.py:128-165), quirks kept."""
        if self._original_flux_median is not None and power == 1:
            return (1e-6 * value_in_ppm + 1) * self._original_flux_median
        elif self._original_flux_median is not None and power == 2:
            return ((1e-6 * value_in_ppm) * self._original_flux_median
                    * self._original_flux_median.unit)
        _units.require_astropy("return_quantity=True")
        return _units.u.Quantity(value_in_ppm, unit=_units.u.cds.ppm)

    # ---- helpers ------------------------------------------------------------
    #: largest array (bytes) that goes through the pinned staging buffer on its way to the device
    PIN_MAX_BYTES = 1 << 26

    def _resid_to_device(self, y):
        """``y - mean`` as a device tensor: the subtraction writes straight into the page-locked staging
        buffer (one pass over y instead of a temporary, a copy and an upload from pageable memory)."""
        import torch
        y = np.ascontiguousarray(y, dtype=np.float64)
        n = y.size
        if y.ndim != 1 or not (0 < n * 8 <= self.PIN_MAX_BYTES):
            return self._to_device(y - self._mean_value)
        pin = getattr(self, "_pin", None)
        if pin is None or pin.numel() < n:
            pin = self._pin = torch.empty((n,), dtype=torch.float64, pin_memory=True)
        stage = pin[:n]
        np.subtract(y, self._mean_value, out=stage.numpy())
        return stage.to(self._device_of())

    def _to_device(self, x):
        """Host array -> float64 device tensor.  Arrays of the usual sizes (one or a few series) pass
        through ONE reusable page-locked buffer: a copy straight from a freshly allocated pageable array
        (``y - mean`` is one) makes the driver lock and unlock its pages, which was measured to stall a
        0.6 ms upload for 15-35 ms every few calls."""
        import torch
        x = np.ascontiguousarray(x, dtype=np.float64)
        dev = self._device_of()
        nbytes = x.size * 8
        if 0 < nbytes <= self.PIN_MAX_BYTES:
            pin = getattr(self, "_pin", None)
            if pin is None or pin.numel() < x.size:
                pin = self._pin = torch.empty((x.size,), dtype=torch.float64, pin_memory=True)
            stage = pin[:x.size]
            np.copyto(stage.numpy(), x.reshape(-1))
            return stage.to(dev).view(x.shape)          # (blocking copy: the buffer is free again on return)
        return torch.as_tensor(x).to(dev)

    def _to_host(self, x):
        """Device tensor -> new float64 numpy array, through a reusable page-locked buffer.  A download
        straight into a fresh pageable array leaves the driver unlocking its pages while the NEXT upload
        waits: 20-30 ms stalls of a 0.2 ms copy, every few calls (measured)."""
        import torch
        n = x.numel()
        if not (0 < n * 8 <= self.PIN_MAX_BYTES):
            return x.cpu().numpy()
        pin = getattr(self, "_pin_out", None)
        if pin is None or pin.numel() < n:
            pin = self._pin_out = torch.empty((n,), dtype=torch.float64, pin_memory=True)
        stage = pin[:n]
        stage.copy_(x.reshape(-1))                      # (blocking: device -> pinned host)
        return stage.numpy().reshape(x.shape).copy()

    def _process_input(self, y, *, require_vector=False):
        if self._t is None:
            raise RuntimeError("The process must be initialized with compute")
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.ndim == 0 or self._t.shape[0] != y.shape[0]:
            raise ValueError("dimension mismatch")
        if require_vector and y.ndim != 1:
            raise ValueError("'y' must be one dimensional")
        if y.ndim > 2:
            raise ValueError("dimension mismatch")
        return y

    # ---- compute (reference gp.py:167-204 + celerite2 base compute) ----------
    def compute(self, t, yerr=None, diag=None, check_sorted=True, quiet=False):
        """Build the generator matrices and factorise K on the device."""
        if _units.is_time(t) or _units.has_unit(t):
            t = self._time_to_freq(t)
        if yerr is not None and _units.has_unit(yerr):
            yerr = self._flux_to_ppm(yerr, is_error=True)
        # reference quirk (gp.py:200): diag is converted only if *yerr* has a unit
        if diag is not None and _units.has_unit(yerr):
            diag = self._flux_to_ppm(diag)

        t = np.ascontiguousarray(t, dtype=np.float64)
        if t.ndim != 1:
            raise ValueError("dimension mismatch")
        N = t.shape[0]
        if check_sorted and N > 1 and not bool(np.all(t[1:] >= t[:-1])):
            raise ValueError("The input coordinates must be sorted")
        self._t = t
        self._size = N
        self._mean_value = (float(self._mean.value) if isinstance(self._mean, _ConstantMean) and np.ndim(self._mean.value) == 0
                            else self._mean(t))
        self._diag = None
        self._t_dev = self._diag_dev = None
        self._diag_const = 0.0              # a scalar diagonal is made on the device, not uploaded
        if yerr is not None and diag is not None:
            raise ValueError("only one of 'diag' and 'yerr' can be provided")
        if yerr is not None and np.ndim(yerr) == 0:
            self._diag_const = float(np.asarray(yerr, dtype=np.float64)) ** 2
        elif yerr is not None:
            self._diag_const = None
            self._diag = np.array(np.broadcast_to(np.asarray(yerr, dtype=np.float64), (N,))) ** 2
        elif diag is not None and np.ndim(diag) == 0:
            self._diag_const = float(np.asarray(diag, dtype=np.float64))
        elif diag is not None:
            self._diag_const = None
            self._diag = np.array(np.broadcast_to(np.asarray(diag, dtype=np.float64), (N,)))
        # statistics of the time axis the engine needs (largest |t|, typical spacing), taken here on the host: as
        # device reductions they cost a 1e6-element sort and two host synchronisations per compute()
        if N > 1:
            if check_sorted:
                tmax = max(abs(float(t[0])), abs(float(t[-1])))
            else:
                tmax = max(abs(float(t.min())), abs(float(t.max())))
            if N <= 65536:
                dt_med = float(np.median(np.diff(t)))
            else:
                # the median spacing of 4097 cadences spread evenly over the series (it only sets how many rows a
                # block of the scaled recurrence may span)
                i = np.linspace(0, N - 2, 4097).astype(np.int64)
                dt_med = float(np.median(t[i + 1] - t[i]))
            self._axis_stats = (tmax, dt_med)
        else:
            self._axis_stats = (abs(float(t[0])) if N else 0.0, 0.0)

        self._do_compute(quiet)

    def recompute(self, *, quiet=False):
        if self._t is None:
            raise RuntimeError("The process must be initialized with compute")
        self._do_compute(quiet)

    # The stored factor (U, V, P, W rows in HBM: what solves, predictions and draws need) is
    # built lazily; `compute` + `log_likelihood` -- the MCMC hot path -- run on the fused
    # engine (time-parallel for long series), which never materialises it.
    @property
    def _engine(self):
        if self._factor is None:
            if self._t is None:
                raise RuntimeError("The process must be initialized with compute")
            if self._fast is not None:
                # time-parallel factorisation kept in scaled form
                self._factor = self._fast.stored_factor()
                info = self._fast.info
            else:
                self._factor = DeviceBatch([self.kernel.get_device_coefficients()], self._t,
                                           diag=self._diag, device=self._device)
                info = self._factor.factor(keep_W=True)
            if int(info[0].item()) and np.isfinite(self._log_det):
                raise LinAlgError("failed to factorize or solve matrix")
        return self._factor

    #: series lengths for which compute() keeps the factor of a W <= 63 kernel (below: the fused sweep is
    #: sequential and a solve would be too; above: 2 KB per row of device memory)
    STORE_MIN_ROWS = 8192
    STORE_MAX_ROWS = 16_000_000

    def _do_compute(self, quiet):
        # celerite2's recompute() refactorises everything: a stored factor of the previous kernel
        # must not survive (apply_inverse / dot_tril / predict / sample rebuild it on first need)
        self._factor = None
        self._fast = None
        co = self.kernel.get_device_coefficients()
        W = len(co[0]) + 2 * len(co[2])
        fast = None
        wide = None
        # (a wide kernel with real terms rides on the fused wide sweep too: engine._complexify_pack)
        if W <= 63 or 2 * (len(co[0]) + len(co[2])) <= 176:
            torch = _lib.require_device()       # (fails loudly without a HIP device: no CPU path)
            f64 = dict(dtype=torch.float64, device=self._device_of())
            const = getattr(self, "_diag_const", None)
            if self._t_dev is None:         # (recompute() with another kernel: t, diag are on the device already)
                self._t_dev = self._to_device(self._t)          # (through the page-locked staging buffer)
                self._diag_dev = (self._to_device(self._diag) if const is None
                                  else torch.full((self._size,), const, **f64))
            fast = StreamingBatch([co], self._t_dev, torch.zeros((self._size,), **f64),
                                  diag=self._diag_dev, device=self._device,
                                  axis_stats=getattr(self, "_axis_stats", None))
            self._axis_stats = fast.axis_stats  # (the same time axis until compute() is called again)
            if fast._wide_ok() and not fast._fused_ok():
                # wide kernel (e.g. the 86-term solar kernel, W = 172): ONE pass of the fused wide sweep
                # factorises and stores the factor in scaled form; solves run on it (engine.WideFactor)
                fast.generator_period = self.generator_period
                wide, fast = fast.stored_factor(), None
            elif not fast._fused_ok():
                fast = None
            else:
                # the drop-in class favours accuracy over the last 10 % of speed: exact generator
                # rows (float64-class results also for ill-conditioned problems, e.g. yerr = 0);
                # the batched throughput paths (BatchedLogLikelihood, bench.py) keep period 16
                fast.generator_period = self.generator_period
        self._fast = fast
        if fast is not None and self.STORE_MIN_ROWS <= self._size <= self.STORE_MAX_ROWS:
            # one long series: the time-parallel factorisation keeps the factor (scaled rows, 1 KB per
            # row, + ~1 ms at N = 1e6), as celerite2's compute does: log_likelihood is then ONE forward
            # sweep (1.5 instead of 5.5 ms at N = 1e6) and predict / apply_inverse / sample start at once
            import torch
            self._factor = fast.stored_factor()
            acc = fast._tp["acc"]
            info = fast.info
            logdet = torch.where(info != 0, torch.full_like(acc[:, 0], float("-inf")), acc[:, 0])
        elif fast is not None:
            _, logdet = fast.evaluate()
            info = fast.info
        elif wide is not None:
            self._factor = wide
            _, logdet = wide.reduce(with_quad=False)
            info = wide.info
        else:
            eng = DeviceBatch([co], self._t, diag=self._diag, device=self._device)
            self._factor = eng
            info = eng.factor(keep_W=True)
            _, logdet = eng.reduce(with_quad=False)
        failed = int(info[0].item())
        self._failed_row = failed
        if failed:
            if not quiet:
                raise LinAlgError(
                    f"failed to factorize or solve matrix (pivot {failed} of "
                    f"{self._size} is not positive)")
            self._log_det = -np.inf
            self._norm = np.inf
        else:
            self._log_det = float(logdet[0].item())
            self._norm = 0.5 * (self._log_det + self._size * LOG_2PI)

    # ---- celerite2 _do_* equivalents ----------------------------------------
    def _do_solve(self, y):
        Y = self._to_device(y).reshape(1, self._size, -1)
        Z = self._engine.apply_inverse(Y)
        return self._to_host(Z.reshape(y.shape))

    def _do_dot_tril(self, y):
        Y = self._to_device(y).reshape(1, self._size, -1)
        Z = self._engine.dot_tril(Y)
        return self._to_host(Z.reshape(y.shape))

    def _do_norm(self, y, resid=False):
        eng = self._engine
        Y = (self._resid_to_device(y) if resid else self._to_device(y)).reshape(1, self._size, 1)
        z = eng.solve_lower(Y).reshape(-1).contiguous()
        # sum z^2 / d on the library's fixed-shape reduction tree (gf_reduce_tile: the same deterministic
        # kernels the batched evaluations finish on)
        torch, lib, p = eng.torch, _lib.load(), _lib.ptr
        N = self._size
        with torch.cuda.device(eng.device):
            work = torch.empty((int(lib.gf_reduce_work(N)),), dtype=torch.float64, device=eng.device)
            acc = torch.empty((1, 3), dtype=torch.float64, device=eng.device)
            st = torch.cuda.current_stream(eng.device).cuda_stream
            _lib.check(lib.gf_reduce_tile(1, N, p(eng.d[0].contiguous()), p(z), p(work), p(acc), 1, st),
                       "gf_reduce_tile")
        return float(acc[0, 1].item())

    def _device_of(self):
        import torch
        return torch.device(self._device if self._device is not None
                            else f"cuda:{torch.cuda.current_device()}")

    # ---- public numerical API (reference gp.py:308-395) ------------------------
    def dot_tril(self, y, *, inplace=False):
        """``L D^{1/2} y`` with K = L D L^T (the mean is not applied)."""
        if _units.has_unit(y):
            y = self._flux_to_ppm(y)
        y_in = y
        y = self._process_input(y)
        out = self._do_dot_tril(y)
        if inplace and isinstance(y_in, np.ndarray) and y_in.dtype == np.float64:
            y_in[...] = out
            return y_in
        return out

    def log_likelihood(self, y, *, inplace=False):
        """Marginalised log-likelihood of ``y`` under the factorised model."""
        if _units.has_unit(y):
            y = self._flux_to_ppm(y)
        y = self._process_input(y, require_vector=True)
        if not np.isfinite(self._log_det):
            return -np.inf
        if self._fast is not None and self._factor is None:
            # fused build + factor + solve + reductions (time-parallel for long series)
            self._fast.set_y(y - self._mean_value)
            out, _ = self._fast.evaluate()
            return float(out[0].item())
        return -0.5 * self._do_norm(y, resid=True) - self._norm

    def apply_inverse(self, y, *, inplace=False):
        """``K^-1 y`` (the mean is not applied)."""
        if _units.has_unit(y):
            y = self._flux_to_ppm(y)
        y_in = y
        y = self._process_input(y)
        out = self._do_solve(y)
        if inplace and isinstance(y_in, np.ndarray) and y_in.dtype == np.float64:
            y_in[...] = out
            return y_in
        return out

    def sample(self, *, size=None, include_mean=True, return_quantity=False):
        """Prior draws (reference gp.py:372-395, including its mean-subtraction quirk:
        the across-realisation mean is removed for ``size`` draws, the time-mean for one)."""
        if self._t is None:
            raise RuntimeError("The process must be initialized with compute")
        # celerite2 draws from numpy's legacy global RNG (tests seed np.random.seed(42))
        if size is None:
            n = np.random.randn(self._size)
        else:
            n = np.random.randn(self._size, size)
        result = self._do_dot_tril(n).T
        if include_mean:
            result = result + self._mean_value
        result -= result.mean(axis=0 if result.ndim == 2 else None)
        if return_quantity:
            return self._ppm_to_flux(result)
        return result

    def sample_device(self, *, size=None, include_mean=True, rng="numpy", seed=None):
        """:meth:`sample` with the draws left on the GPU: a float64 tensor of shape (N,) or
        (size, N) [ppm], same arithmetic and mean-subtraction quirk -- for pipelines that continue on
        the device (:meth:`gadfly_amd.PowerSpectrum.from_flux`).  Extension; not in the reference.

        ``rng="numpy"`` (default) draws the normal vectors from numpy's legacy global RNG exactly as
        :meth:`sample` / celerite2 do (same draws for a given ``np.random.seed``), which costs a host
        ``randn`` and a PCIe copy (~0.4 s for 64 x 5e5).  ``rng="device"`` draws them on the GPU
        (``torch.randn``, optional ``seed``): statistically equivalent draws in milliseconds, but NOT the
        reference's random stream."""
        if self._t is None:
            raise RuntimeError("The process must be initialized with compute")
        if rng == "device":
            import torch
            dev = self._device_of()
            gen = None
            if seed is not None:
                gen = torch.Generator(device=dev)
                gen.manual_seed(int(seed))
            shape = (self._size,) if size is None else (self._size, int(size))
            nd = torch.randn(shape, dtype=torch.float64, device=dev, generator=gen)
            Z = self._engine.dot_tril(nd.reshape(1, self._size, -1))
            result = Z.reshape(shape).T.contiguous() if len(shape) == 2 else Z.reshape(-1)
            if include_mean:
                result = result + self._to_device(np.broadcast_to(self._mean_value, (self._size,)).copy())
            return result - (result.mean(dim=0) if result.ndim == 2 else result.mean())
        if rng != "numpy":
            raise ValueError("rng must be 'numpy' or 'device'")
        n = np.random.randn(self._size) if size is None else np.random.randn(self._size, size)
        Z = self._engine.dot_tril(self._to_device(n).reshape(1, self._size, -1))
        result = Z.reshape(n.shape).T.contiguous() if n.ndim == 2 else Z.reshape(-1)
        if include_mean:
            result = result + self._to_device(np.broadcast_to(self._mean_value, (self._size,)).copy())
        return result - (result.mean(dim=0) if result.ndim == 2 else result.mean())

    def condition(self, y, t=None, include_mean=True, kernel=None,
                  return_quantity=False):
        """Condition the GP on observations ``y`` (reference gp.py:206-239)."""
        if t is not None and (_units.is_time(t) or _units.has_unit(t)):
            t = self._time_to_freq(t)
        if _units.has_unit(y):
            y = self._flux_to_ppm(y)
        result = self.conditional_distribution(
            self, y, t=t, include_mean=include_mean, kernel=kernel)
        if return_quantity:
            return self._ppm_to_flux(result.mean)
        return result

    def predict(self, y, t=None, return_cov=False, return_var=False,
                include_mean=True, kernel=None, return_quantity=False):
        """Conditional mean (and variance / covariance) (reference gp.py:241-306)."""
        if _units.has_unit(y):
            y = self._flux_to_ppm(y)
        if _units.is_time(t) or _units.has_unit(t):
            t = self._time_to_freq(t)

        cond = self.condition(y, t=t, include_mean=include_mean, kernel=kernel)

        if return_var and return_quantity:
            return (self._ppm_to_flux(cond.mean),
                    self._ppm_to_flux(cond.variance, power=2))
        elif return_cov and return_quantity:
            return (self._ppm_to_flux(cond.mean),
                    self._ppm_to_flux(cond.covariance, power=2))
        elif return_quantity:
            return self._ppm_to_flux(cond.mean)
        elif return_var:
            return cond.mean, cond.variance
        elif return_cov:
            return cond.mean, cond.covariance
        return cond.mean
