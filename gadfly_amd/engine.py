"""
Device engine: B independent celerite problems resident in HBM.

This is the host-side owner of the device buffers that the C-ABI
(``include/gadfly_hip.h``) operates on.  It plays the role of celerite2's
``numpy.GaussianProcess._do_compute/_do_solve/_do_norm/_do_dot_tril`` (reached from
/root/reference/gadfly/gp.py:202, :350, :370, :327) for a *batch* of problems that
share N and the term structure (Jr, Jc): independent light curves (own t, y) or MCMC
walkers (shared t, y; own hyperparameters) -- SURVEY.md 8e.

HBM layout (float64, row-major, all torch tensors on one device):
    t, diag        (Bt, N)  Bt = 1 when shared
    a, d, z        (B, N)
    U, V, P, W     (B, N, ld)   ld = W rounded up to 16 (rows 128-byte aligned; pad cols
                                hold 0 for U/V/W and 1 for P)
    c              (B, Wd)      decay rates per column (prediction hops only)
"""
import functools
import math

import numpy as np

from . import _lib


def _on_device(method):
    """Run an engine method with the engine's own device current (allocations made inside, and
    per-device attributes the library applies, then belong to the device of the buffers and the
    stream -- whatever the caller's current device is)."""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self.torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapper


def _coeff_pack(coeffs_list):
    """list of B (ar, cr, ac, bc, cc, dc, shift) -> stacked float64 arrays."""
    Jr = len(coeffs_list[0][0])
    Jc = len(coeffs_list[0][2])
    B = len(coeffs_list)
    real = np.zeros((2, B, max(Jr, 1)))
    comp = np.zeros((4, B, max(Jc, 1)))
    diag_add = np.zeros(B)
    c = np.zeros((B, Jr + 2 * Jc))
    for b, (ar, cr, ac, bc, cc, dc, shift) in enumerate(coeffs_list):
        if len(ar) != Jr or len(ac) != Jc:
            raise ValueError("all problems of a batch must share the term structure")
        real[0, b, :Jr], real[1, b, :Jr] = ar, cr
        comp[0, b, :Jc], comp[1, b, :Jc] = ac, bc
        comp[2, b, :Jc], comp[3, b, :Jc] = cc, dc
        diag_add[b] = float(np.sum(ar) + np.sum(ac) + shift)
        c[b, :Jr] = cr
        c[b, Jr::2] = cc
        c[b, Jr + 1::2] = cc
    return Jr, Jc, real, comp, diag_add, c


def _complexify_pack(Jr, Jc, real, comp, diag_add, c):
    """The same kernels with every real term a e^{-c tau} written as the complex term (a, b = 0, c, d = 0):
    two state columns instead of one (the sine column is identically zero), in exchange for a term structure
    the fused wide sweep takes (k_factorw generates complex terms only).  Stacked arrays in, stacked arrays
    out (see _coeff_pack)."""
    B = comp.shape[1]
    J = Jr + Jc
    comp2 = np.zeros((4, B, max(J, 1)))
    comp2[0, :, :Jr], comp2[2, :, :Jr] = real[0, :, :Jr], real[1, :, :Jr]
    comp2[:, :, Jr:J] = comp[:, :, :Jc]
    c2 = np.zeros((B, 2 * J))
    c2[:, 0::2] = comp2[2, :, :J]
    c2[:, 1::2] = comp2[2, :, :J]
    return 0, J, np.zeros((2, B, 1)), comp2, diag_add, c2


class DeviceBatch:
    """B problems on one GPU.  Every method only *enqueues* work on the current stream;
    results stay on the device until the caller reads them."""

    def __init__(self, coeffs_list, t, diag=None, device=None):
        torch = _lib.require_device()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None
                                   else f"cuda:{torch.cuda.current_device()}")   # the rank's own GPU
        self.B = len(coeffs_list)
        self.Jr, self.Jc, real, comp, diag_add, c = _coeff_pack(coeffs_list)
        self.W = self.Jr + 2 * self.Jc
        self._dev = None
        if self.W < 1 or self.W > _lib.GF_MAX_WIDTH:
            raise ValueError(
                f"celerite width {self.W} unsupported (1..{_lib.GF_MAX_WIDTH})")
        self.ld = self.lib.gf_leading_dim(self.W)
        f64 = dict(dtype=torch.float64, device=self.device)

        def dev(x):
            if isinstance(x, torch.Tensor):
                return x.to(**f64).contiguous()
            return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)

        self._dev = dev
        self.t = dev(t)
        if self.t.ndim == 1:
            self.t = self.t[None, :]
        self.N = int(self.t.shape[1])
        if self.t.shape[0] not in (1, self.B):
            raise ValueError("dimension mismatch")
        self.diag = None
        if diag is not None:
            self.diag = dev(diag)
            if self.diag.ndim == 1:
                self.diag = self.diag[None, :]
            if self.diag.shape[1] != self.N or self.diag.shape[0] not in (1, self.B):
                raise ValueError("dimension mismatch")
        self._real = dev(real)
        self._comp = dev(comp)
        self._diag_add = dev(diag_add)
        self.c = dev(c)
        B, N, ld = self.B, self.N, self.ld
        self.a = torch.empty((B, N), **f64)
        self.U = torch.empty((B, N, ld), **f64)
        self.V = torch.empty((B, N, ld), **f64)
        self.P = torch.empty((B, N, ld), **f64)
        self.d = None
        self.Wm = None
        self.z = None
        self.info = torch.zeros((B,), dtype=torch.int32, device=self.device)
        # optional HIP-event timing of the factor launch (bench.py roofline leg)
        self.time_factor = False
        self.factor_events = []
        self._build()

    # -- fresh hyperparameters (MCMC walkers): O(J) upload + matrix rebuild ---
    def pack_coefficients(self, coeffs_list):
        """Upload a new set of B coefficient tuples; returns an opaque device pack that
        :meth:`use_coefficients` switches to without further host work."""
        Jr, Jc, real, comp, diag_add, c = _coeff_pack(coeffs_list)
        if (Jr, Jc) != (self.Jr, self.Jc) or len(coeffs_list) != self.B:
            raise ValueError("coefficient pack does not match the batch structure")
        dev = self._dev
        return dev(real), dev(comp), dev(diag_add), dev(c)

    def use_coefficients(self, pack, rebuild=True):
        self._real, self._comp, self._diag_add, self.c = pack
        if rebuild:
            self._build()

    def set_coefficients(self, coeffs_list):
        self.use_coefficients(self.pack_coefficients(coeffs_list))

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _bs(x):
        """batch stride in elements (0 = shared)."""
        return 0 if x.shape[0] == 1 else x.stride(0)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    @_on_device
    def _build(self):
        p = _lib.ptr
        st = self.lib.gf_build_matrices(
            self.B, self.N, 0, self.Jr, self.Jc, self.ld,
            p(self._real[0]), p(self._real[1]),
            p(self._comp[0]), p(self._comp[1]), p(self._comp[2]), p(self._comp[3]),
            p(self._diag_add),
            p(self.t), self._bs(self.t),
            p(self.diag), 0 if self.diag is None else self._bs(self.diag),
            p(self.a), p(self.U), p(self.V), p(self.P), self._stream())
        _lib.check(st, "gf_build_matrices")

    @_on_device
    def matrices_at(self, tstar):
        """U*, V* at new times (M,) or (B, M) with zero diagonal (prediction)."""
        torch = self.torch
        ts = tstar if isinstance(tstar, torch.Tensor) else torch.as_tensor(
            np.ascontiguousarray(tstar, dtype=np.float64))
        ts = ts.to(dtype=torch.float64, device=self.device).contiguous()
        if ts.ndim == 1:
            ts = ts[None, :]
        M = int(ts.shape[1])
        Us = torch.empty((self.B, M, self.ld), dtype=torch.float64, device=self.device)
        Vs = torch.empty_like(Us)
        p = _lib.ptr
        st = self.lib.gf_build_matrices(
            self.B, M, 0, self.Jr, self.Jc, self.ld,
            p(self._real[0]), p(self._real[1]),
            p(self._comp[0]), p(self._comp[1]), p(self._comp[2]), p(self._comp[3]),
            p(self._diag_add), p(ts), self._bs(ts), None, 0,
            None, p(Us), p(Vs), None, self._stream())
        _lib.check(st, "gf_build_matrices")
        return ts, Us, Vs

    # -- factor / log-likelihood ------------------------------------------
    @_on_device
    def factor(self, y=None, keep_W=True):
        """LDL^T factor; with ``y`` (resid, (N,) | (1,N) | (B,N) device tensor) also the
        forward solve z = L^-1 y in the same sweep."""
        torch = self.torch
        B, N, ld = self.B, self.N, self.ld
        f64 = dict(dtype=torch.float64, device=self.device)
        if self.d is None:
            self.d = torch.empty((B, N), **f64)
        if keep_W and self.Wm is None:
            self.Wm = torch.empty((B, N, ld), **f64)
        yb = 0
        if y is not None:
            if y.ndim == 1:
                y = y[None, :]
            if y.shape[1] != N or y.shape[0] not in (1, B):
                raise ValueError("dimension mismatch")
            y = y.contiguous()
            yb = self._bs(y)
            if self.z is None:
                self.z = torch.empty((B, N), **f64)
        p = _lib.ptr
        if self.time_factor:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self.info.zero_()
        st = self.lib.gf_factor(
            B, N, 0, self.W, ld, p(self.a), p(self.U), p(self.V), p(self.P),
            p(y), yb, p(self.d), p(self.Wm) if keep_W else None,
            p(self.z) if y is not None else None, None, None, p(self.info),
            self._stream())
        if self.time_factor:
            e1.record()
            self.factor_events.append((e0, e1))
        _lib.check(st, "gf_factor")
        return self.info

    @_on_device
    def reduce(self, with_quad):
        """(loglike (B,), logdet (B,)) device tensors from d (and z when with_quad)."""
        torch = self.torch
        B, N = self.B, self.N
        f64 = dict(dtype=torch.float64, device=self.device)
        work = torch.empty((B * int(self.lib.gf_reduce_work(N)),), **f64)
        acc = torch.empty((B, 3), **f64)
        out = torch.empty((B,), **f64)
        logdet = torch.empty((B,), **f64)
        p = _lib.ptr
        st = self.lib.gf_reduce_tile(B, N, p(self.d), p(self.z) if with_quad else None,
                                     p(work), p(acc), 1, self._stream())
        _lib.check(st, "gf_reduce_tile")
        st = self.lib.gf_loglike_finish(B, N, p(acc), p(self.info), p(out), p(logdet),
                                        self._stream())
        _lib.check(st, "gf_loglike_finish")
        return out, logdet

    def log_likelihood(self, y, keep_W=False):
        """One full evaluation per problem: factor + forward solve + reductions."""
        self.factor(y=y, keep_W=keep_W)
        out, _ = self.reduce(with_quad=True)
        return out

    # -- sweeps with a stored factor ---------------------------------------
    @_on_device
    def _sweep(self, mode, Y, scale=None, out=None):
        """Y: (B, N, R) contiguous device tensor."""
        torch = self.torch
        if self.Wm is None:
            raise RuntimeError("factor(keep_W=True) must run before solves")
        B, N, R = Y.shape
        Z = out if out is not None else torch.empty_like(Y)
        p = _lib.ptr
        st = self.lib.gf_solve(mode, B, N, self.W, self.ld, R, p(self.U), p(self.Wm),
                               p(self.P), p(scale), p(Y), p(Z), self._stream())
        _lib.check(st, "gf_solve")
        return Z

    def solve_lower(self, Y):
        return self._sweep(_lib.GF_SOLVE_LOWER, Y)

    def solve_upper(self, Y, scale=None):
        return self._sweep(_lib.GF_SOLVE_UPPER, Y, scale=scale)

    def apply_inverse(self, Y):
        """K^-1 Y = L^-T D^-1 L^-1 Y (division by d fused into the upper sweep)."""
        Z = self._sweep(_lib.GF_SOLVE_LOWER, Y)
        return self._sweep(_lib.GF_SOLVE_UPPER, Z, scale=self.d, out=Z)

    def dot_tril(self, Y):
        """L D^{1/2} Y (sqrt(d) fused into the sweep)."""
        return self._sweep(_lib.GF_MATMUL_LOWER, Y, scale=self.d)

    @_on_device
    def predict_at(self, alpha, ts, Us, Vs, other=None):
        """Conditional mean at new times; ``other`` supplies (c, U, V, P) of a different
        kernel evaluated at the observed times (celerite2's ``kernel=`` argument)."""
        torch = self.torch
        src = other if other is not None else self
        B, N = self.B, self.N
        M = int(ts.shape[1])
        f64 = dict(dtype=torch.float64, device=self.device)
        work = torch.empty((int(self.lib.gf_general_matmul_work(B, M, N, src.W)),), **f64)
        mu = torch.empty((B, M), **f64)
        # number of observed rows at or before each query time (device searchsorted)
        qidx = torch.searchsorted(self.t.expand(B, N).contiguous(),
                                  ts.expand(B, M).contiguous(), right=True).contiguous()
        p = _lib.ptr
        st = self.lib.gf_general_matmul(
            B, M, N, src.W, src.ld, p(src.c),
            p(ts), self._bs(ts), p(Us), p(Vs),
            p(self.t), self._bs(self.t), p(src.U), p(src.V), p(src.P), p(alpha), p(qidx),
            p(work), p(mu), self._stream())
        _lib.check(st, "gf_general_matmul")
        return mu

    @property
    def nbytes_algorithmic_loglike(self):
        """ALGORITHMIC bytes of one log-likelihood evaluation, SURVEY.md 8d:
        8 N (3 W + 4) per problem (celerite2-equivalent data flow, each array once)."""
        return 8 * self.N * (3 * self.W + 4) * self.B


LOG_2PI = math.log(2.0 * math.pi)
#: largest phase |d t| the in-kernel sincos takes (FM_SINCOS_RANGE in csrc/fastmath.h)
SINCOS_RANGE = 1.0e12




class ScaledFactor:
    """
    The LDL^T factor of B problems stored in block-scaled form (rows u~, w~ = r/d, pivots d,
    reset spans de, per-chunk closed-loop transitions Phi) -- what `apply_inverse`, `dot_tril`,
    `predict` and `sample` need (/root/reference/gadfly/gp.py:370, :327, :232, :391).  All sweeps are
    chunk-parallel in time: local pass, linear combine of the chunk states, final pass
    (gf_chunk_linear / gf_chunk_linear_combine).  Same method surface as :class:`DeviceBatch`.
    """

    RMAX = 64                           # right-hand sides per sweep (one wave of lanes)

    def __init__(self, owner, chunk_len, nch):
        self.owner = owner
        self.torch = owner.torch
        self.lib = owner.lib
        self.device = owner.device
        self.B, self.N, self.W = owner.B, owner.N, owner.W
        self.chunk_len, self.nch = chunk_len, nch
        self.block = int(owner._pack[5])        # scaling block of the stored rows: reset rows at its multiples
        w = owner._tp
        self.Ut, self.Wt, self.de, self.Phi = w["Ut"], w["Wt"], w["de"], w["PhiT"]
        # own copy of the pivots: the shared buffer holds the nominal pass' values while a later
        # evaluation is in flight
        self.d = w["d"][:self.B * self.N].view(self.B, self.N).clone()
        # the true chunk transitions Phi_c are computed at the first solve (_ensure_transitions): from the rows r
        # of the final pass, still in the owner's buffer unless it has run another evaluation since
        self._phi_ready = nch <= 1
        self._r_rows, self._r_generation = w["r"], getattr(owner, "_tp_generation", 0)
        self.c = owner._pack[3]
        self.t = owner.t
        self.info = owner.info
        self._v1 = None
        self._D, self._D_key = None, None
        self._Psi = None
        self._PhiT = self._PsiT = None      # (transposes of Phi / Psi for the backward combine: with _Psi)

    @_on_device
    def _ensure_transitions(self):
        """Phi_c of the TRUE factor's chunks: the transition sweep on the stored rows (u~, r = w~ d, d, reset
        spans), once, at the first solve that chains chunk states."""
        if self._phi_ready:
            return
        torch, lib, p, o = self.torch, self.lib, _lib.ptr, self.owner
        B, N = self.B, self.N
        w = o._tp
        r = self._r_rows
        if getattr(o, "_tp_generation", 0) != self._r_generation:
            # the owner has swept again since: its row buffer holds another pass' rows -- r = w~ d from the factor
            r = torch.empty((B * N + 8, 64), dtype=torch.float64, device=self.device)
            r[B * N:].zero_()
            torch.mul(self.Wt.view(B * N, 64), self.d.reshape(B * N, 1), out=r[:B * N])
        f64 = dict(dtype=torch.float64, device=self.device)
        G, m = torch.empty((B * self.nch, 4096), **f64), torch.empty((B * self.nch, 64), **f64)   # (by-products)
        dd = torch.empty((B * N + 8,), **f64)
        dd[:B * N] = self.d.reshape(-1)
        dd[B * N:] = 1.0
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.gf_chunk_transition(B, N, self.chunk_len, self.nch, 0, self.nch, o.Jr, o.Jc,
                                     int(o.sweep_variant), p(self.c), p(self.de), p(dd), p(dd),
                                     p(r), p(self.Ut), p(self.Phi), p(G), p(m), st)
        _lib.check(rc, "gf_chunk_transition")
        self._phi_ready = True
        self._r_rows = None

    @_on_device
    def _sweep(self, mode, Y, scale):
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        B, N, R = Y.shape
        if mode != _lib.GF_MATMUL_LOWER:
            self._ensure_transitions()
        if R > self.RMAX:               # tile the right-hand sides
            return torch.cat([self._sweep(mode, Y[:, :, r0:r0 + self.RMAX].contiguous(), scale)
                              for r0 in range(0, R, self.RMAX)], dim=2)
        Y = Y.contiguous()
        Z = None                        # (allocated behind the local pass' launch: the device starts earlier)
        f64 = dict(dtype=torch.float64, device=self.device)
        chunk_len, nch = self.chunk_len, self.nch
        if mode == _lib.GF_MATMUL_LOWER and nch > 1:
            # dot_tril's chunk transitions are diagonal products of the reset decays, computed for ANY
            # chunking: cut the series finer than the factorisation did (two waves per SIMD; the chunks
            # of the factor leave half the SIMDs without a wave when there is one RHS tile)
            chunk_len, nch = self._mm_chunking(R)
        # (every local pass starts from zero by itself, but only dot_tril's diagonal combine keeps the state
        # rows apart: the solves' combine multiplies the padding rows >= W, which no pass writes, by zeros)
        lazy = nch > 1 and mode == _lib.GF_MATMUL_LOWER
        F = (torch.empty if lazy else torch.zeros)((B * nch, 64 * R), **f64)
        st = torch.cuda.current_stream(self.device).cuda_stream
        args = (mode, B, N, chunk_len, nch, self.W, R)
        rows = (p(self.c), p(self.Ut), p(self.Wt), p(self.d), p(self.de))
        if nch > 1:
            rc = lib.gf_chunk_linear(*args, int(scale), 0, *rows, p(Y), p(Y), p(F), st)    # (writes no rows)
            _lib.check(rc, "gf_chunk_linear")
            Z = torch.empty_like(Y)
            fresh_D = False
            if mode == _lib.GF_MATMUL_LOWER and (self._D is None or self._D_key != (chunk_len, nch)):
                # the chunks' diagonal transitions belong to the factor and this chunking: formed once
                self._D, self._D_key, fresh_D = torch.empty((B * nch, 64), **f64), (chunk_len, nch), True
            seg = self._segments() if mode != _lib.GF_MATMUL_LOWER else None
            if seg is not None:         # long series: two-level combine on the composed transitions
                seg_len, Psi = seg
                V = torch.empty((B * (-(-nch // seg_len)), 64 * R), **f64)
                rc = lib.gf_chunk_linear_combine_seg(mode, B, nch, seg_len, R, p(self.Phi), p(Psi),
                                                     p(self._PhiT), p(self._PsiT), p(F), p(V), st)
                _lib.check(rc, "gf_chunk_linear_combine_seg")
            elif mode == _lib.GF_MATMUL_LOWER and not fresh_D:
                _lib.check(lib.gf_chunk_diag_scan(B, nch, 64, R, p(self._D), p(F), st), "gf_chunk_diag_scan")
            else:
                rc = lib.gf_chunk_linear_combine(
                    *args, p(self.c), p(self.de),
                    None if mode == _lib.GF_MATMUL_LOWER else p(self.Phi),
                    p(self._D) if mode == _lib.GF_MATMUL_LOWER else None, p(F), st)
                _lib.check(rc, "gf_chunk_linear_combine")
        if Z is None:
            Z = torch.empty_like(Y)
        rc = lib.gf_chunk_linear(*args, int(scale), 1, *rows, p(Y), p(Z), p(F), st)
        _lib.check(rc, "gf_chunk_linear")
        return Z

    SEG_MIN_CHUNKS = 32                 # below this the plain sequential combine is as fast

    def _segments(self):
        """(seg_len, Psi) of the two-level combine of the solves, or None for short chains.  The
        composed segment transitions are part of the factor: built on first use, then kept.
        Sequential depth 2 seg_len + nch / seg_len chunks -> seg_len = sqrt(nch / 2)."""
        if self.nch < self.SEG_MIN_CHUNKS:
            return None
        self._ensure_transitions()
        if self._Psi is None:
            seg_len = max(2, int(round(math.sqrt(self.nch / 2.0))))
            nseg = -(-self.nch // seg_len)
            Psi = self.torch.empty((self.B * nseg, 4096), dtype=self.torch.float64, device=self.device)
            # (and the transposes of both, for the backward solve's combine: its loads then run along the lanes)
            self._PhiT = self.torch.empty_like(self.Phi)
            self._PsiT = self.torch.empty_like(Psi)
            st = self.torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.gf_chunk_segment_transitions(self.B, self.nch, seg_len, _lib.ptr(self.Phi),
                                                       _lib.ptr(Psi), _lib.ptr(self._PhiT), _lib.ptr(self._PsiT), st)
            _lib.check(rc, "gf_chunk_segment_transitions")
            self._Psi = (seg_len, Psi)
        return self._Psi

    def _mm_chunking(self, R):
        """Chunks for the dot_tril sweeps: ~2048 waves (two per SIMD) over B problems and the RHS
        tiles, at least 128 rows each, on multiples of the scaling block (every chunk starts on a reset row;
        rounding to 64 left 5 % of the workgroup slots empty at cfg5's size)."""
        # (R >= 16 runs on the matrix pipe, two-wave workgroups: k_mmR_mfma)
        tiles = -(-R // 64) if R > 1 else 1
        want = max(1, (1024 if R >= 16 else 2048) // (self.B * tiles))
        q = max(self.block, 16)
        chunk_len = max(128, -(-self.N // want))
        chunk_len = (chunk_len + q - 1) // q * q
        return chunk_len, -(-self.N // chunk_len)

    def solve_lower(self, Y):
        return self._sweep(_lib.GF_SOLVE_LOWER, Y, 0)

    def solve_upper(self, Y, scale=None):
        return self._sweep(_lib.GF_SOLVE_UPPER, Y, 0 if scale is None else 1)

    def apply_inverse(self, Y):
        return self._sweep(_lib.GF_SOLVE_UPPER, self._sweep(_lib.GF_SOLVE_LOWER, Y, 0), 1)

    def dot_tril(self, Y):
        return self._sweep(_lib.GF_MATMUL_LOWER, Y, 1)

    # prediction at new times: short sequential hop kernels on unscaled generator rows
    def _unscaled(self):
        if self._v1 is None:
            o = self.owner
            if o._coeffs_list is None:
                raise RuntimeError("prediction at new times needs the construction-time "
                                   "coefficients (use_coefficients() replaced them)")
            self._v1 = DeviceBatch(o._coeffs_list, o.t, diag=o.diag, device=self.device)
        return self._v1

    def matrices_at(self, tstar):
        return self._unscaled().matrices_at(tstar)

    def predict_at(self, alpha, ts, Us, Vs, other=None):
        return self._unscaled().predict_at(alpha, ts, Us, Vs, other=other)


class WideFactor:
    """
    The LDL^T factor of B problems with a WIDE kernel (64 <= W <= 176, complex terms), stored in
    block-scaled form by the fused wide sweep (k_factorw: rows u~, w~ = r/d, pivots d, reset spans)
    and swept with the general-width kernels: in scaled coordinates the row-to-row propagator is 1
    (exp(-c de) at a reset row, ``gf_scaled_propagator``), so ``gf_solve`` runs on these rows
    unchanged.  One sequential pass builds it (no chunk-parallel machinery at these widths yet);
    same method surface as :class:`DeviceBatch` / :class:`ScaledFactor`.
    """

    def __init__(self, owner, time_parallel=None, chunk_len=None):
        torch = owner.torch
        self.owner, self.torch, self.lib, self.device = owner, torch, owner.lib, owner.device
        self.B, self.N, self.W = owner.B, owner.N, owner.W
        lib, p = self.lib, _lib.ptr
        B, N = self.B, self.N
        self.ld = int(lib.gf_fused_row_stride(owner.Jr, owner.Jc))
        f64 = dict(dtype=torch.float64, device=self.device)
        # (eight spare rows: the transition sweep fetches its rows ahead, unconditionally -- what it reads
        # there is never used, so nothing is zeroed: every row and column that IS used is written by the sweep)
        self.Ut = torch.empty((B * N + 8, self.ld), **f64)[:B * N].view(B, N, self.ld)
        self.Wt = torch.empty((B, N, self.ld), **f64)
        self.de = torch.empty((B * N + 8,), **f64)[:B * N].view(B, N)
        self.d = torch.empty((B * N + 8,), **f64)[:B * N].view(B, N)
        self.z = torch.empty((B * N + 8,), **f64)[:B * N].view(B, N)
        real, comp, diag_add, c, cmax, block, _ = owner._pack
        self.c, self.t = c, owner.t
        self.info = torch.zeros((B,), dtype=torch.int32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        # (the chunk-parallel SWEEPS on the stored factor are for one series; a batch is swept whole)
        self.time_parallel = ((owner._wide_tp_ok() and B == 1) if time_parallel is None
                              else bool(time_parallel))
        self.chunk_len, self.nch = N, 1
        self._tp_bufs = None
        if self.time_parallel:
            # exact time-parallel factorisation: chunks swept concurrently, stitched by the LFT combine
            _, _, self.chunk_len, self.nch = owner._tp_run_wide(
                chunk_len, stores=(self.Ut, self.Wt, self.de), d=self.d, z=self.z, info=self.info)
            self._tp_bufs = owner._wide_tp_bufs          # r rows of the TRUE factor, h / Phi scratch
        else:
            S = torch.zeros((B, int(lib.gf_fused_state_size(owner.Jr, owner.Jc))), **f64)
            bs = owner._bs
            rc = lib.gf_chunk_sweep(
                B, N, N, 1, 0, 1, owner.Jr, owner.Jc, block, int(owner.generator_period), _lib.GF_SWEEP_AUTO,
                p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]),
                p(diag_add), p(cmax), p(owner.t), bs(owner.t), p(owner.diag),
                0 if owner.diag is None else bs(owner.diag), p(owner.y), bs(owner.y),
                p(self.d), p(self.z), None, p(self.Ut), p(self.Wt), p(self.de),
                p(S), None, p(self.info), st)
            _lib.check(rc, "gf_chunk_sweep")
        self.P = torch.empty((B, N, self.ld), **f64)
        rc = lib.gf_scaled_propagator(B, N, self.W, self.ld, p(c), p(self.de), p(self.P), st)
        _lib.check(rc, "gf_scaled_propagator")
        self._v1 = None
        self._Phi = None
        self._chain = None

    def reduce(self, with_quad):
        """(loglike (B,), logdet (B,)) of the pass that built the factor (z = L^-1 of the owner's y)."""
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        B, N = self.B, self.N
        f64 = dict(dtype=torch.float64, device=self.device)
        work = torch.empty((B * int(lib.gf_reduce_work(N)),), **f64)
        acc = torch.empty((B, 3), **f64)
        out, logdet = torch.empty((B,), **f64), torch.empty((B,), **f64)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(lib.gf_reduce_tile(B, N, p(self.d), p(self.z) if with_quad else None, p(work),
                                      p(acc), 1, st), "gf_reduce_tile")
        _lib.check(lib.gf_loglike_finish(B, N, p(acc), p(self.info), p(out), p(logdet), st),
                   "gf_loglike_finish")
        return out, logdet

    def _true_transitions(self):
        """Closed-loop transitions Phi_c of the TRUE factor's chunks (k_phiw on the rows the final pass
        stored), computed on first use: what the chunk-parallel solves chain their states with."""
        if self._Phi is None:
            torch, lib, p = self.torch, self.lib, _lib.ptr
            w = self._tp_bufs
            st = torch.cuda.current_stream(self.device).cuda_stream
            rc = lib.gf_chunk_transition_wide(1, self.N, self.chunk_len, self.nch, 0, self.nch, self.owner.Jc, p(self.c),
                                              p(self.de), p(self.d), p(w["r"]), p(self.Ut), p(w["h"]),
                                              p(w["Phi"]), st)
            _lib.check(rc, "gf_chunk_transition_wide")
            nS = w["Phi"].shape[1]
            P3 = w["Phi"].view(self.nch, self.ld, nS // self.ld)
            self._Phi = P3[:, :self.W, :self.W].transpose(1, 2).contiguous()      # Phi_c(i, j)
            self._tp_bufs = None                        # the row scratch is no longer needed
        return self._Phi

    def _gemm(self, batch, ta, tb, M, N, K, A, lda, sA, Bm, ldb, sB, D, ldd, sD, C, ldc, sC):
        """C_b = D_b + op(A_b) op(B_b) on the library's FP64-MFMA GEMM tiles (gf_bgemm); operands are
        (views of) device tensors, leading dimensions and batch strides in elements."""
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.gf_bgemm(batch, ta, tb, M, N, K, _lib.ptr(A), lda, sA, _lib.ptr(Bm), ldb, sB,
                               _lib.ptr(D), ldd, sD, _lib.ptr(C), ldc, sC, st)
        _lib.check(rc, "gf_bgemm")

    def _chain_segments(self, up):
        """The chunks' closed-loop transitions in sweep order (Phi_c ascending for the forward solve,
        Phi_c^T descending for the backward one), cut into segments of L chunks (identity maps pad the
        last one), with every segment's composed transition: (L, nseg, A (nseg, L, W, W), Psi (nseg, W, W)).
        Built on first use, kept with the factor."""
        if self._chain is None:
            self._chain = {}
        if up not in self._chain:
            torch = self.torch
            Phi = self._true_transitions()
            A = Phi.transpose(1, 2).flip(0) if up else Phi
            nch, W = self.nch, self.W
            L = max(2, int(round(math.sqrt(nch / 2.0))))
            nseg = -(-nch // L)
            pad = nseg * L - nch
            if pad:
                eye = torch.eye(W, dtype=A.dtype, device=A.device).expand(pad, W, W)
                A = torch.cat([A, eye])
            A = A.contiguous().view(nseg, L, W, W)
            WW = W * W
            ping = [torch.empty((nseg, W, W), dtype=A.dtype, device=A.device) for _ in range(2)]
            Psi = A[:, 0]
            sP = L * WW
            for j in range(1, L):                       # Psi <- A_j Psi
                out = ping[j & 1]
                self._gemm(nseg, 0, 0, W, W, W, A[:, j], W, L * WW, Psi, W, sP, None, 0, 0, out, W, WW)
                Psi, sP = out, WW
            self._chain[up] = (L, nseg, A, Psi.contiguous())
        return self._chain[up]

    def _chain_states(self, up, loc):
        """True start state of every chunk from the chunks' local end states: x_{s+1} = loc_s + A_s x_s
        along the sweep, for one right-hand side ((nch, W)) or R of them ((nch, W, R)).  Two levels --
        every segment from a zero start (L batched steps), the segments chained through their composed
        transitions (nseg steps), every segment again from its true start (L batched steps) -- instead
        of nch dependent products; every step is one gf_bgemm launch."""
        torch = self.torch
        L, nseg, A, Psi = self._chain_segments(up)
        nch, W = self.nch, self.W
        vec = loc.ndim == 2                             # (nch, W), or (nch, W, R) for R right-hand sides
        if vec:
            loc = loc.unsqueeze(-1)
        R = loc.shape[-1]
        if up:
            loc = loc.flip(0)
        pad = nseg * L - nch
        if pad:
            loc = torch.cat([loc, torch.zeros((pad, W, R), dtype=loc.dtype, device=loc.device)])
        loc = loc.contiguous().view(nseg, L, W, R)
        WW, WR = W * W, W * R
        kw = dict(dtype=loc.dtype, device=loc.device)
        E = torch.zeros((nseg, L + 1, W, R), **kw)      # E[:, j]: state after j chunks from a zero start
        for j in range(L):
            self._gemm(nseg, 0, 0, W, R, W, A[:, j], W, L * WW, E[:, j], R, (L + 1) * WR,
                       loc[:, j], R, L * WR, E[:, j + 1], R, (L + 1) * WR)
        out = torch.zeros((nseg, L, W, R), **kw)        # out[s, j]: true start state of chunk s L + j
        for s in range(nseg - 1):                       # the segments' true start states
            self._gemm(1, 0, 0, W, R, W, Psi[s], W, 0, out[s, 0], R, 0, E[s, L], R, 0, out[s + 1, 0], R, 0)
        for j in range(L - 1):                          # every chunk's true start state
            self._gemm(nseg, 0, 0, W, R, W, A[:, j], W, L * WW, out[:, j], R, L * WR,
                       loc[:, j], R, L * WR, out[:, j + 1], R, L * WR)
        out = out.view(nseg * L, W, R)[:nch]
        if up:
            out = out.flip(0)
        return out.squeeze(-1) if vec else out

    def _mm_chunking(self):
        chunk_len = max(128, -(-self.N // 2048))
        chunk_len = (chunk_len + 63) // 64 * 64
        return chunk_len, -(-self.N // chunk_len)

    def _sweep_chunked(self, mode, Y, scale, Z):
        """ONE right-hand side, chunk-parallel (gf_solve_chunk): local pass, combine of the chunk states,
        final pass.  dot_tril's transitions are diagonal (any chunking, scanned on the device); the
        solves chain the chunks' closed-loop transitions (W x W mat-vecs, sequential over the chunks)."""
        torch, lib, p = self.torch, self.lib, _lib.ptr
        N, W, ld = self.N, self.W, self.ld
        f64 = dict(dtype=torch.float64, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        mm = mode == _lib.GF_MATMUL_LOWER
        L, nch = self._mm_chunking() if mm else (self.chunk_len, self.nch)
        F = torch.zeros((nch, ld), **f64)
        rows = (p(self.Ut), p(self.Wt), p(self.P), p(scale), p(Y), p(Z))
        _lib.check(lib.gf_solve_chunk(mode, 1, N, L, nch, W, ld, *rows, p(F), 0, st), "gf_solve_chunk")
        if mm:
            span = torch.zeros((nch * L,), **f64)
            span[:N] = self.de.reshape(-1).clamp(min=0.0)           # reset spans inside each chunk
            D = torch.ones((nch, ld), **f64)
            D[:, :W] = torch.exp(-span.view(nch, L).sum(dim=1)[:, None] * self.c.reshape(1, W))
            _lib.check(lib.gf_chunk_diag_scan(1, nch, ld, 1, p(D), p(F), st), "gf_chunk_diag_scan")
        else:
            F[:, :W] = self._chain_states(mode == _lib.GF_SOLVE_UPPER, F[:, :W])
        _lib.check(lib.gf_solve_chunk(mode, 1, N, L, nch, W, ld, *rows, p(F), 1, st), "gf_solve_chunk")
        return Z

    @_on_device
    def _sweep(self, mode, Y, scale=None, out=None):
        torch = self.torch
        B, N, R = Y.shape
        Y = Y.contiguous()
        Z = out if out is not None else torch.empty_like(Y)
        p = _lib.ptr
        st = torch.cuda.current_stream(self.device).cuda_stream
        if R == 1 and B == 1 and self.time_parallel and self.nch > 1:
            return self._sweep_chunked(mode, Y, scale, Z)
        if (B == 1 and self.time_parallel and self.nch > 1 and mode != _lib.GF_MATMUL_LOWER
                and self.nch <= 65535):
            # several right-hand sides (conditional variance / covariance): the same three steps with
            # W x R chunk states (k_solve_rhs in chunk mode, the chain as batched GEMMs)
            lib = self.lib
            N, W, ld, L, nch = self.N, self.W, self.ld, self.chunk_len, self.nch
            F = torch.zeros((nch, ld, R), dtype=torch.float64, device=self.device)
            rows = (p(self.Ut), p(self.Wt), p(self.P), p(scale), p(Y), p(Z))
            _lib.check(lib.gf_solve_chunk_rhs(mode, 1, N, L, nch, W, ld, R, *rows, p(F), 0, st),
                       "gf_solve_chunk_rhs")
            F[:, :W] = self._chain_states(mode == _lib.GF_SOLVE_UPPER, F[:, :W])
            _lib.check(lib.gf_solve_chunk_rhs(mode, 1, N, L, nch, W, ld, R, *rows, p(F), 1, st),
                       "gf_solve_chunk_rhs")
            return Z
        rc = self.lib.gf_solve(mode, B, N, self.W, self.ld, R, p(self.Ut), p(self.Wt), p(self.P),
                               p(scale), p(Y), p(Z), st)
        _lib.check(rc, "gf_solve")
        return Z

    def solve_lower(self, Y):
        return self._sweep(_lib.GF_SOLVE_LOWER, Y)

    def solve_upper(self, Y, scale=None):
        return self._sweep(_lib.GF_SOLVE_UPPER, Y, scale=scale)

    def apply_inverse(self, Y):
        Z = self._sweep(_lib.GF_SOLVE_LOWER, Y)
        return self._sweep(_lib.GF_SOLVE_UPPER, Z, scale=self.d, out=Z)

    def dot_tril(self, Y):
        return self._sweep(_lib.GF_MATMUL_LOWER, Y, scale=self.d)

    # prediction at new times: short hop kernels on unscaled generator rows
    def _unscaled(self):
        if self._v1 is None:
            o = self.owner
            if o._coeffs_list is None:
                raise RuntimeError("prediction at new times needs the construction-time "
                                   "coefficients (use_coefficients() replaced them)")
            self._v1 = DeviceBatch(o._coeffs_list, o.t, diag=o.diag, device=self.device)
        return self._v1

    def matrices_at(self, tstar):
        return self._unscaled().matrices_at(tstar)

    def predict_at(self, alpha, ts, Us, Vs, other=None):
        return self._unscaled().predict_at(alpha, ts, Us, Vs, other=other)


class StreamingBatch:
    """
    B independent log-likelihood evaluations streamed through fixed-size tile buffers.

    The time axis is cut into tiles of ``tile_rows`` rows.  For each tile the generator rows
    (U, V, P: B x tile_rows x ld) are built, the factor + forward-solve sweep advances every
    problem by one tile (recurrence state handed over in HBM: S (W x W) and F (W) per
    problem), and the tile's sum log d / sum z^2/d are accumulated.  U/V/P therefore never
    exist for the whole series: a walker costs ~3 * 8 * tile_rows * ld bytes of HBM instead
    of 3 * 8 * N * ld, which is what lets hundreds of N = 1e6 evaluations share one GPU.
    Two tile buffers and two streams overlap the build of tile k+1 with the sweep of tile k.
    """

    def __init__(self, coeffs_list, t, y, diag=None, tile_rows=8192, device=None,
                 force_v1=False, overlap_build=False, allow_fused=True, axis_stats=None):
        torch = _lib.require_device()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None
                                   else f"cuda:{torch.cuda.current_device()}")   # the rank's own GPU
        self.B = len(coeffs_list)
        self.Jr, self.Jc, real, comp, diag_add, c = _coeff_pack(coeffs_list)
        # A WIDE kernel with real terms (an overdamped SHO term, Q < 1/2, next to 30+ others) rides on the fused
        # wide sweep with its real terms written as degenerate complex ones (two columns each, the second all
        # zeros) rather than on the materialised-row kernels of round 1.
        self._struct0 = (self.Jr, self.Jc)
        self._complexified = bool(
            allow_fused and not force_v1 and self.Jr > 0 and self.Jr + 2 * self.Jc > 63
            and self.lib.gf_fused_supported(0, self.Jr + self.Jc))
        if self._complexified:
            self.Jr, self.Jc, real, comp, diag_add, c = _complexify_pack(self.Jr, self.Jc, real, comp, diag_add, c)
        self.W = self.Jr + 2 * self.Jc
        if self.W < 1 or self.W > _lib.GF_MAX_WIDTH:
            raise ValueError(
                f"celerite width {self.W} unsupported (1..{_lib.GF_MAX_WIDTH})")
        self.ld = self.lib.gf_leading_dim(self.W)
        f64 = dict(dtype=torch.float64, device=self.device)

        def dev(x):
            if isinstance(x, torch.Tensor):
                return x.to(**f64).contiguous()
            return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)

        self._dev = dev

        def rows(x, name):
            x = dev(x)
            if x.ndim == 1:
                x = x[None, :]
            if x.shape[0] not in (1, self.B):
                raise ValueError("dimension mismatch")
            return x

        def padded(x):
            # spare elements at the very end: the sweeps prefetch up to three rows ahead
            # unconditionally (a clamped index would defeat scalar-load code generation)
            buf = torch.zeros((x.shape[0] * x.shape[1] + 4,), **f64)
            buf[:x.numel()] = x.reshape(-1)
            return buf[:x.numel()].view(x.shape[0], x.shape[1]), buf

        t = rows(t, "t")
        self.N = int(t.shape[1])
        # (largest |t| and the median spacing: two reductions with a host synchronisation each -- a caller that
        # factorises the same time axis again, GaussianProcess.recompute(), passes what `axis_stats` held the first time)
        self._tmax = float(t.abs().max()) if axis_stats is None else float(axis_stats[0])
        self.t, self._tpad = padded(t)
        y = rows(y, "y")
        if y.shape[1] != self.N:
            raise ValueError("dimension mismatch")
        self.y, self._ypad = padded(y)
        self.diag = None
        if diag is not None:
            diag = rows(diag, "diag")
            if diag.shape[1] != self.N:
                raise ValueError("dimension mismatch")
            self.diag, self._dpad = padded(diag)
        # largest user diagonal per problem (condition estimates and the accuracy guard)
        self._diag_amax = None if self.diag is None else self.diag.amax(dim=1)
        self._coeff_host = (real, comp, diag_add, c)
        self._coeffs_list = list(coeffs_list)
        B, ld = self.B, self.ld
        # W <= 64: block-scaled one-wave-per-problem kernels (k_build2 / k_factor2)
        self.scaled = bool(self.lib.gf_scaled_supported(self.W)) and not force_v1
        # 64 < W <= 256: the same block-scaled rows swept by several waves per problem (k_factor2w)
        self.scaled_wide = bool(self.lib.gf_scaled_wide_supported(self.W)) and not force_v1
        # W <= 63 and phases inside the fused kernel's sincos range: nothing is materialised
        self.allow_fused = bool(allow_fused) and self.scaled and self.W <= 63
        # 64 <= W <= 192, complex terms only: the fused sweep on one workgroup per problem (k_factorw)
        self.allow_wide = (bool(allow_fused) and not force_v1 and self.W > 63
                           and bool(self.lib.gf_fused_supported(self.Jr, self.Jc)))
        # which fused sweep runs (GF_SWEEP_AUTO: lane-tiled where the term structure allows it);
        # a call argument of the C-ABI, so engines with different settings coexist in one process
        self.sweep_variant = _lib.GF_SWEEP_AUTO
        # rows between exact re-anchorings of the in-register generator (`gen_period` argument):
        # 4 = throughput setting (log-likelihood within ~2e-9 at a condition of 4e5; 3 % slower than
        # 16, which reaches 1e-8 there), 1 = exact generation every row (float64-class accuracy,
        # 15 % slower)
        self.generator_period = 4           # (1 on time axes whose phase quantum rules rotation out: below)
        # time-parallel log-likelihood WITHOUT a final pass (nominal sums + per-chunk corrections from the
        # start states: gf_chunk_corrections / gf_wide_combine(acc)); off by default -- the rows d, z then hold
        # the NOMINAL values and a non-positive pivot shows up as NaN, which the caller must resolve with a
        # final pass (BatchedLogLikelihood does, for kernels that are positive semi-definite by construction)
        self.two_sweep = False
        T = int(min(max(int(tile_rows), 1), self.N))
        if T < self.N:
            T = max(64, T // 64 * 64)         # tiles start on a reset row (any block <= 64)
        self.tile_rows = T
        # typical cadence (median spacing) sets how many rows a scaled block may span
        tt = self.t[0]
        if axis_stats is not None:
            self._dt_med = float(axis_stats[1])
        else:
            self._dt_med = float(torch.median(tt[1:] - tt[:-1])) if self.N > 1 else 0.0
        self.axis_stats = (self._tmax, self._dt_med)
        self._pack = self._make_pack(*self._coeff_host)
        # a time axis far from zero (JD: phases of 5e9 rad): rotation rows would differ from celerite2's
        # rounded-phase rows by more than the accuracy target allows at an ordinary conditioning -- exact rows
        # until a caller calibrates (period_for_condition applies the same term to a measured condition)
        self.generator_period = self.period_for_condition(self.NOMINAL_COND) if self.generator_period > 1 else 1
        self.generator_period = min(self.generator_period, 4)

        def rows_buf():
            # one spare row: the sweep prefetches row n+1 unconditionally
            return torch.zeros((B * T + 2, ld), **f64)

        self._rows_buf = rows_buf
        self._nbuf = 2 if (T < self.N and overlap_build) else 1
        self.bufs = None                      # allocated on first use (not needed when fused)
        self.d = torch.empty((B, T), **f64)
        self.z = torch.empty((B, T), **f64)
        nS = 64 * 64 if self.scaled else int(self.lib.gf_state_size(self.W))
        if self.allow_wide:
            nS = max(nS, int(self.lib.gf_fused_state_size(self.Jr, self.Jc)))
        nF = 64 if self.scaled else int(self.lib.gf_state_cols(self.W))
        self.S_state = torch.empty((B, nS), **f64)
        self.F_state = torch.empty((B, nF), **f64)
        self.acc = torch.empty((B, 3), **f64)
        self.work = torch.empty((B * int(self.lib.gf_reduce_work(T)),), **f64)
        self.info = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.out = torch.empty((B,), **f64)
        # build of tile k+1 on a side stream (overlapping the sweep of tile k) or in line
        self.overlap_build = bool(overlap_build)
        self.side = torch.cuda.Stream(device=self.device) if self.overlap_build else None
        self.time_factor = False
        self.factor_events = []

    @staticmethod
    def _bs(x):
        return 0 if x.shape[0] == 1 else x.stride(0)

    def _alloc_bufs(self):
        torch = self.torch
        f64 = dict(dtype=torch.float64, device=self.device)
        n = self.B * self.tile_rows + 2
        self.bufs = [dict(a=torch.zeros((n,), **f64), U=self._rows_buf(), V=self._rows_buf(),
                          P=None if (self.scaled or self.scaled_wide) else self._rows_buf(),
                          de=torch.zeros((n,), **f64))
                     for _ in range(self._nbuf)]

    def _fused_ok(self):
        """Phases d*t must stay inside fm_sincos's range (|x| < 1e12: FM_SINCOS_RANGE of fastmath.h)."""
        return self.allow_fused and self._pack[6] * self._tmax < SINCOS_RANGE

    def _wide_ok(self):
        """The fused sweep for wide kernels (same phase range; log-likelihood streaming only)."""
        return self.allow_wide and self._pack[6] * self._tmax < SINCOS_RANGE

    # error of the log-likelihood ~ GEN_ERR * period * condition (measured: 1e-8 at period 16 and a
    # condition of 4e5)
    GEN_ERR = 1.6e-15
    # ... + PHASE_ERR * quantum * condition / sqrt(N) for any period > 1, quantum = max|d t| 2^-53: celerite2's rows
    # (and the oracle's, and this library's exact rows) take cos / sin of theta = fl(d t), which carries up to half
    # an ulp of the PHASE as rounding -- 5e-7 rad at the 5e9 rad a JD-based axis reaches with the solar p-modes
    # (/root/reference/gadfly/gp.py:79-80) -- while a rotation step by d (t_n - t_{n-1}) follows the true phase.  The
    # two sets of rows differ incoherently (zero mean, row by row), hence the 1 / sqrt(N); measured on time axes
    # moved by 0 ... 2e6 (units of 1e6 s), N = 16 384 and 131 072, conditions of 34 and 135, periods 4 and 64
    # (tools/phase_quantum.py, profiles/r04_phase_quantum.txt): observed error <= 0.03 quantum cond / sqrt(N), up
    # to 1e-7 on a JD axis.  The kernels therefore take the step's angle from the difference of the ROUNDED
    # products wherever phases exceed QMODE_PHASE (RowGen::qmode): the same measurement then shows 1e-13 at every
    # offset and period, and the term below only covers the phases under that threshold.
    PHASE_ERR = 0.1
    #: condition assumed for the phase-quantum term before any evaluation has measured one
    NOMINAL_COND = 1.0e3

    #: phases beyond this at the first row of a tile / chunk switch the generator's rotation steps to the ROUNDED
    #: phase differences celerite2's rows carry (RowGen::qmode in gadfly_hip.hip): the phase quantum then drops out
    QMODE_PHASE = 4.0e6

    def phase_quantum(self):
        """Half an ulp of the largest phase |d t| a rotation step may still see un-corrected: tiles and chunks that
        start beyond QMODE_PHASE follow the rounded phases exactly."""
        return min(float(self._pack[6]) * self._tmax, self.QMODE_PHASE) * 2.0 ** -53

    def phase_error_coefficient(self):
        """Relative log-likelihood error per unit of condition number that a generator period > 1 adds through
        the phase quantum (see PHASE_ERR)."""
        return self.PHASE_ERR * self.phase_quantum() / math.sqrt(max(self.N, 1))

    def generator_error_coefficient(self, period):
        """err / condition of an evaluation at this generator period (0 for exact rows' own float64 class)."""
        period = int(period)
        return 0.0 if period <= 1 else self.GEN_ERR * period + self.phase_error_coefficient()

    def last_acc(self):
        """(B, 3) device tensor [sum log d, sum z^2/d, min d] of the LAST evaluation, whichever route ran."""
        if getattr(self, "_last_wide_tp", False):
            return self._wide_tp["acc"]
        return self._tp["acc"] if getattr(self, "_tp_used", False) else self.acc

    #: a two-sweep evaluation knows the pivots of its NOMINAL pass only (every chunk from a zero start: never
    #: smaller than the true ones, and within a few per cent of them a few dozen rows into a chunk): its
    #: condition estimates are scaled by this margin
    TWO_SWEEP_MARGIN = 1.5

    def condition_estimate(self):
        """max(a) / min(d) over all problems of the LAST evaluation (synchronises): the factor by
        which rounding in the generator rows shows up in the log-likelihood."""
        dmin = float(self.last_acc()[:, 2].min().item())
        if getattr(self, "_two_sweep_used", False):
            dmin /= self.TWO_SWEEP_MARGIN
        amax = float(self._pack[2].max().item())
        if self._diag_amax is not None:
            amax += float(self._diag_amax.max().item())      # (a ragged batch sets it from its real rows)
        return amax / dmin if dmin > 0.0 else float("inf")

    def period_for_condition(self, cond, target=1e-9):
        """Longest generator period (a power of two in 1..64) whose share GEN_ERR * period * cond of the
        relative log-likelihood error stays below ``target``."""
        period = 1
        while period < 64 and self.generator_error_coefficient(2 * period) * cond <= target:
            period *= 2
        return period

    def calibrate_generator(self, target=1e-9):
        """Choose the generator period (rows between exact re-anchorings, a power of two in 1..64)
        for the following evaluations from the condition estimate of the last one, so that the
        generator's contribution to the relative log-likelihood error stays below ``target``.
        Returns (condition estimate, period)."""
        cond = self.condition_estimate()
        self.generator_period = self.period_for_condition(cond, target)
        return cond, self.generator_period

    def _make_pack(self, real, comp, diag_add, c):
        dev = self._dev
        cmax = np.max(c, axis=1)
        dmax = float(np.max(np.abs(comp[3]))) if comp.size else 0.0
        # rows between forced resets of the scaled coordinates: largest power of two with
        # 1.5 * (block - 1) * cmax * cadence <= 28 (see k_build2)
        x = 1.5 * float(np.max(cmax)) * max(getattr(self, "_dt_med", 0.0), 0.0)
        block = 64
        while block > 1 and (block - 1) * x > 28.0:
            block //= 2
        # ONE host-to-device copy for the five arrays (a chain of single evaluations pays every copy's latency
        # before its first kernel starts): views into one buffer, each starting on a 16-byte boundary
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (real, comp, diag_add, c, cmax)]
        offs, total = [], 0
        for a in arrs:
            offs.append(total)
            total += (a.size + 1) // 2 * 2
        host = np.zeros(max(total, 2), dtype=np.float64)
        for a, o in zip(arrs, offs):
            host[o:o + a.size] = a.reshape(-1)
        buf = dev(host)
        real, comp, diag_add, c, cmax = (buf[o:o + a.size].view(a.shape) for a, o in zip(arrs, offs))
        return real, comp, diag_add, c, cmax, block, dmax

    def pack_coefficients(self, coeffs_list):
        if len(coeffs_list) != self.B:
            raise ValueError("coefficient pack does not match the batch structure")
        return self.pack_arrays(*_coeff_pack(coeffs_list))

    def pack_arrays(self, Jr, Jc, real, comp, diag_add, c):
        """Device pack from stacked coefficient arrays of the batch's ORIGINAL term structure (_coeff_pack /
        batch.sho_coefficient_pack)."""
        if (Jr, Jc) != self._struct0 or comp.shape[1] != self.B:
            raise ValueError("coefficient pack does not match the batch structure")
        if self._complexified:
            Jr, Jc, real, comp, diag_add, c = _complexify_pack(Jr, Jc, real, comp, diag_add, c)
        return self._make_pack(real, comp, diag_add, c)

    def use_coefficients(self, pack):
        self._pack = pack
        self._coeffs_list = None

    def _build_tile(self, k, buf, stream):
        n0 = k * self.tile_rows
        rows = min(self.tile_rows, self.N - n0)
        real, comp, diag_add, _, cmax, block, _ = self._pack
        p = _lib.ptr
        if self.scaled or self.scaled_wide:
            st = self.lib.gf_build_scaled(
                self.B, rows, n0, self.Jr, self.Jc, self.ld,
                p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]),
                p(diag_add), p(cmax), block, p(self.t), self._bs(self.t),
                p(self.diag), 0 if self.diag is None else self._bs(self.diag),
                p(buf["a"]), p(buf["U"]), p(buf["V"]), p(buf["de"]), stream.cuda_stream)
            _lib.check(st, "gf_build_scaled")
            return
        st = self.lib.gf_build_matrices(
            self.B, rows, n0, self.Jr, self.Jc, self.ld,
            p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]),
            p(diag_add), p(self.t), self._bs(self.t),
            p(self.diag), 0 if self.diag is None else self._bs(self.diag),
            p(buf["a"]), p(buf["U"]), p(buf["V"]), p(buf["P"]), stream.cuda_stream)
        _lib.check(st, "gf_build_matrices")

    @_on_device
    def log_likelihood(self):
        """Enqueue one evaluation per problem; returns the (B,) device tensor (no sync)."""
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        main = torch.cuda.current_stream(self.device)
        side = self.side if self.overlap_build else main
        T, N, B = self.tile_rows, self.N, self.B
        ntiles = (N + T - 1) // T
        self.S_state.zero_()
        self.F_state.zero_()
        self.info.zero_()
        self._tp_used, self._last_wide_tp = False, False     # (whose `acc` the last evaluation filled)
        self._two_sweep_used = False
        if self._fused_ok() or self._wide_ok():
            return self._log_likelihood_fused(main)
        if self.bufs is None:
            self._alloc_bufs()
        nb = len(self.bufs)
        built = [None] * nb          # event: tile in buffer i is built
        freed = [None] * nb          # event: sweep that used buffer i is done
        if side is not main:
            side.wait_stream(main)
        for k in range(min(nb, ntiles)):
            with torch.cuda.stream(side):
                self._build_tile(k, self.bufs[k % nb], side)
                built[k % nb] = side.record_event()
        for k in range(ntiles):
            i = k % nb
            buf = self.bufs[i]
            n0 = k * T
            rows = min(T, N - n0)
            main.wait_event(built[i])
            if self.time_factor:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(main)
            if self.scaled or self.scaled_wide:
                st = lib.gf_factor_scaled(
                    B, rows, n0, self.W, self.ld, p(self._pack[3]), p(buf["a"]),
                    p(buf["U"]), p(buf["V"]), p(buf["de"]), self.y.data_ptr() + 8 * n0,
                    self._bs(self.y), p(self.d), p(self.z), p(self.S_state),
                    p(self.F_state), p(self.info), main.cuda_stream)
            else:
                st = lib.gf_factor(
                    B, rows, n0, self.W, self.ld, p(buf["a"]), p(buf["U"]), p(buf["V"]),
                    p(buf["P"]), self.y.data_ptr() + 8 * n0, self._bs(self.y),
                    p(self.d), None, p(self.z), p(self.S_state), p(self.F_state),
                    p(self.info), main.cuda_stream)
            _lib.check(st, "gf_factor")
            if self.time_factor:
                e1.record(main)
                self.factor_events.append((e0, e1, rows))
            freed[i] = main.record_event()
            if k + nb < ntiles:          # refill this buffer with tile k + nb on the side stream
                side.wait_event(freed[i])
                with torch.cuda.stream(side):
                    self._build_tile(k + nb, buf, side)
                    built[i] = side.record_event()
            st = lib.gf_reduce_tile(B, rows, p(self.d), p(self.z), p(self.work), p(self.acc),
                                    1 if k == 0 else 0, main.cuda_stream)
            _lib.check(st, "gf_reduce_tile")
        out = torch.empty((B,), dtype=torch.float64, device=self.device)
        st = lib.gf_loglike_finish(B, N, p(self.acc), p(self.info), p(out), None,
                                   main.cuda_stream)
        _lib.check(st, "gf_loglike_finish")
        return out

    def _log_likelihood_fused(self, main):
        """gf_loglike_fused per tile: generator rows never leave the registers."""
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        T, N, B = self.tile_rows, self.N, self.B
        real, comp, diag_add, _, cmax, block, _ = self._pack
        period, variant = int(self.generator_period), int(self.sweep_variant)
        self.kernel_used = "fused-wide" if self.W > 63 else "fused"
        for k in range((N + T - 1) // T):
            n0 = k * T
            rows = min(T, N - n0)
            if self.time_factor:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(main)
            st = lib.gf_loglike_fused(
                B, rows, n0, self.Jr, self.Jc, block, period, variant,
                p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]),
                p(diag_add), p(cmax), p(self.t), self._bs(self.t),
                p(self.diag), 0 if self.diag is None else self._bs(self.diag),
                p(self.y), self._bs(self.y), p(self.d), p(self.z),
                p(self.S_state), p(self.F_state), p(self.info), main.cuda_stream)
            _lib.check(st, "gf_loglike_fused")
            if self.time_factor:
                e1.record(main)
                self.factor_events.append((e0, e1, rows))
            st = lib.gf_reduce_tile(B, rows, p(self.d), p(self.z), p(self.work), p(self.acc),
                                    1 if k == 0 else 0, main.cuda_stream)
            _lib.check(st, "gf_reduce_tile")
        out = torch.empty((B,), dtype=torch.float64, device=self.device)
        st = lib.gf_loglike_finish(B, N, p(self.acc), p(self.info), p(out), None,
                                   main.cuda_stream)
        _lib.check(st, "gf_loglike_finish")
        return out

    # -- exact time-parallel evaluation (few problems, long series) ------------------------
    #: two-sweep evaluations of at most this many rows in all (B N), at widths beyond 48, run on ~1024 chunks instead of ~2048
    two_sweep_small_rows = 1_500_000

    def _tp_chunking(self, chunk_len, store=False):
        N, B = self.N, self.B
        block = self._pack[5]
        if chunk_len is None:
            # B * nch ~ 2048 waves = 2 per SIMD, the occupancy the sweep kernels are built for; below ~512 rows per
            # chunk the extra tree levels cost more than the sweeps save.  (+ 1: the last chunk sits out the nominal
            # pass, the first one the final pass; the two-sweep log-likelihood sweeps all chunks in its nominal pass.)
            # The two-sweep route pays per chunk (tree levels, corrections) what the three-sweep route pays per row
            # (the final pass).  At widths beyond 48 -- whose combine kernels fit one workgroup per CU -- it is
            # faster on half the chunks, one wave per SIMD, up to 1.5e6 rows in all: B = 1, N = 1e6, W = 60: 3.67 ms
            # on 977 chunks, 3.89 on 1954; 16 x 65 000: 3.16 against 3.44; beyond 2e6 rows the sweeps dominate and
            # 2048 wins again.  At widths <= 48 (two or four combine workgroups per CU) 2048 wins throughout
            # (grid over B, N, W = 20 / 40 / 60: DESIGN.md 6)
            two = self.two_sweep and not store
            waves = 1024 if (two and self.W > 48 and B * N <= self.two_sweep_small_rows) else 2048
            nch = max(1, waves // B) + (0 if two else 1)
            chunk_len = max(512, -(-N // nch))
        chunk_len = max(block, (int(chunk_len) + 63) // 64 * 64)
        return chunk_len, -(-N // chunk_len)

    @_on_device
    def _tp_run(self, chunk_len=None, store=False):
        """Chunk-parallel factor + forward solve.  store=True also keeps the factor in scaled
        form (u~, w~ rows, reset spans, per-chunk true transitions) for :class:`ScaledFactor`."""
        if not self._fused_ok():
            raise ValueError("time-parallel evaluation needs W <= 63 and |d t| < 1e12")
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        N, B = self.N, self.B
        real, comp, diag_add, _, cmax, block, _ = self._pack
        chunk_len, nch = self._tp_chunking(chunk_len, store)
        opts = (int(self.generator_period), int(self.sweep_variant))
        f64 = dict(dtype=torch.float64, device=self.device)
        key = (chunk_len, nch)

        def rows(*tail):
            # a row array with its eight spare rows (fetched ahead unconditionally, never used) cleared; the
            # rows themselves are written by the sweeps before anything reads them -- clearing 0.5 GB per
            # array and compute() at N = 1e6 was 0.3 ms of memsets
            x = torch.empty((B * N + 8,) + tail, **f64)
            x[B * N:].zero_()
            return x

        if getattr(self, "_tp_key", None) != key:
            ns = B * nch                    # chunk maps and states: one slot per (problem, chunk)
            self._tp = dict(
                S=torch.empty((ns, 4096), **f64), F=torch.empty((ns, 64), **f64),
                Phi=torch.empty((ns, 4096), **f64), G=torch.empty((ns, 4096), **f64),
                m=torch.empty((ns, 64), **f64),
                # (eight spare rows: the transition sweep fetches r-bar / d-bar rows ahead, unconditionally)
                d=rows(), z=rows(), r=rows(64), Un=rows(64), den=rows(),
                info=torch.zeros((B * nch,), dtype=torch.int32, device=self.device),
                work=torch.empty((B * int(lib.gf_reduce_work(N)),), **f64),
                acc=torch.empty((B, 3), **f64))
            self._tp_key = key
        w = self._tp
        if store and "Ut" not in w:
            w["Ut"] = rows(64)
            w["Wt"] = torch.empty((B * N, 64), **f64)
            w["de"] = rows()
            # the factor's own copy of the chunk transitions: a later non-storing evaluation
            # (log_likelihood of another y) refills "Phi" with the NOMINAL pass' transitions
            w["PhiT"] = torch.empty((B * nch, 4096), **f64)
        st = torch.cuda.current_stream(self.device).cuda_stream
        w["info"].zero_()
        # nominal passes start every chunk from zero by themselves: the state slots are outputs only
        zopts = (opts[0], opts[1] | _lib.GF_SWEEP_ZERO_START)
        coeffs = (p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]))
        tyd = (p(self.t), self._bs(self.t), p(self.diag),
               0 if self.diag is None else self._bs(self.diag), p(self.y), self._bs(self.y))
        none3 = (None, None, None)
        ci = w["info"].view(B, nch)

        def transition(first, count, Ut, de, phi="Phi"):
            rc = lib.gf_chunk_transition(B, N, chunk_len, nch, first, count, self.Jr, self.Jc,
                                         int(self.sweep_variant), p(self._pack[3]), p(de), p(w["d"]), p(w["z"]),
                                         p(w["r"]), p(Ut), p(w[phi]), p(w["G"]), p(w["m"]), st)
            _lib.check(rc, "gf_chunk_transition")

        if self.two_sweep and not store and nch > 1 and B * nch <= 65535:
            return self._tp_two_sweep(w, chunk_len, nch, zopts, coeffs, tyd, transition, st)
        # chunk 0 starts from the zero state: its nominal pass IS its final pass (not when the factor is
        # stored: the final pass also writes the w~ rows)
        skip_first = nch > 1 and not store
        ci0 = None
        if nch > 1:
            # nominal pass (u~ rows and reset spans stored for the transition sweep).  Nothing of the LAST
            # chunk's map is ever needed: it is left out
            rc = lib.gf_chunk_sweep(B, N, chunk_len, nch, 0, nch - 1, self.Jr, self.Jc, block, *zopts, *coeffs,
                                    p(diag_add), p(cmax), *tyd, p(w["d"]), p(w["z"]), p(w["r"]),
                                    p(w["Un"]), None, p(w["den"]), p(w["S"]), p(w["F"]), p(w["info"]), st)
            _lib.check(rc, "gf_chunk_sweep")
            self._clear_slots(w["S"], nch, nch - 1)
            self._clear_slots(w["F"], nch, nch - 1)
            # transitions and Gram sums of the chunks 1 .. nch - 2; from a zero start state the first chunk's
            # map acts through its end state alone (Phi = G = m = 0 for the combine), the last one's not at all
            transition(1, nch - 2, w["Un"], w["den"])
            for k in ("Phi", "G", "m"):
                self._clear_slots(w[k], nch, 0, nch - 1)
            self._tp_combine(w, nch, st)
            # a nominal pass (zero start state: pivots >= the true ones) can only fail at or after
            # the true failing row; the final pass decides, from exact start states up to there --
            # except for chunk 0, whose nominal pass is exact
            ci0 = ci[:, 0].clone()
            w["info"].zero_()
        stores = (p(w["Ut"]), p(w["Wt"]), p(w["de"])) if store else none3
        f0 = 1 if skip_first else 0         # (chunk 0's d, z rows stay where the nominal pass wrote them)
        rc = lib.gf_chunk_sweep(B, N, chunk_len, nch, f0, nch - f0, self.Jr, self.Jc, block,
                                *(zopts if nch == 1 else opts), *coeffs,
                                p(diag_add), p(cmax), *tyd, p(w["d"]), p(w["z"]),
                                p(w["r"]) if (store and nch > 1) else None, *stores,
                                p(w["S"]), p(w["F"]), p(w["info"]), st)
        _lib.check(rc, "gf_chunk_sweep")
        if skip_first:
            ci[:, 0] = ci0
        # (the TRUE chunk transitions -- the transition sweep once more, on the true rows -- are part of the stored
        # factor but needed by the solves only: ScaledFactor computes them at its first solve, compute() does not
        # wait for them: 0.67 of its 6 ms of kernels at N = 1e6)
        self._tp_generation = getattr(self, "_tp_generation", 0) + 1
        rc = lib.gf_reduce_tile(B, N, p(w["d"]), p(w["z"]), p(w["work"]), p(w["acc"]), 1, st)
        _lib.check(rc, "gf_reduce_tile")
        # a chunk that failed marks its problem with the FIRST non-positive pivot (celerite2 and the
        # sequential sweep stop there; chunks after a failed one ran from meaningless start states)
        big = torch.iinfo(torch.int32).max
        first = torch.where(ci > 0, ci, torch.full_like(ci, big)).min(dim=1).values
        self.info.copy_(torch.where(first == big, torch.zeros_like(first), first))
        out = torch.empty((B,), **f64)
        rc = lib.gf_loglike_finish(B, N, p(w["acc"]), p(self.info), p(out), None, st)
        _lib.check(rc, "gf_loglike_finish")
        self._tp_used, self._last_wide_tp, self._two_sweep_used = True, False, False
        return out, chunk_len, nch

    def _tp_two_sweep(self, w, chunk_len, nch, opts, coeffs, tyd, transition, st):
        """Log-likelihood from TWO sweeps: the nominal pass over all chunks (its rows reduced as they are), the
        transition sweep of the chunks 1 ... nch - 1, the combine for the start states, and per chunk the
        corrections  log det(I - X G)  and  e^T G v - 2 m^T e - m^T X m  (gf_chunk_corrections) -- no final pass."""
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        N, B = self.N, self.B
        real, comp, diag_add, _, cmax, block, _ = self._pack
        self._tp_generation = getattr(self, "_tp_generation", 0) + 1
        rc = lib.gf_chunk_sweep(B, N, chunk_len, nch, 0, nch, self.Jr, self.Jc, block, *opts, *coeffs,
                                p(diag_add), p(cmax), *tyd, p(w["d"]), p(w["z"]), p(w["r"]),
                                p(w["Un"]), None, p(w["den"]), p(w["S"]), p(w["F"]), p(w["info"]), st)
        _lib.check(rc, "gf_chunk_sweep")
        rc = lib.gf_reduce_tile(B, N, p(w["d"]), p(w["z"]), p(w["work"]), p(w["acc"]), 1, st)
        _lib.check(rc, "gf_reduce_tile")
        transition(1, nch - 1, w["Un"], w["den"])
        for k in ("Phi", "G", "m"):
            self._clear_slots(w[k], nch, 0)
        self._tp_combine(w, nch, st)
        if "corr" not in w:
            w["corr"] = torch.empty((int(lib.gf_chunk_corrections_work(B, nch)),), dtype=torch.float64,
                                    device=self.device)
        rc = lib.gf_chunk_corrections(B, nch, self.W, 1, nch - 1, p(w["S"]), p(w["F"]), p(w["G"]), p(w["m"]),
                                      p(w["acc"]), p(w["corr"]), st)
        _lib.check(rc, "gf_chunk_corrections")
        # a chunk of the nominal pass that failed: the first such row (exact for chunk 0; later chunks start
        # from zero, where pivots are never smaller than the true ones: the caller's final pass decides)
        ci = w["info"].view(B, nch)
        big = torch.iinfo(torch.int32).max
        first = torch.where(ci > 0, ci, torch.full_like(ci, big)).min(dim=1).values
        self.info.copy_(torch.where(first == big, torch.zeros_like(first), first))
        out = torch.empty((B,), dtype=torch.float64, device=self.device)
        rc = lib.gf_loglike_finish(B, N, p(w["acc"]), p(self.info), p(out), None, st)
        _lib.check(rc, "gf_loglike_finish")
        self._tp_used, self._last_wide_tp, self._two_sweep_used = True, False, True
        return out, chunk_len, nch

    #: chunk counts above this use the log-depth tree combine (2 log2 P levels of ~0.13 ms)
    #: instead of the sequential one (~0.12 ms per chunk)
    tree_min_chunks = 24

    def _clear_slots(self, x, nch, *slots):
        """Zero the chunk slots `slots` of every problem in x ([B * nch, n]): one strided fill kernel each."""
        v = x.view(self.B, nch, -1)
        for s in slots:
            v[:, s].zero_()

    def _tp_combine(self, w, nch, st):
        """S/F slot c <- true start state of chunk c (sequential or tree LFT combine)."""
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        B = self.B
        if nch < self.tree_min_chunks:
            rc = lib.gf_chunk_combine(B, nch, self.W, p(w["Phi"]), p(w["G"]), p(w["m"]), p(w["S"]),
                                      p(w["F"]), st)
            _lib.check(rc, "gf_chunk_combine")
            return
        # The scan runs on the chunk maps where the sweeps put them (slot = problem * nch + chunk) and overwrites
        # them: G and m, which the two-sweep corrections read afterwards, go through copies; its output buffers
        # change places with the state slots instead of being copied back.  Its own slots beyond nch (P per problem,
        # the power of two at or above nch) live in a work buffer.
        P = 1 << (nch - 1).bit_length()
        tr = w.get("tree")
        if tr is None or tr["key"] != (P, nch):
            f64 = dict(dtype=torch.float64, device=self.device)
            tr = w["tree"] = dict(
                key=(P, nch), G=torch.empty((B * nch, 4096), **f64), X=torch.empty((B * nch, 4096), **f64),
                m=torch.empty((B * nch, 64), **f64), Y=torch.empty((B * nch, 64), **f64),
                work=torch.empty((int(lib.gf_chunk_combine_tree_work(B, P, nch)),), **f64))
        tr["G"].copy_(w["G"])
        tr["m"].copy_(w["m"])
        rc = lib.gf_chunk_combine_tree(B, P, nch, self.W, p(w["Phi"]), p(tr["G"]), p(tr["m"]), p(w["S"]),
                                       p(w["F"]), p(tr["X"]), p(tr["Y"]), p(tr["work"]), st)
        _lib.check(rc, "gf_chunk_combine_tree")
        w["S"], tr["X"] = tr["X"], w["S"]
        w["F"], tr["Y"] = tr["Y"], w["F"]

    # -- exact time-parallel evaluation of FEW series with a wide kernel -----------------------------
    #: shortest series for which the wide time-parallel run replaces the sequential sweep
    wide_tp_min_rows = 16384

    WIDE_TP_COEF = 0.1

    def _wide_slots(self):
        """Workgroups of the wide sweep the chip runs at full per-workgroup speed: three per CU up to
        W = 95 (four waves each), two up to W = 127, one beyond (seven waves of ~230 VGPRs)."""
        nw = (self.W + 32) // 32
        return 768 if nw <= 3 else (512 if nw == 4 else 256)

    def _wide_tp_ok(self):
        """Chunking pays (three sweeps of N / nch rows instead of one of N) while the batch alone leaves
        most of the workgroup slots empty."""
        return (self._wide_ok() and self.N >= self.wide_tp_min_rows
                and 6 * self.B <= self._wide_slots())

    def _wide_chunking(self, chunk_len):
        if chunk_len is None:
            if self.B == 1:
                # three sweeps of N / nch rows (~1.4 us per row at W = 172) against the dense combine's
                # 2 log2(nch) levels: nch ~ sqrt(WIDE_TP_COEF N), a power of two (the scan pads to one)
                want = max(2.0, math.sqrt(self.WIDE_TP_COEF * self.N))
                nch = min(1 << int(round(math.log2(want))), 512)
            else:
                # a batch: fill the workgroup slots, chunks of at least 2048 rows
                nch = max(2, min(self._wide_slots() // self.B, self.N // 2048))
            chunk_len = -(-self.N // nch)
        chunk_len = max(64, (int(chunk_len) + 63) // 64 * 64)
        return chunk_len, -(-self.N // chunk_len)

    def _wide_ws(self, L, nch, keep):
        """Row and state buffers of the wide time-parallel run, kept between evaluations (rows of all
        problems back to back, eight spare rows: the sweeps prefetch ahead unconditionally)."""
        torch, lib = self.torch, self.lib
        key = (L, nch)
        # (a stored factor keeps r-bar / h / Phi for its lazily built chunk transitions: buffers of its own)
        ws = None if keep else getattr(self, "_wide_ws_cache", None)
        if ws is None or ws["key"] != key:
            B, N = self.B, self.N
            ld = int(lib.gf_fused_row_stride(self.Jr, self.Jc))
            nS = int(lib.gf_fused_state_size(self.Jr, self.Jc))
            f64 = dict(dtype=torch.float64, device=self.device)
            rows = B * N + 8
            ws = dict(key=key, ld=ld, nS=nS,
                      dbar=torch.empty((rows,), **f64), zbar=torch.empty((rows,), **f64),
                      rbar=torch.empty((rows, ld), **f64), h=torch.empty((rows, ld), **f64),
                      S=torch.empty((B * nch, nS), **f64), Phi=torch.empty((B * nch, nS), **f64),
                      cinfo=torch.zeros((B * nch,), dtype=torch.int32, device=self.device),
                      work=torch.empty((max(1, int(lib.gf_wide_combine_work(B, max(nch, 2), self.Jc))),), **f64),
                      red=torch.empty((B * int(lib.gf_reduce_work(N)),), **f64),
                      acc=torch.empty((B, 3), **f64))
            if not keep:
                self._wide_ws_cache = ws
        def rows(name):
            """row buffer `name` of the workspace, allocated on first use"""
            if name not in ws:
                shape = (self.B * self.N + 8, ws["ld"]) if name == "Ut" else (self.B * self.N + 8,)
                ws[name] = torch.empty(shape, dtype=torch.float64, device=self.device)
            return ws[name]

        ws["rows"] = rows
        return ws

    @_on_device
    def _tp_run_wide(self, chunk_len=None, stores=None, d=None, z=None, info=None):
        """Chunk-parallel factor + forward solve of B series with a wide kernel (64 <= W <= 176):
        nominal pass (k_factorw, zero start states, rows stored) -> closed-loop transitions (k_phiw)
        -> Gram sums and the tree combine of the W x W chunk maps (gf_wide_combine: FP64-MFMA GEMM
        jobs and one gf_dense_solve per level, all in the library) -> final pass from the true start
        states.  Fills d, z (and the stored factor rows); returns (loglike (B,), logdet (B,), chunk_len,
        nch)."""
        if not self._wide_ok():
            raise ValueError("wide time-parallel evaluation needs a wide fused kernel (64 <= W <= 176)")
        torch = self.torch
        lib, p = self.lib, _lib.ptr
        B, N = self.B, self.N
        real, comp, diag_add, c, cmax, block, _ = self._pack
        L, nch = self._wide_chunking(chunk_len)
        keep = stores is not None and nch > 1
        ws = self._wide_ws(L, nch, keep=stores is not None)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if stores is None:
            Ut, Wt, de = ws["rows"]("Ut"), None, ws["rows"]("de")
        else:
            Ut, Wt, de = stores
        dbar, zbar, rbar, h, S, Phi, cinfo = (ws[k] for k in ("dbar", "zbar", "rbar", "h", "S", "Phi", "cinfo"))
        opts = (int(self.generator_period), _lib.GF_SWEEP_AUTO)
        coeffs = (p(real[0]), p(real[1]), p(comp[0]), p(comp[1]), p(comp[2]), p(comp[3]))
        tyd = (p(self.t), self._bs(self.t), p(self.diag),
               0 if self.diag is None else self._bs(self.diag), p(self.y), self._bs(self.y))

        def sweep(first, count, dd, zz, r_out, st_rows, zero_start=False):
            # (zero_start: a nominal pass -- the state slots are outputs only, nothing has to be cleared)
            variant = opts[1] | (_lib.GF_SWEEP_ZERO_START if zero_start else 0)
            rc = lib.gf_chunk_sweep(B, N, L, nch, first, count, self.Jr, self.Jc, block, opts[0], variant,
                                    *coeffs, p(diag_add), p(cmax), *tyd, p(dd), p(zz), r_out, *st_rows,
                                    p(S), None, p(cinfo), st)
            _lib.check(rc, "gf_chunk_sweep")

        cinfo.zero_()
        ci = cinfo.view(B, nch)
        if self.two_sweep and stores is None and nch > 1:
            # log-likelihood from TWO sweeps: nominal pass over all chunks (its rows reduced as they are),
            # transitions of the chunks 1 ... nch - 1, the combine with the per-chunk corrections added to the
            # accumulators (gf_wide_combine with acc) -- no final pass
            acc = ws["acc"]
            sweep(0, nch, dbar, zbar, p(rbar), (p(Ut), None, p(de)), zero_start=True)
            _lib.check(lib.gf_reduce_tile(B, N, p(dbar), p(zbar), p(ws["red"]), p(acc), 1, st), "gf_reduce_tile")
            rc = lib.gf_chunk_transition_wide(B, N, L, nch, 1, nch - 1, self.Jc, p(c), p(de), p(dbar),
                                              p(rbar), p(Ut), p(h), p(Phi), st)
            _lib.check(rc, "gf_chunk_transition_wide")
            rc = lib.gf_wide_combine(B, N, L, nch, self.Jc, p(h), p(dbar), p(zbar), p(Phi), p(S), p(acc),
                                     p(ws["work"]), st)
            _lib.check(rc, "gf_wide_combine")
            big = torch.iinfo(torch.int32).max
            first = torch.where(ci > 0, ci, torch.full_like(ci, big)).min(dim=1).values
            self.info.copy_(torch.where(first == big, torch.zeros_like(first), first))
            out = torch.empty((B,), dtype=torch.float64, device=self.device)
            logdet = torch.empty_like(out)
            _lib.check(lib.gf_loglike_finish(B, N, p(acc), p(self.info), p(out), p(logdet), st),
                       "gf_loglike_finish")
            self._wide_tp = dict(d=dbar, z=zbar, acc=acc, chunk_len=L, nch=nch)
            self._last_wide_tp, self._tp_used, self._two_sweep_used = True, False, True
            return out, logdet, L, nch
        skip_first = nch > 1 and stores is None     # chunk 0 starts from zero: its nominal pass IS its final pass
        ci0 = None
        if nch > 1:
            # 1. nominal pass: zero start states; d-bar, z-bar, r-bar, u~ rows, reset spans.  Nothing of the LAST
            #    chunk's map is ever needed (gf_wide_combine): it is left out
            sweep(0, nch - 1, dbar, zbar, p(rbar), (p(Ut), None, p(de)), zero_start=True)
            self._clear_slots(S, nch, nch - 1)
            # 2. closed-loop transitions and the rows h (not for the first chunk either: from a zero start
            #    state its map acts through its end state alone)
            rc = lib.gf_chunk_transition_wide(B, N, L, nch, 1, nch - 2, self.Jc, p(c), p(de), p(dbar),
                                              p(rbar), p(Ut), p(h), p(Phi), st)
            _lib.check(rc, "gf_chunk_transition_wide")
            # 3. chunk maps (Phi, G, Xbar, Ybar, m) and their tree combine: S <- true start states
            rc = lib.gf_wide_combine(B, N, L, nch, self.Jc, p(h), p(dbar), p(zbar), p(Phi), p(S), None,
                                     p(ws["work"]), st)
            _lib.check(rc, "gf_wide_combine")
            ci0 = ci[:, 0].clone()      # (a nominal pass can only fail at or after the true failing row: the
            cinfo.zero_()               #  final pass decides -- except for chunk 0, whose nominal pass is exact)
        # 4. final pass from the true start states
        dd = d.reshape(-1) if d is not None else ws["rows"]("d")
        zz = z.reshape(-1) if z is not None else ws["rows"]("z")
        if skip_first:
            L0 = min(L, N)
            dd[:B * N].view(B, N)[:, :L0] = dbar[:B * N].view(B, N)[:, :L0]
            zz[:B * N].view(B, N)[:, :L0] = zbar[:B * N].view(B, N)[:, :L0]
        f0 = 1 if skip_first else 0
        sweep(f0, nch - f0, dd, zz, p(rbar) if keep else None,     # (the TRUE factor's r rows replace the nominal ones)
              (p(Ut), p(Wt), p(de)) if stores is not None else (None, None, None), zero_start=nch == 1)
        if skip_first:
            ci[:, 0] = ci0
        self._wide_tp_bufs = dict(r=rbar, h=h, Phi=Phi) if keep else None
        # a chunk that failed marks its problem with the FIRST non-positive pivot
        big = torch.iinfo(torch.int32).max
        first = torch.where(ci > 0, ci, torch.full_like(ci, big)).min(dim=1).values
        flag = torch.where(first == big, torch.zeros_like(first), first)
        if info is not None:
            info.copy_(flag)
        self.info.copy_(flag)
        acc = ws["acc"]
        out = torch.empty((B,), dtype=torch.float64, device=self.device)
        logdet = torch.empty_like(out)
        _lib.check(lib.gf_reduce_tile(B, N, p(dd), p(zz), p(ws["red"]), p(acc), 1, st), "gf_reduce_tile")
        _lib.check(lib.gf_loglike_finish(B, N, p(acc), p(self.info), p(out), p(logdet), st),
                   "gf_loglike_finish")
        self._wide_tp = dict(d=dd, z=zz, acc=acc, chunk_len=L, nch=nch)
        self._last_wide_tp, self._tp_used, self._two_sweep_used = True, False, False
        return out, logdet, L, nch

    @_on_device
    def log_likelihood_time_parallel(self, chunk_len=None):
        """
        One evaluation per problem with the time axis swept in parallel chunks and stitched by
        the exact LFT combine (gf_chunk_sweep / gf_chunk_transition / gf_chunk_combine).
        Same result as :meth:`log_likelihood` to rounding; ~3.5x the flops but O(N / nch)
        sequential depth -- the latency path for B = 1.  Needs the fused kernel's conditions.
        """
        if self._wide_ok() and not self._fused_ok():
            return self._tp_run_wide(chunk_len)[0]
        if not self._fused_ok():
            raise ValueError("time-parallel evaluation needs W <= 63 and |d t| < 1e12")
        if self._tp_chunking(chunk_len)[1] == 1:
            return self.log_likelihood()
        return self._tp_run(chunk_len, store=False)[0]

    @_on_device
    def stored_factor(self, chunk_len=None):
        """Factorise and keep the factor for triangular sweeps (time-parallel for W <= 63; one
        sequential pass of the fused wide sweep beyond)."""
        if self._wide_ok() and not self._fused_ok():
            return WideFactor(self, chunk_len=chunk_len)
        _, chunk_len, nch = self._tp_run(chunk_len, store=True)
        return ScaledFactor(self, chunk_len, nch)

    def set_y(self, resid):
        """Replace the right-hand side(s) (same shape as at construction)."""
        r = self._dev(resid)
        if r.ndim == 1:
            r = r[None, :]
        if r.shape != self.y.shape:
            raise ValueError("dimension mismatch")
        self.y.copy_(r)

    @_on_device
    def evaluate(self, time_parallel=None):
        """(loglike (B,), logdet (B,)) device tensors; picks the time-parallel evaluation for
        few long series (B * N large per problem, B small) unless told otherwise."""
        auto = time_parallel is None
        force = getattr(self, "force_streaming", False)
        if auto:
            # chunking costs ~3.5x the flops: it pays while the batch alone fills less than
            # ~1/8 of the 2048 wave slots (``force_streaming``: benchmarks of the streamed sweep)
            time_parallel = self._fused_ok() and self.B <= 256 and self.N >= 8192 and not force
        # few long series with a wide kernel: the exact time-parallel evaluation on W x W chunk maps
        wide_tp = (self._wide_tp_ok() and not self._fused_ok()
                   and (bool(time_parallel) or (auto and not force)))
        if time_parallel and self._fused_ok():
            out = self.log_likelihood_time_parallel()
        elif wide_tp:
            out = self._tp_run_wide()[0]
        else:
            out = self.log_likelihood()
        acc = self.last_acc()
        torch = self.torch
        logdet = torch.where(self.info != 0, torch.full_like(acc[:, 0], float("-inf")), acc[:, 0])
        return out, logdet

    @property
    def nbytes_algorithmic_loglike(self):
        return 8 * self.N * (3 * self.W + 4) * self.B
