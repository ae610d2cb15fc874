"""
Host-side celerite term algebra for the sum-of-SHO kernels gadfly builds.

The reference assembles its kernel as
``celerite2.terms.TermConvolution(TermSum(SHOTerm x J), delta)``
(/root/reference/gadfly/core.py:336, :371-373, :379, :394).  celerite2 is a
third-party dependency that is not part of the reference tree, so the classes
here restate the *published* celerite algebra (Foreman-Mackey et al. 2017;
SURVEY.md Appendix A.1-A.4) in a compact complex-exponential form:

    k(tau) = sum_r a_r exp(-c_r tau)
           + sum_c Re[(a_c - i b_c) exp(-(c_c - i d_c) tau)],  tau = |t_i - t_j|

Only O(J) coefficient arithmetic lives here (it runs once per kernel on the
host); every O(N) operation is done on the GPU by ``gadfly_amd.csrc``.

``Term.get_value`` / ``Term.get_psd`` are closed forms used for plotting,
for the dense small-M conditional variance (celerite2 builds that densely too)
and by the tests.
"""
import numpy as np

__all__ = ["Term", "SHOTerm", "TermSum", "TermConvolution"]

_EMPTY = np.empty(0, dtype=np.float64)


def _as_vec(x):
    return np.atleast_1d(np.asarray(x, dtype=np.float64))


class Term:
    """Base class: a celerite kernel defined by six coefficient vectors."""

    name = None

    # -- to be provided by subclasses ------------------------------------
    def get_coefficients(self):
        """Return ``(a_real, c_real, a_comp, b_comp, c_comp, d_comp)``."""
        raise NotImplementedError

    # -- derived quantities ----------------------------------------------
    @property
    def terms(self):
        return (self,)

    def __add__(self, other):
        return TermSum(self, other)

    def __radd__(self, other):
        if other == 0:
            return self
        return TermSum(other, self)

    def __len__(self):
        ar, _, ac, _, _, _ = self.get_coefficients()
        return len(ar) + 2 * len(ac)

    def get_width(self):
        """celerite width W = J_real + 2 J_complex (columns of U, V, W)."""
        return len(self)

    def get_diag_shift(self):
        """Amount added to the user's ``diag`` before the matrix build."""
        return 0.0

    def get_device_coefficients(self):
        """
        Everything the device matrix build needs:
        ``(a_real, c_real, a_comp, b_comp, c_comp, d_comp, diag_shift)``.
        """
        ar, cr, ac, bc, cc, dc = (
            _as_vec(v) for v in self.get_coefficients()
        )
        return ar, cr, ac, bc, cc, dc, float(self.get_diag_shift())

    def get_value(self, tau):
        """Covariance function k(|tau|)."""
        ar, cr, ac, bc, cc, dc = (_as_vec(v) for v in self.get_coefficients())
        tau = np.abs(np.asarray(tau, dtype=np.float64))
        t = tau[..., None]
        k = np.zeros(tau.shape, dtype=np.float64)
        if len(ar):
            k = k + np.sum(ar * np.exp(-cr * t), axis=-1)
        if len(ac):
            k = k + np.sum(
                (ac * np.cos(dc * t) + bc * np.sin(dc * t)) * np.exp(-cc * t),
                axis=-1,
            )
        return k

    def get_psd(self, omega):
        """Power spectral density at angular frequency ``omega``.

        Normalisation is celerite's (and /root/reference/gadfly/core.py:33-41):
        S(w) = sqrt(2/pi) * ...
        """
        ar, cr, ac, bc, cc, dc = (_as_vec(v) for v in self.get_coefficients())
        omega = np.asarray(omega, dtype=np.float64)
        w2 = (omega ** 2)[..., None]
        psd = np.zeros(omega.shape, dtype=np.float64)
        if len(ar):
            psd = psd + np.sum(ar * cr / (cr ** 2 + w2), axis=-1)
        if len(ac):
            c2 = cc ** 2
            d2 = dc ** 2
            num = (ac * cc + bc * dc) * (c2 + d2) + (ac * cc - bc * dc) * w2
            den = w2 ** 2 + 2.0 * (c2 - d2) * w2 + (c2 + d2) ** 2
            psd = psd + np.sum(num / den, axis=-1)
        return np.sqrt(2.0 / np.pi) * psd

    def to_dense(self, x, diag):
        """
        Dense covariance matrix *as the semiseparable solver sees it*:
        ``diag(a) + tril(U V^T o Phi, -1) + triu(...)^T``, i.e.
        the off-diagonal uses :meth:`get_coefficients` at every lag and the
        diagonal is ``diag + diag_shift + sum(a_r) + sum(a_c)``.
        Small N only (tests, dense conditional variance).
        """
        x = np.asarray(x, dtype=np.float64)
        K = Term.get_value(self, x[:, None] - x[None, :])
        K[np.diag_indices_from(K)] += (
            np.asarray(diag, dtype=np.float64) + self.get_diag_shift()
        )
        return K


class SHOTerm(Term):
    r"""
    Stochastically driven, damped harmonic oscillator
    (celerite2 ``SHOTerm(S0, w0, Q, eps=1e-5)``; constructed by the reference at
    /root/reference/gadfly/core.py:371-373 with keys exactly ``S0, w0, Q``).

    PSD:  S(w) = sqrt(2/pi) S0 w0^4 / ((w^2-w0^2)^2 + w^2 w0^2 / Q^2)
    (/root/reference/gadfly/core.py:33-41).
    The alternative ``sigma/rho/tau`` parameterisation of celerite2 is also
    accepted (w0 = 2pi/rho, Q = w0 tau / 2, S0 = sigma^2/(w0 Q)).
    """

    def __init__(self, *, S0=None, w0=None, Q=None, sigma=None, rho=None,
                 tau=None, eps=1e-5, name=None):
        self.eps = float(eps)
        if name is not None:
            self.name = name
        if w0 is None:
            if rho is None:
                raise ValueError("either w0 or rho must be given")
            w0 = 2.0 * np.pi / float(rho)
        if Q is None:
            if tau is None:
                raise ValueError("either Q or tau must be given")
            Q = 0.5 * float(w0) * float(tau)
        if S0 is None:
            if sigma is None:
                raise ValueError("either S0 or sigma must be given")
            S0 = float(sigma) ** 2 / (float(w0) * float(Q))
        self.S0 = float(S0)
        self.w0 = float(w0)
        self.Q = float(Q)

    def get_coefficients(self):
        S0, w0, Q = self.S0, self.w0, self.Q
        if Q < 0.5:
            # overdamped: two real exponentials
            f = np.sqrt(max(1.0 - 4.0 * Q * Q, self.eps))
            amp = 0.5 * S0 * w0 * Q
            ar = amp * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f])
            cr = 0.5 * w0 / Q * np.array([1.0 - f, 1.0 + f])
            return ar, cr, _EMPTY, _EMPTY, _EMPTY, _EMPTY
        # underdamped (Q == 0.5 lands here with f = sqrt(eps))
        f = np.sqrt(max(4.0 * Q * Q - 1.0, self.eps))
        a = S0 * w0 * Q
        c = 0.5 * w0 / Q
        return (_EMPTY, _EMPTY, np.array([a]), np.array([a / f]),
                np.array([c]), np.array([c * f]))

    def get_psd(self, omega):
        omega = np.asarray(omega, dtype=np.float64)
        w02 = self.w0 ** 2
        w2 = omega ** 2
        return (np.sqrt(2.0 / np.pi) * self.S0 * w02 * w02
                / ((w2 - w02) ** 2 + w2 * w02 / self.Q ** 2))


class TermSum(Term):
    """Sum of terms: coefficient vectors are concatenated in term order."""

    def __init__(self, *terms):
        flat = []
        for t in terms:
            if isinstance(t, TermSum):
                flat.extend(t.terms)
            else:
                flat.append(t)
        self._terms = tuple(flat)

    @property
    def terms(self):
        return self._terms

    def get_coefficients(self):
        if not self._terms:
            return (_EMPTY,) * 6
        if all(type(t) is SHOTerm for t in self._terms):
            return self._sho_coefficients()
        cols = [[_as_vec(v) for v in t.get_coefficients()] for t in self._terms]
        return tuple(np.concatenate([c[k] for c in cols]) for k in range(6))

    def _sho_coefficients(self):
        """A sum of SHO terms only (every kernel the reference builds, core.py:371-373): the formulas of
        :meth:`SHOTerm.get_coefficients` on arrays of the J terms' parameters -- the same operations element
        by element (bit-identical results), without J x 6 small arrays (86 terms: 1 ms, in front of every
        ``compute`` of the reference's default kernel)."""
        S0 = np.array([t.S0 for t in self._terms])
        w0 = np.array([t.w0 for t in self._terms])
        Q = np.array([t.Q for t in self._terms])
        eps = np.array([t.eps for t in self._terms])
        over = Q < 0.5
        So, wo, Qo = S0[over], w0[over], Q[over]
        f = np.sqrt(np.maximum(1.0 - 4.0 * Qo * Qo, eps[over]))
        amp = 0.5 * So * wo * Qo
        ar = np.stack([amp * (1.0 + 1.0 / f), amp * (1.0 - 1.0 / f)], axis=-1).reshape(-1)
        cr = np.stack([0.5 * wo / Qo * (1.0 - f), 0.5 * wo / Qo * (1.0 + f)], axis=-1).reshape(-1)
        Su, wu, Qu = S0[~over], w0[~over], Q[~over]
        f = np.sqrt(np.maximum(4.0 * Qu * Qu - 1.0, eps[~over]))
        a = Su * wu * Qu
        c = 0.5 * wu / Qu
        return ar, cr, a, a / f, c, c * f

    def get_diag_shift(self):
        return float(sum(t.get_diag_shift() for t in self._terms))

    def get_value(self, tau):
        tau = np.asarray(tau, dtype=np.float64)
        k = np.zeros(tau.shape, dtype=np.float64)
        for t in self._terms:
            k = k + t.get_value(tau)
        return k

    def get_psd(self, omega):
        omega = np.asarray(omega, dtype=np.float64)
        p = np.zeros(omega.shape, dtype=np.float64)
        for t in self._terms:
            p = p + t.get_psd(omega)
        return p


class TermConvolution(Term):
    r"""
    Kernel of a process integrated over a boxcar exposure of length ``delta``
    (celerite2 ``TermConvolution(term, delta)``; base class of the reference's
    ``StellarOscillatorKernel`` at /root/reference/gadfly/core.py:336, :394).

    With z = c - i d and A = a - i b per exponential component,

        k_delta(tau) = Re[A I(tau)] / delta^2
        I(tau) = 2 (cosh(z delta) - 1) exp(-z tau) / z^2             tau >= delta
        I(tau) = 2 (delta - tau)/z
                 + (e^{-z(delta-tau)} + e^{-z(delta+tau)} - 2 e^{-z tau})/z^2
                                                                      tau <  delta

    (derivation: k_delta(tau) = delta^-2 int_{-delta}^{delta} (delta-|s|) k(tau+s) ds;
    checked against numerical quadrature in tests/test_terms.py).
    For tau >= delta this is again a celerite kernel with A' = 2A(cosh(z delta)-1)/(z delta)^2,
    which is what :meth:`get_coefficients` returns; the solver therefore sees the
    exact kernel at every lag >= delta plus the exact variance on the diagonal
    (:meth:`get_diag_shift`), exactly as celerite2 does.
    ``cosh/sinh`` overflow for ``c*delta > ~710`` is reproduced, not hidden
    (parity hazard ii).
    """

    def __init__(self, term, delta):
        self.term = term
        self.delta = float(delta)

    # complex amplitudes / rates of the wrapped term: real terms have d = b = 0
    def _complex_form(self):
        ar, cr, ac, bc, cc, dc = (
            _as_vec(v) for v in self.term.get_coefficients()
        )
        A = np.concatenate([ar.astype(np.complex128), ac - 1j * bc])
        z = np.concatenate([cr.astype(np.complex128), cc - 1j * dc])
        return len(ar), A, z

    def get_coefficients(self):
        nr, A, z = self._complex_form()
        zd = z * self.delta
        with np.errstate(over="ignore", invalid="ignore"):
            Ap = 2.0 * A * (np.cosh(zd) - 1.0) / zd ** 2
        return (Ap[:nr].real.copy(), z[:nr].real.copy(),
                Ap[nr:].real.copy(), -Ap[nr:].imag,
                z[nr:].real.copy(), -z[nr:].imag)

    def get_diag_shift(self):
        _, A, z = self._complex_form()
        zd = z * self.delta
        with np.errstate(over="ignore", invalid="ignore"):
            shift = np.sum((2.0 * A * (zd - np.sinh(zd)) / zd ** 2).real)
        return float(shift) + float(self.term.get_diag_shift())

    def get_device_coefficients(self):
        """:meth:`get_coefficients` and :meth:`get_diag_shift` from ONE evaluation of the wrapped term's
        coefficients (the same operations, bit for bit: a chain of single evaluations pays this host algebra per
        step, 0.24 ms for 30 terms when done twice)."""
        nr, A, z = self._complex_form()
        zd = z * self.delta
        with np.errstate(over="ignore", invalid="ignore"):
            Ap = 2.0 * A * (np.cosh(zd) - 1.0) / zd ** 2
            shift = np.sum((2.0 * A * (zd - np.sinh(zd)) / zd ** 2).real)
        return (Ap[:nr].real.copy(), z[:nr].real.copy(), Ap[nr:].real.copy(), -Ap[nr:].imag,
                z[nr:].real.copy(), -z[nr:].imag, float(shift) + float(self.term.get_diag_shift()))

    def get_value(self, tau):
        _, A, z = self._complex_form()
        delta = self.delta
        tau = np.abs(np.asarray(tau, dtype=np.float64))
        t = tau[..., None]
        with np.errstate(over="ignore", invalid="ignore"):
            large = 2.0 * (np.cosh(z * delta) - 1.0) * np.exp(-z * t) / z ** 2
            tm = np.minimum(t, delta)       # keep the unused branch finite
            small = (2.0 * (delta - tm) / z
                     + (np.exp(-z * (delta - tm)) + np.exp(-z * (delta + tm))
                        - 2.0 * np.exp(-z * tm)) / z ** 2)
        I = np.where(t >= delta, large, small)
        return np.sum((A * I).real, axis=-1) / delta ** 2

    def get_psd(self, omega):
        omega = np.asarray(omega, dtype=np.float64)
        arg = 0.5 * self.delta * omega
        # sinc(0) = 1
        sinc = np.sinc(arg / np.pi)
        return self.term.get_psd(omega) * sinc ** 2

    def to_dense(self, x, diag):
        x = np.asarray(x, dtype=np.float64)
        # solver view: transformed coefficients at *every* off-diagonal lag
        K = Term.get_value(self, x[:, None] - x[None, :])
        K[np.diag_indices_from(K)] += (
            np.asarray(diag, dtype=np.float64) + self.get_diag_shift()
        )
        return K
