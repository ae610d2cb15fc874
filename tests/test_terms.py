"""CPU: host-side term algebra and the kernel classes' API surface."""
import warnings

import numpy as np
import pytest
from scipy.integrate import quad

import gadfly_amd
from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
from gadfly_amd.synth import solar_like_hyperparameters


def test_sho_coefficients_branches():
    ar, cr, ac, bc, cc, dc = SHOTerm(S0=2.0, w0=1.5, Q=0.3).get_coefficients()
    assert len(ar) == 2 and len(ac) == 0
    f = np.sqrt(1 - 4 * 0.09)
    np.testing.assert_allclose(ar, 0.5 * 2.0 * 1.5 * 0.3 * np.array([1 + 1 / f, 1 - 1 / f]))
    np.testing.assert_allclose(cr, 0.5 * 1.5 / 0.3 * np.array([1 - f, 1 + f]))
    # Q = 0.5 takes the underdamped branch with f = sqrt(eps)
    ar, cr, ac, bc, cc, dc = SHOTerm(S0=1.0, w0=2.0, Q=0.5).get_coefficients()
    assert len(ar) == 0 and len(ac) == 1
    np.testing.assert_allclose(dc, cc * np.sqrt(1e-5))
    # sigma / rho / tau parameterisation
    t1 = SHOTerm(sigma=1.3, rho=2.0, tau=5.0)
    assert np.isclose(t1.w0, np.pi) and np.isclose(t1.Q, 0.5 * np.pi * 5.0)
    assert np.isclose(t1.S0, 1.3 ** 2 / (t1.w0 * t1.Q))


@pytest.mark.parametrize("Q", [0.3, 0.6, 5.0, 1500.0])
def test_psd_is_fourier_pair_of_kernel(Q):
    """generic coefficient PSD == closed-form SHO PSD (reference core.py:33-41)."""
    term = SHOTerm(S0=0.7, w0=3.0, Q=Q)
    w = np.linspace(0.01, 12, 200)
    from gadfly_amd.terms import Term
    np.testing.assert_allclose(Term.get_psd(term, w), term.get_psd(w), rtol=1e-4 if Q < 0.5 else 1e-11)
    from gadfly_amd.core import _sho_psd
    np.testing.assert_allclose(term.get_psd(w), _sho_psd(w, 0.7, 3.0, Q), rtol=1e-14)


def test_term_convolution_against_quadrature():
    base = TermSum(SHOTerm(S0=1.0, w0=3.0, Q=5.0), SHOTerm(S0=0.5, w0=1.0, Q=0.3))
    delta = 0.05
    k = TermConvolution(base, delta)
    for tau in [0.0, 0.013, 0.05, 0.0731, 0.4, 2.3]:
        f = lambda s: (delta - abs(s)) * base.get_value(np.array(tau + s))   # noqa: E731
        val = quad(f, -delta, delta, points=[0.0, -tau] if tau < delta else None,
                   epsabs=1e-14, epsrel=1e-13)[0] / delta ** 2
        assert abs(k.get_value(np.array(tau)) - val) < 1e-11 * abs(val)
    # solver view: transformed coefficients + diagonal shift reproduce k(0) and k(tau >= delta)
    K = k.to_dense(np.array([0.0, delta, 1.0]), np.zeros(3))
    assert abs(K[0, 0] - k.get_value(np.zeros(1))[0]) < 1e-12 * K[0, 0]
    assert abs(K[0, 2] - k.get_value(np.array(1.0))) < 1e-12 * abs(K[0, 2])
    # PSD picks up sinc^2
    w = np.array([0.0, 1.0, 40.0])
    np.testing.assert_allclose(k.get_psd(w), base.get_psd(w) * np.sinc(0.5 * delta * w / np.pi) ** 2)


def test_term_convolution_overflow_is_reproduced_not_hidden():
    """cosh(c delta) overflows for c delta > ~710 (ShotNoiseKernel w0 = 1e7 with a long exposure,
    SURVEY.md section 7 hazard ii): celerite2's formulas give inf/nan and so do these."""
    k = gadfly_amd.StellarOscillatorKernel(
        terms=[gadfly_amd.ShotNoiseKernel(S0=1.0, w0=1e7, Q=0.5)], delta=100e-6)
    co = k.get_coefficients()
    assert not np.all(np.isfinite(co[2]))


def test_kernel_api_surface():
    hp = solar_like_hyperparameters(6)
    assert isinstance(hp, gadfly_amd.Hyperparameters) and len(hp) == 6
    assert "showing 1 of 6" in repr(hp)
    k = gadfly_amd.StellarOscillatorKernel(hp, texp=60.0)
    assert k.name == hp.name and k.hyperparameters is hp
    assert np.isclose(k.delta, 60e-6) and len(k.term.terms) == 6 and len(k) == 12
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        k0 = gadfly_amd.StellarOscillatorKernel(hp)
        assert np.isclose(k0.delta, 60e-6) and any("exposure time" in str(x.message) for x in w)
    # __add__ bookkeeping (reference core.py:405-427)
    k2 = k + gadfly_amd.ShotNoiseKernel(S0=1e-3, w0=1e5, Q=0.5)
    assert k2.name == hp.name + " + Shot noise" and len(k2.term.terms) == 7 and k2.delta == k.delta
    assert gadfly_amd.ShotNoiseKernel.w0 == 1e7 and gadfly_amd.ShotNoiseKernel.Q == 0.5
    # names: a nameless kernel takes the other's name; an unnamed term adds none (core.py:414-422)
    kn = gadfly_amd.StellarOscillatorKernel(gadfly_amd.Hyperparameters(list(hp)), texp=60.0)
    assert kn.name is None
    assert (kn + gadfly_amd.ShotNoiseKernel(S0=1e-3, w0=1e5, Q=0.5, name="Kepler")).name == "Kepler"
    k3 = k2 + gadfly_amd.ShotNoiseKernel(S0=2e-3, w0=1e5, Q=0.5, name="TESS")
    assert k3.name == hp.name + " + Shot noise + TESS" and len(k3.term.terms) == 8
    # the sum is a new kernel over the same term objects, coefficients concatenated in order (TermSum)
    assert k2.term.terms[:6] == k.term.terms and k2.get_coefficients()[2].shape == (7,)
    # a LIST is refused exactly as the reference refuses it: its list branch reads `other.terms` (core.py:412)
    with pytest.raises(AttributeError, match="terms"):
        k + [gadfly_amd.ShotNoiseKernel(S0=1e-3, w0=1e5, Q=0.5)]
    # from_soho_virgo returns the raw 9-entry fit whose oscillation entries lack w0
    raw = gadfly_amd.Hyperparameters.from_soho_virgo()
    assert len(raw) == 9 and raw.name == "SOHO VIRGO/PMO6"
    assert "w0" not in raw[5]["hyperparameters"]
    with pytest.raises(ValueError, match="tynt"):          # default Kepler bandpass needs tynt
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gadfly_amd.SolarOscillatorKernel(texp=60.0)
    lo, up = gadfly_amd.ShotNoiseKernel.kepler_mag_to_noise_amplitude(12.0)
    assert 0 < lo < up


def test_get_psd_matches_sum_of_sho_with_exposure():
    hp = solar_like_hyperparameters(8)
    k = gadfly_amd.StellarOscillatorKernel(hp, texp=60.0)
    from gadfly_amd.core import _sho_psd
    w = 2 * np.pi * np.linspace(3.0, 4000.0, 300)
    ref = sum(_sho_psd(w, **p["hyperparameters"]) for p in hp) * np.sinc(0.5 * k.delta * w / np.pi) ** 2
    np.testing.assert_allclose(k.get_psd(w), ref, rtol=1e-12)


def test_scaling_relations_to_solar():
    """reference gadfly/tests/test_core.py:54-66"""
    from gadfly_amd import scale
    ones = np.array([scale.amplitude_with_wavelength("SOHO VIRGO", 5777.0),
                     scale.nu_max(1.0, 5777.0, 1.0), scale.delta_nu(1.0, 1.0),
                     scale.tau_gran(1.0, 5777.0, 1.0), scale.granulation_amplitude(1.0, 5777.0, 1.0),
                     scale.p_mode_amplitudes(1.0, 5777.0, 1.0), scale.tau_eff(3090.0)])
    np.testing.assert_allclose(ones, np.ones_like(ones))
    assert scale.amplitude_with_wavelength("SOHO VIRGO", 5777.0) == 1      # test_core.py:70
    # a Kepler-like top hat (0.43-0.89 micron) lands near the tynt Kepler value 2.795 (test_core.py:71-73)
    wl = np.linspace(0.3, 1.1, 400)
    alpha = scale.amplitude_with_wavelength((wl, ((wl > 0.43) & (wl < 0.89)).astype(float)), 5777.0)
    assert 2.4 < alpha < 3.2
    assert np.isclose(scale.c_K(5934.0), 1.0)
    # the Kiefer envelope is 1 at nu_max by construction and decays away from it
    env = scale.p_mode_intensity(5777.0, np.array([3090.0, 2000.0, 4500.0]), 3090.0, 135.1)
    assert np.isclose(env[0], 1.0) and 0 < env[1] < 1 and 0 < env[2] < 1


def test_solar_kernel_for_star():
    """SolarOscillatorKernel = for_star at solar parameters (reference core.py:430-461):
    5 granulation terms verbatim + 81 Broomhall modes with the solar-case reduction."""
    import json
    from gadfly_amd.core import default_hyperparameter_path, _sho_psd
    k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
    hp = k.hyperparameters
    assert len(hp) == 86 and len(k) == 172
    raw = json.load(open(default_hyperparameter_path))
    for got, want in zip(hp[:5], raw[:5]):
        assert got["hyperparameters"] == want["hyperparameters"]
    from gadfly_amd.scale import broomhall_p_mode_freqs
    nu, ell = broomhall_p_mode_freqs()
    osc = {r["metadata"]["degree"]: r["hyperparameters"] for r in raw[5:]}
    for j, p in enumerate(hp[5:]):
        h, o = p["hyperparameters"], osc[int(ell[j])]
        assert np.isclose(h["w0"], 2 * np.pi * nu[j])
        Gamma = nu[j] / (2 * o["Q"])
        assert np.isclose(h["Q"], o["Q"] * 1.02 / Gamma)
        assert h["S0"] > 0 and p["metadata"]["degree"] == int(ell[j])
    # the strongest modes sit near nu_max (envelope)
    S0 = np.array([p["hyperparameters"]["S0"] * p["hyperparameters"]["Q"] ** 2 for p in hp[5:]])
    assert abs(nu[np.argmax(S0)] - 3090.0) < 400.0
    # other stars scale nu_max
    hp2 = gadfly_amd.Hyperparameters.for_star(1.2, 1.5, 6000.0, 3.0, bandpass="SOHO VIRGO", name="x")
    assert hp2.name == "x" and len(hp2) == 86
    w = np.array([p["hyperparameters"]["w0"] for p in hp2[5:]])
    from gadfly_amd import scale
    assert np.isclose(np.median(w) / np.median(2 * np.pi * nu),
                      scale.nu_max(1.2, 6000.0, 1.5), rtol=0.1)


def test_vectorised_sho_pack_equals_per_kernel_path():
    """gadfly_amd.batch.sho_coefficient_pack (the term algebra of B kernels at once)
    gives the numbers of the per-object path, overdamped and Q = 1/2 terms included."""
    from gadfly_amd.batch import sho_coefficient_pack
    from gadfly_amd.engine import _coeff_pack
    from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution
    rng = np.random.default_rng(0)
    B, J = 16, 9
    S0 = np.exp(rng.uniform(-2, 4, (B, J)))
    w0 = np.exp(rng.uniform(0, 8, (B, J)))
    Q = np.exp(rng.uniform(np.log(0.5), 5, (B, J)))
    Q[:, 2] = rng.uniform(0.05, 0.45, B)
    Q[:, 7] = rng.uniform(0.05, 0.45, B)
    Q[:, 4] = 0.5
    delta = 60e-6
    ks = [TermConvolution(TermSum(*[SHOTerm(S0=S0[b, j], w0=w0[b, j], Q=Q[b, j]) for j in range(J)]), delta)
          for b in range(B)]
    ref = _coeff_pack([k.get_device_coefficients() for k in ks])
    got = sho_coefficient_pack(S0, w0, Q, delta)
    assert got[:2] == ref[:2] == (4, 7)
    for a, b in zip(ref[2:], got[2:]):
        np.testing.assert_array_equal(a, b)
    Q[3, 0] = 0.3                      # a different overdamped pattern in one problem
    with pytest.raises(ValueError):
        sho_coefficient_pack(S0, w0, Q, delta)


@pytest.mark.parametrize("example_freq", [3033.886, 3082.471, 3098.327, 3160.028, 3168.773, 3217.916])
def test_broomhall_known_frequencies(example_freq):
    """The reference's own known answers for the p-mode table (/root/reference/gadfly/tests/
    test_sun.py:7-19): each example frequency is the table entry nearest to it."""
    from gadfly_amd.synth import broomhall_modes
    nu, ell = broomhall_modes()
    assert len(nu) == len(ell) == 81 and set(ell) <= {0, 1, 2, 3}
    closest = nu[np.argmin(np.abs(nu - example_freq))]
    np.testing.assert_allclose(example_freq, closest)


# ---------------------------------------------------------------------------------------------
# pinned to the reference: outputs of /root/reference/gadfly/core.py's own statements
# (tests/golden/reference/make_core_golden.py ran them in the build container)
# ---------------------------------------------------------------------------------------------
def _core_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "reference", "core_functions.npz"))


def test_sho_psd_matches_the_reference_function():
    """`_sho_psd` (reference core.py:33-41) -- the product's copy, SHOTerm.get_psd, the generic coefficient
    PSD of the celerite form, and the oracle's restatement all reproduce the reference's outputs."""
    from gadfly_amd.core import _sho_psd
    from gadfly_amd.terms import Term
    from oracle import terms_ref
    fx = _core_fixture()
    omega, want = fx["omega"], fx["sho_psd"]
    for (S0, w0, Q), ref in zip(fx["sho_params"], want):
        np.testing.assert_allclose(_sho_psd(omega, S0, w0, Q), ref, rtol=1e-15, atol=0.0)
        np.testing.assert_allclose(SHOTerm(S0=S0, w0=w0, Q=Q).get_psd(omega), ref, rtol=4e-15, atol=0.0)
        np.testing.assert_allclose(terms_ref.sho_psd(omega, S0, w0, Q), ref, rtol=1e-15, atol=0.0)
        # the PSD computed from the celerite coefficients (what a TermSum / TermConvolution with delta = 0
        # evaluates): exact for Q >= 1/2 up to the cancellation in w^4 + 2 (c^2 - d^2) w^2 + (c^2 + d^2)^2
        # (at Q = 1/2 celerite2's SHOTerm regularises f = sqrt(max(4 Q^2 - 1, eps = 1e-5)): O(eps) in the PSD)
        if Q >= 0.5:
            got = Term.get_psd(SHOTerm(S0=S0, w0=w0, Q=Q), omega)
            # (and the coefficient form cancels in its numerator far above w0: compared up to 30 w0)
            rtol = 1e-4 if 4 * Q * Q - 1 < 1e-3 else 1e-7 * max(1.0, Q)
            near = omega <= 30.0 * w0
            np.testing.assert_allclose(got[near], ref[near], rtol=rtol, atol=0.0)


def test_kepler_noise_amplitude_matches_the_reference_statements():
    """ShotNoiseKernel.kepler_mag_to_noise_amplitude (reference core.py:522-544: Jenkins et al. 2010): the
    fixture holds sigma_lower, sigma_upper from the reference's own assignment statements; the function
    returns 1e6 [sigma_lower, sigma_upper] (in ppm^2: the reference tags the unit on its return line)."""
    fx = _core_fixture()
    for m, sig in zip(fx["kepler_mag"], fx["sigma"]):
        got = gadfly_amd.ShotNoiseKernel.kepler_mag_to_noise_amplitude(float(m))
        np.testing.assert_allclose(np.asarray(got, dtype=float), 1e6 * sig, rtol=1e-15, atol=0.0)
    got = gadfly_amd.ShotNoiseKernel.kepler_mag_to_noise_amplitude(fx["kepler_mag"])
    np.testing.assert_allclose(np.asarray(got[0], dtype=float), 1e6 * fx["sigma_array_form"][:, 0], rtol=1e-15)
    np.testing.assert_allclose(np.broadcast_to(got[1], fx["kepler_mag"].shape),
                               1e6 * fx["sigma_array_form"][:, 1], rtol=1e-15)


# ---------------------------------------------------------------------------------------------
# the product's coefficient algebra against the oracle's independent restatement (oracle/terms_ref.py:
# real arithmetic) -- the golden vectors take their coefficient inputs from the oracle
# ---------------------------------------------------------------------------------------------
def _against_oracle(kernel, triples, delta):
    from oracle import terms_ref
    got = kernel.get_device_coefficients()
    want = terms_ref.kernel_coefficients(triples, delta)
    # exposure integration: both formulations take cosh(z delta) - 1 at |z delta| << 1 (the real form as
    # celerite2 writes it, the product's in complex arithmetic): each is good to ~eps / |z delta|^2, which is
    # what they may differ by -- 1e-10 for the slowest granulation term at a one-minute exposure
    raw = terms_ref.sum_coefficients(triples)
    eps = 2.3e-16

    def tol(x2):                        # per term: eps / |z delta|^2, at least a few ulps
        return np.maximum(2e-13, 20 * eps / np.minimum(x2, 1.0)) if delta is not None else np.full(x2.shape, 2e-13)

    tol_r, tol_c = tol((raw[1] * (delta or 1.0)) ** 2), tol((raw[4] ** 2 + raw[5] ** 2) * (delta or 1.0) ** 2)
    for k, (g, w) in enumerate(zip(got[:6], want[:6])):
        assert np.shape(g) == np.shape(w)
        if np.size(w):
            rt = tol_r if k < 2 else tol_c
            # (a' and b' of a complex term are compared on the scale of the pair: either may nearly cancel)
            scale = np.abs(w) + ((np.abs(want[2]) + np.abs(want[3])) if k in (2, 3) else 0.0)
            assert np.all(np.abs(g - w) <= rt * scale), (k, g, w)
    # the diagonal correction is a difference of nearly equal terms (x - sinh x at x = c delta << 1): both
    # formulations carry ~1e-9 of ITSELF, i.e. ~1e-12 of the variance k(0) it corrects
    raw0 = terms_ref.sum_coefficients(triples)
    k0 = float(np.sum(np.abs(raw0[0])) + np.sum(np.abs(raw0[2])))
    assert abs(got[6] - want[6]) <= 1e-11 * k0 + 1e-13 * abs(want[6])


@pytest.mark.parametrize("J", [1, 5, 6, 20, 30, 40, 86])
def test_stellar_kernel_coefficients_against_oracle(J):
    hp = solar_like_hyperparameters(min(J, 40)) if J < 86 else None
    if hp is None:
        k = gadfly_amd.SolarOscillatorKernel(texp=60.0, bandpass="SOHO VIRGO")
        hp = k.hyperparameters
    else:
        k = gadfly_amd.StellarOscillatorKernel(hp, texp=60.0)
    triples = [tuple(p["hyperparameters"][q] for q in ("S0", "w0", "Q")) for p in hp]
    _against_oracle(k, triples, k.delta)


def test_random_and_edge_kernels_against_oracle():
    rng = np.random.default_rng(5)
    cases = [[(1.0, 2 * np.pi * 3.0, 100.0)], [(2.0, 1.5, 0.3)], [(1.0, 2.0, 0.5)],
             [(1.0, 3.0, 5.0), (0.5, 1.0, 0.3), (2.0, 0.7, 0.6)]]
    for _ in range(20):
        n = int(rng.integers(1, 8))
        cases.append([(10 ** rng.uniform(-3, 2), 10 ** rng.uniform(-1, 3), 10 ** rng.uniform(-0.6, 3))
                      for _ in range(n)])
    for triples in cases:
        # (overdamped terms come first in the oracle's concatenation only if they do in the product's: both
        # concatenate per term, real parts and complex parts separately, in term order)
        for delta in (None, 0.01, 0.3):
            base = TermSum(*[SHOTerm(S0=a, w0=b, Q=c) for a, b, c in triples])
            k = base if delta is None else TermConvolution(base, delta)
            _against_oracle(k, triples, delta)


def test_golden_inputs_come_from_the_oracle():
    """The committed golden vectors carry the oracle's coefficients (tests/golden/make_golden.py), so the HIP
    path is checked on inputs the product did not compute."""
    import os
    from tests import util
    from tests.golden.make_golden import CASES, oracle_coefficients
    for name, (kind, kw) in CASES.items():
        prob = util.solar_problem(**kw) if kind == "solar" else util.generic_problem(**kw)
        want = oracle_coefficients(prob["kernel"])
        fx = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
        for key, w in zip(("ar", "cr", "ac", "bc", "cc", "dc"), want[:6]):
            assert np.array_equal(fx[key], w), (name, key)
        assert float(fx["diag_shift"]) == want[6]
