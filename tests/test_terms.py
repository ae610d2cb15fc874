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
