"""
Host logic of the ragged batches (gadfly_amd/batch.py: lists of series of different lengths become rectangular arrays
with missing-data rows at the end) and of the two-sweep gate -- no GPU.  What the padding must guarantee is checked
here in plain numpy on the oracle's recurrence: a row with the diagonal 2^1000 leaves the state, the other rows'
pivots and z^2 / d sums untouched and adds exactly log 2^1000 to sum log d.
"""
import numpy as np
import pytest

from gadfly_amd import batch
from gadfly_amd.terms import SHOTerm, TermSum, TermConvolution


def test_is_ragged_and_padding_shapes():
    t = [np.arange(5) * 1.0, np.arange(3) * 2.0 + 1.0, np.arange(1) + 7.0]
    y = [np.ones(5), 2 * np.ones(3), 3 * np.ones(1)]
    assert batch._is_ragged(t) and not batch._is_ragged(np.zeros((2, 3))) and not batch._is_ragged(np.zeros(3))
    assert not batch._is_ragged([np.zeros(4), np.zeros(4)])             # equal lengths: an ordinary (B, N) array
    T, Y, D, rows, dmax, dmin_real, dt_min, tabs = batch._pad_ragged(t, y, [np.full(5, 4.0), 9.0, np.array([1.0])],
                                                                      mean=[0.5, 0.0, 1.0])
    assert T.shape == Y.shape == D.shape == (3, 5) and list(rows) == [5, 3, 1]
    np.testing.assert_array_equal(T[1], [1.0, 3.0, 5.0, 7.0, 9.0])      # the series' own cadence continued
    np.testing.assert_array_equal(T[2], [7.0, 8.0, 9.0, 10.0, 11.0])    # one row: unit spacing
    np.testing.assert_array_equal(Y[0], 0.5 * np.ones(5))               # y - mean
    np.testing.assert_array_equal(Y[1], [2.0, 2.0, 2.0, 0.0, 0.0])
    np.testing.assert_array_equal(Y[2], [2.0, 0.0, 0.0, 0.0, 0.0])
    assert np.all(D[1, 3:] == batch.PAD_DIAG) and np.all(D[1, :3] == 9.0) and np.all(D[0] == 4.0)
    assert np.all(np.diff(T, axis=1) > 0)
    np.testing.assert_array_equal(dmax, [4.0, 9.0, 1.0])
    assert dmin_real == 1.0 and dt_min == 1.0 and tabs == 7.0
    assert batch.PAD_DIAG == 2.0 ** 1000 and abs(np.log(batch.PAD_DIAG) - 1000 * np.log(2.0)) < 1e-12


def test_padding_errors():
    t = [np.arange(5) * 1.0, np.arange(3) * 1.0]
    y = [np.ones(5), np.ones(3)]
    with pytest.raises(ValueError, match="dimension mismatch"):
        batch._pad_ragged(t, [np.ones(5), np.ones(4)], None, 0.0)
    with pytest.raises(ValueError, match="dimension mismatch"):
        batch._pad_ragged(t, np.ones((2, 5)), None, 0.0)
    with pytest.raises(ValueError, match="sorted"):
        batch._pad_ragged([t[0][::-1], t[1]], y, None, 0.0)
    with pytest.raises(ValueError, match="dimension mismatch"):
        batch._pad_ragged(t, y, np.ones((2, 5)), 0.0)                   # a scalar, one value per problem, or a list
    with pytest.raises(ValueError, match="dimension mismatch"):
        batch._pad_ragged(t, y, [np.ones(5)], 0.0)


def test_a_missing_data_row_is_invisible_to_the_recurrence():
    """celerite's recurrence (oracle/seq.py) on a series with and without rows of diagonal 2^1000 at the end and in
    the middle: same pivots and z for the real rows, pivot 2^1000 exactly at the pad rows, z^2 / d of order 2^-1000 there."""
    from oracle import seq
    rng = np.random.default_rng(3)
    k = TermConvolution(TermSum(SHOTerm(S0=3.0, w0=40.0, Q=2.0), SHOTerm(S0=1.0, w0=900.0, Q=30.0),
                                SHOTerm(S0=5.0, w0=3.0, Q=0.3)), 1e-3)
    N = 300
    t = np.arange(N) * 2e-3
    y = rng.normal(size=N)
    co = k.get_device_coefficients()
    pad = np.zeros(N, bool)
    pad[250:] = True
    pad[100:103] = True                                                 # (missing data in the middle works too)
    diag = np.where(pad, batch.PAD_DIAG, 0.04) + co[6]
    c, a, U, V = seq.celerite_matrices(co[:6], t, diag)
    d, Wm, info = seq.factor(t, c, a, U, V)
    z = seq.solve_lower(t, c, U, Wm, np.where(pad, 0.0, y))
    assert info == 0 and np.all(d[pad] == batch.PAD_DIAG)
    assert np.all(z[pad] ** 2 / d[pad] < 1e-290)               # (2^-1000 of an O(1) number: nothing at double precision)
    keep = ~pad
    c2, a2, U2, V2 = seq.celerite_matrices(co[:6], t[keep], np.full(keep.sum(), 0.04) + co[6])
    d2, W2, info2 = seq.factor(t[keep], c2, a2, U2, V2)
    z2 = seq.solve_lower(t[keep], c2, U2, W2, y[keep])
    assert info2 == 0
    np.testing.assert_allclose(d[keep], d2, rtol=1e-12)
    np.testing.assert_allclose(z[keep], z2, rtol=1e-9, atol=1e-12)
    ll_pad = -0.5 * (np.sum(np.log(d)) + N * np.log(2 * np.pi)) - 0.5 * np.sum(z * z / d)
    ll_real = -0.5 * (np.sum(np.log(d2)) + keep.sum() * np.log(2 * np.pi)) - 0.5 * np.sum(z2 * z2 / d2)
    corr = 0.5 * pad.sum() * (np.log(batch.PAD_DIAG) + np.log(2 * np.pi))
    assert abs((ll_pad + corr) - ll_real) <= 1e-11 * abs(ll_real)


def test_two_sweep_gate():
    """SHO terms with positive parameters; an exposure-integrated kernel only while the stamps resolve the exposure
    (up to EXPOSURE_SLACK and the rounding of the stamps)."""
    sho = TermSum(SHOTerm(S0=1.0, w0=2.0, Q=0.7), SHOTerm(S0=2.0, w0=5.0, Q=3.0))
    assert batch._sho_only(sho) and batch._sho_only(sho, None)
    assert not batch._sho_only(TermSum(SHOTerm(S0=-1.0, w0=2.0, Q=0.7)))
    conv = TermConvolution(sho, 1.0)
    assert not batch._sho_only(conv) and not batch._sho_only(conv, None)       # spacing unknown
    assert batch._sho_only(conv, 1.0) and batch._sho_only(conv, 0.95) and not batch._sho_only(conv, 0.2)
    assert batch._exposure_resolved(0.0, None) and batch._exposure_resolved(60e-6, 60e-6 - 2e-11, 2.12e5)
    assert not batch._exposure_resolved(60e-6, 50e-6, 2.12e5)
