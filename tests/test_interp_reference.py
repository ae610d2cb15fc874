"""
The restatement of ``interpolate_missing_data`` (oracle/interp_ref.py) against fixtures produced by
the REFERENCE ITSELF (/root/reference/gadfly/interp.py:6-60 run by tests/golden/reference/make_interp_golden.py
in the build container): bit-for-bit.  This is the one part of the oracle that is pinned at the
reference level; the celerite recurrences stay "parity unpinned" (DESIGN.md 3).
The GPU counterpart is tests/test_gpu_interp.py::test_matches_reference_fixtures.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import interp_ref
from tests.interp_cases import CASES, make_case, FULL_ARRAYS_BELOW

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference", "interp_reference.npz")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def check_against_fixture(name, fn):
    """fn(times, fluxes, **kw) -> (times, fluxes) must reproduce the reference's stored output."""
    g = np.load(GOLDEN)
    t, f, kw = make_case(name)
    n_out = int(g[f"{name}/n_out"])
    if n_out < FULL_ARRAYS_BELOW:           # the stored inputs are what the reference was run on
        np.testing.assert_array_equal(t, g[f"{name}/t_in"])
        np.testing.assert_array_equal(f, g[f"{name}/f_in"])
        if kw:
            np.testing.assert_array_equal(kw["cadences"], g[f"{name}/cad_in"])
    tt, ff = fn(t, f, **kw)
    assert len(tt) == n_out
    if n_out < FULL_ARRAYS_BELOW:
        np.testing.assert_array_equal(np.asarray(tt), g[f"{name}/t_out"])
        np.testing.assert_array_equal(np.asarray(ff), g[f"{name}/f_out"])
    assert digest(tt, ff) == str(g[f"{name}/sha256"])


@pytest.mark.parametrize("name", list(CASES))
def test_restatement_matches_reference_output(name):
    check_against_fixture(name, interp_ref.interpolate_missing_data)


def test_drift_cases_really_reorder():
    """The drift fixtures exercise the merge-by-time: some missing cadence's grid time precedes
    the stamp of the point before it in cadence order."""
    for name in ("drift_10_cad", "drift_20000_cad"):
        t, f, kw = make_case(name)
        cad = kw["cadences"]
        dt = np.median(np.diff(t) / np.diff(cad))
        idx = cad - cad[0]
        missing = np.setdiff1d(np.arange(idx.min(), idx.max()), idx)
        grid = t[0] + missing * dt
        prev = t[np.searchsorted(idx, missing) - 1]        # point before it in cadence order
        assert np.any(grid < prev), name
