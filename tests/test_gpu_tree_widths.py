"""
The chunk combines at every width class and slot layout (round 4): the combines' kernels come in three LDS sizes
(32 x 32 matrices for widths <= 32, 48 x 48 up to 48 -- four and two workgroups per CU --, 64 x 64 beyond), their products in unrolled forms for
three and four 16-column tiles and a loop for fewer, the scan runs on the sweeps' own slot arrays whatever the
chunk count (it knows the real one: pairs with a padding right range copy, pairs of padding are not launched, its
slots beyond the real chunks live in a work buffer), it leaves out its top levels, and the sequential combine skips
the first chunk's solve.

Every case: log-likelihood by the two-sweep and the three-sweep route against the oracle's C restatement at 1e-8, the
first failing row of a matrix that is not positive definite, and the stored factor's solve at 1e-6.  Widths
W = 2 J from 6 to 62 (the class boundaries 32 | 34, 48 | 50 included), chunk counts 5 ... 70 per problem.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-8
TOL_VEC = 1e-6


def _ref(kernel, t, diag, y):
    from oracle import cref
    co = kernel.get_device_coefficients()
    v, info = cref.loglike(co[:6], t, diag + co[6], y)
    return v, info


# (J, B, N, chunk_len): chunk counts 36 (tree, padded), 32 (tree, power of two), 70, 5 (sequential combine)
CASES = [(3, 1, 18_000, 512), (8, 1, 18_000, 512), (12, 3, 18_000, 512), (16, 1, 16_384, 512), (17, 2, 18_000, 512),
         (20, 1, 35_500, 512), (20, 3, 16_384, 512), (24, 1, 18_000, 512), (24, 2, 2_500, 512), (25, 1, 18_000, 512),
         (25, 3, 16_384, 512), (31, 1, 18_000, 512), (31, 2, 2_500, 512)]


@pytest.mark.parametrize("J,B,N,L", CASES, ids=[f"W{2 * c[0]}-B{c[1]}-N{c[2]}" for c in CASES])
def test_every_width_class_and_slot_layout(hip, J, B, N, L):
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.synth import jitter_hyperparameters, solar_like_hyperparameters
    prob = util.solar_problem(J, N, seed=100 + J)
    t, y, diag = prob["t"], prob["y"], prob["diag_user"]
    kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 7 * J + i),
                                                  texp=60.0) for i in range(B)]
    refs = np.array([_ref(k, t, diag, y)[0] for k in kernels])
    eng = StreamingBatch([k.get_device_coefficients() for k in kernels], t, y, diag=diag, tile_rows=1024)
    assert eng._fused_ok()
    nch = eng._tp_chunking(L)[1]
    for two in (True, False):
        eng.two_sweep = two
        for _ in range(2):              # (the second evaluation finds the slot sets the first one left behind)
            ll = eng.log_likelihood_time_parallel(chunk_len=L).cpu().numpy()
            assert eng._tp_used and eng._two_sweep_used == two
            assert np.max(np.abs(ll - refs) / np.abs(refs)) <= RTOL_LL, (two, nch, ll, refs)
    # enough chunks for the tree: the scan ran on the sweeps' own slot arrays (no padded copies)
    if nch >= eng.tree_min_chunks:
        assert eng._tp["S"].shape[0] == B * nch and eng._tp["tree"]["X"].shape[0] == B * nch
    # the stored factor (three sweeps with stores) and one solve
    from oracle import cref, seq
    fac = eng.stored_factor(chunk_len=L)
    import torch
    Y = np.random.default_rng(J).normal(size=(N, 2))
    got = fac.apply_inverse(torch.as_tensor(Y).cuda().reshape(1, N, 2).expand(B, N, 2).contiguous())[0].cpu().numpy()
    c, a, U, V = seq.celerite_matrices(kernels[0].get_device_coefficients()[:6], t,
                                       diag + kernels[0].get_device_coefficients()[6])
    d_ref, W_ref, info = cref.factor(t, c, a, U, V)
    assert info == 0
    want = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y) / d_ref[:, None])
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < TOL_VEC


@pytest.mark.parametrize("J,B", [(20, 1), (20, 2), (30, 1)], ids=["W40-B1", "W40-B2", "W60-B1"])
def test_first_failing_row_through_the_combines(hip, J, B):
    """A diagonal that turns the matrix indefinite from one row on: -inf and celerite2's failing row (the first
    non-positive pivot), whichever combine stitched the chunks."""
    import gadfly_amd
    N, L, bad = 18_000, 512, 11_111
    prob = util.solar_problem(J, N, seed=5)
    k = prob["kernel"]
    diag = prob["diag_user"].copy()
    diag[bad:] = -2.0 * k.get_value(np.zeros(1))[0]
    _, info_ref = _ref(k, prob["t"], diag, prob["y"])
    assert info_ref == bad + 1
    ev = gadfly_amd.BatchedLogLikelihood([k] * B, prob["t"], prob["y"], diag=diag)
    ev.engine._tp_chunking = (lambda chunk_len, store=False, _o=ev.engine._tp_chunking: _o(L, store))
    got = ev.evaluate()
    assert np.all(got == -np.inf) and np.all(ev.engine.info.cpu().numpy() == bad + 1)


@pytest.mark.parametrize("R", [16, 40, 70], ids=["R16", "R40", "R70"])
def test_many_right_hand_sides_through_the_segment_scans(hip, R):
    """Solves with 16 or more right-hand sides stitch their chunks 64 right-hand sides at a time on the matrix pipe
    (k_lincombine_R): R = 16 (one tile), 40 (a ragged tile), 70 (two tiles), forward and backward sweep, against
    the oracle's recurrences at 1e-6; two problems, 79 chunks each."""
    import torch
    import gadfly_amd
    from gadfly_amd.engine import StreamingBatch
    from gadfly_amd.synth import jitter_hyperparameters, solar_like_hyperparameters
    from oracle import cref, seq
    J, N, L = 30, 40_000, 512
    prob = util.solar_problem(J, N, seed=77)
    t, diag = prob["t"], prob["diag_user"]
    kernels = [gadfly_amd.StellarOscillatorKernel(jitter_hyperparameters(solar_like_hyperparameters(J), 900 + i),
                                                  texp=60.0) for i in range(2)]
    eng = StreamingBatch([k.get_device_coefficients() for k in kernels], t, prob["y"], diag=diag, tile_rows=1024)
    fac = eng.stored_factor(chunk_len=L)
    assert fac._segments() is not None          # (the two-level combine, not the plain scan)
    Y = np.random.default_rng(R).normal(size=(2, N, R))
    got = fac.apply_inverse(torch.as_tensor(Y).cuda()).cpu().numpy()
    for b, k in enumerate(kernels):
        co = k.get_device_coefficients()
        c, a, U, V = seq.celerite_matrices(co[:6], t, diag + co[6])
        d_ref, W_ref, info = cref.factor(t, c, a, U, V)
        assert info == 0
        want = cref.solve_upper(t, c, U, W_ref, cref.solve_lower(t, c, U, W_ref, Y[b]) / d_ref[:, None])
        assert np.max(np.abs(got[b] - want)) / np.max(np.abs(want)) < TOL_VEC, (b, R)
