"""gf_dense_solve: the batched Gauss-Jordan solve behind the time-parallel combine of wide kernels
(dense W x W chunk maps).  Checked against LAPACK (numpy.linalg.solve) on the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(hip, A, B):
    import torch
    Ad = torch.as_tensor(np.ascontiguousarray(A)).cuda()
    Bd = torch.as_tensor(np.ascontiguousarray(B)).cuda()
    rc = hip.load().gf_dense_solve(A.shape[0], A.shape[1], B.shape[2], hip.ptr(Ad), hip.ptr(Bd), None)
    hip.check(rc, "gf_dense_solve")
    torch.cuda.synchronize()
    assert np.array_equal(Ad.cpu().numpy(), A, equal_nan=True)      # A is read only
    return Bd.cpu().numpy()


@pytest.mark.parametrize("n,nrhs", [(1, 1), (5, 3), (64, 64), (65, 7), (96, 193), (100, 1), (128, 129),
                                    (160, 321), (172, 345), (172, 173), (176, 50), (177, 49), (192, 385)])
def test_dense_solve_random(hip, n, nrhs):
    rng = np.random.default_rng(1000 * n + nrhs)
    batch = 3
    A = rng.normal(size=(batch, n, n))                  # no diagonal dominance: pivoting is needed
    A[1] = np.eye(n)[rng.permutation(n)] + 0.05 * rng.normal(size=(n, n))     # zero-ish diagonal
    B = rng.normal(size=(batch, n, nrhs))
    X = _solve(hip, A, B)
    ref = np.linalg.solve(A, B)
    for b in range(batch):
        scale = np.abs(ref[b]).max()
        cond = np.linalg.cond(A[b])
        assert np.abs(X[b] - ref[b]).max() <= 1e-13 * cond * scale + 1e-13 * scale, (b, cond)
        # backward error: the residual is at rounding level whatever the conditioning
        res = np.abs(A[b] @ X[b] - B[b]).max()
        assert res <= 1e-11 * (np.abs(A[b]).max() * scale * n), (b, res)


def test_dense_solve_map_like_systems(hip):
    """I - X G with symmetric X, G (what the chunk combine solves), many systems per launch."""
    rng = np.random.default_rng(7)
    n, batch = 172, 40
    L = rng.normal(size=(batch, n, 12))
    X = L @ L.transpose(0, 2, 1) / 12.0
    M = rng.normal(size=(batch, n, 9))
    G = -(M @ M.transpose(0, 2, 1)) / 9.0
    A = np.eye(n) - X @ G
    B = np.concatenate([rng.normal(size=(batch, n, n)), X, rng.normal(size=(batch, n, 1))], axis=2)
    got = _solve(hip, A, B)
    ref = np.linalg.solve(A, B)
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


def test_dense_solve_singular_entries_do_not_disturb_the_others(hip):
    rng = np.random.default_rng(3)
    n, nrhs = 130, 40
    A = rng.normal(size=(4, n, n))
    A[1] = 0.0                                          # singular
    A[2, :, 5] = np.nan                                 # garbage map of a failed chunk
    B = rng.normal(size=(4, n, nrhs))
    X = _solve(hip, A, B)
    for b in (0, 3):
        ref = np.linalg.solve(A[b], B[b])
        assert np.abs(X[b] - ref).max() <= 1e-9 * np.abs(ref).max()


def test_dense_solve_rejects_bad_sizes(hip):
    import torch
    A = torch.zeros((1, 193, 193), dtype=torch.float64, device="cuda")
    B = torch.zeros((1, 193, 2), dtype=torch.float64, device="cuda")
    assert hip.load().gf_dense_solve(1, 193, 2, hip.ptr(A), hip.ptr(B), None) != 0
    assert "unsupported" in hip.last_error()
    assert hip.load().gf_dense_solve(1, 0, 2, hip.ptr(A), hip.ptr(B), None) != 0


@pytest.mark.parametrize("nch,W", [(11, 70), (16, 172), (3, 96)])
def test_lft_tree_scan_against_sequential_application(hip, nch, W):
    """The Blelloch scan over dense chunk maps (engine._lft_tree_scan: batched GEMMs on strided views +
    gf_dense_solve, identity maps padding to a power of two) gives the start state of every chunk that
    applying the maps one after the other gives; the captured-graph replay returns the same numbers."""
    import torch
    from gadfly_amd.engine import _lft_tree_scan, _lft_apply, _TreeScanGraph
    g = torch.Generator(device="cuda").manual_seed(100 * nch + W)
    kw = dict(dtype=torch.float64, device="cuda", generator=g)
    Ph = 0.7 * torch.randn((nch, W, W), **kw) / W ** 0.5
    L = torch.randn((nch, W, 6), **kw)
    Xb = L @ L.transpose(1, 2) / 6
    M = torch.randn((nch, W, 6), **kw)
    G = -(M @ M.transpose(1, 2)) / 6
    Yb = torch.randn((nch, W), **kw)
    m = torch.randn((nch, W), **kw)
    Xs, Ys = _lft_tree_scan(torch, Ph, G, Xb, Yb, m)
    X = torch.zeros((1, W, W), dtype=torch.float64, device="cuda")
    Y = torch.zeros((1, W), dtype=torch.float64, device="cuda")
    for c in range(nch):
        sx = float(X.abs().max()) + 1e-300
        assert float((Xs[c] - X[0]).abs().max()) <= 1e-9 * max(sx, 1.0), c
        assert float((Ys[c] - Y[0]).abs().max()) <= 1e-9 * max(float(Y.abs().max()), 1.0), c
        X, Y = _lft_apply(torch, [a[c:c + 1] for a in (Ph, G, Xb, Yb, m)], X, Y)
    Xg, Yg = _TreeScanGraph.run(torch, Ph, G, Xb, Yb, m)
    assert not _TreeScanGraph.disabled
    assert torch.equal(Xg, Xs) and torch.equal(Yg, Ys)
