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


def _padded(a, WP):
    """(n, W[, W]) -> zero-padded (n, WP[, WP])"""
    out = np.zeros((a.shape[0],) + (WP,) * (a.ndim - 1))
    out[(slice(None),) + tuple(slice(0, k) for k in a.shape[1:])] = a
    return out


@pytest.mark.parametrize("B,nch,W", [(1, 11, 70), (1, 16, 172), (1, 3, 96), (3, 5, 80), (2, 2, 66), (1, 1, 80)])
def test_lft_tree_scan_against_sequential_application(hip, B, nch, W):
    """The exclusive scan over dense chunk maps (gf_lft_tree_scan: FP64-MFMA GEMM jobs + gf_dense_solve per
    level, identity maps padding to a power of two, several problems per launch) gives the start state of
    every chunk that applying the maps one after the other gives (oracle/lft.py, LAPACK solves)."""
    import torch
    from oracle import lft
    lib, p = hip.load(), hip.ptr
    rng = np.random.default_rng(100 * nch + W + B)
    WP = lib.gf_dense_width(W)
    P = 1 << max(0, (nch - 1).bit_length())
    probs = [lft.random_maps(rng, nch, W) for _ in range(B)]
    arrs = []
    for k in range(5):
        a = np.zeros((B * P, WP, WP) if k < 3 else (B * P, WP))
        for b in range(B):
            a[b * P:b * P + nch] = _padded(np.stack([M[k] for M in probs[b]]), WP)
            if k == 0:
                a[b * P + nch:(b + 1) * P] = np.eye(WP)         # identity maps pad the scan
        arrs.append(torch.as_tensor(a).cuda())
    Xs = torch.full((B * P, WP, WP), np.nan, dtype=torch.float64, device="cuda")
    Ys = torch.full((B * P, WP), np.nan, dtype=torch.float64, device="cuda")
    nw = int(lib.gf_lft_tree_work(B, P, WP))
    assert nw >= 0
    work = torch.empty((max(nw, 1),), dtype=torch.float64, device="cuda")
    rc = lib.gf_lft_tree_scan(B, P, WP, *[p(a) for a in arrs], p(Xs), p(Ys), p(work), None)
    hip.check(rc, "gf_lft_tree_scan")
    torch.cuda.synchronize()
    Xs, Ys = Xs.cpu().numpy(), Ys.cpu().numpy()
    assert np.all(np.isfinite(Xs)) and np.all(np.isfinite(Ys))
    for b in range(B):
        for c, (X, Y) in enumerate(lft.start_states(probs[b])):
            got_X, got_Y = Xs[b * P + c], Ys[b * P + c]
            assert np.abs(got_X[:W, :W] - X).max() <= 1e-9 * max(np.abs(X).max(), 1.0), (b, c)
            assert np.abs(got_Y[:W] - Y).max() <= 1e-9 * max(np.abs(Y).max(), 1.0), (b, c)
            assert np.array_equal(got_X, got_X.T)                 # exactly symmetric by construction
            assert not got_X[W:].any() and not got_X[:, W:].any() and not got_Y[W:].any()


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 0, 64, 64, 64), (0, 0, 172, 172, 172), (0, 1, 80, 96, 50),
                                          (1, 0, 100, 1, 172), (1, 1, 33, 70, 17), (0, 0, 176, 7, 176),
                                          (1, 0, 300, 300, 1000)])
def test_bgemm(hip, ta, tb, M, N, K):
    """gf_bgemm: C = D + op(A) op(B), batched with strides, against numpy."""
    import torch
    lib, p = hip.load(), hip.ptr
    rng = np.random.default_rng(M + 7 * N + 13 * K + ta + 2 * tb)
    batch = 3
    A = rng.normal(size=(batch,) + ((K, M) if ta else (M, K)))
    Bm = rng.normal(size=(batch,) + ((N, K) if tb else (K, N)))
    D = rng.normal(size=(batch, M, N))
    ref = (A.transpose(0, 2, 1) if ta else A) @ (Bm.transpose(0, 2, 1) if tb else Bm)
    Ad, Bd, Dd = (torch.as_tensor(x).cuda() for x in (A, Bm, D))
    for with_d in (False, True):
        C = torch.full((batch, M, N + 3), np.nan, dtype=torch.float64, device="cuda")      # ldc > N
        rc = lib.gf_bgemm(batch, ta, tb, M, N, K, p(Ad), A.shape[2], A.shape[1] * A.shape[2],
                          p(Bd), Bm.shape[2], Bm.shape[1] * Bm.shape[2],
                          p(Dd) if with_d else None, N, M * N, p(C), N + 3, M * (N + 3), None)
        hip.check(rc, "gf_bgemm")
        got = C.cpu().numpy()
        want = ref + (D if with_d else 0.0)
        assert np.abs(got[:, :, :N] - want).max() <= 1e-12 * K * max(1.0, np.abs(want).max())
        assert np.all(np.isnan(got[:, :, N:]))                      # nothing written beyond N columns
    assert lib.gf_bgemm(0, 0, 0, 4, 4, 4, p(Ad), 4, 0, p(Bd), 4, 0, None, 0, 0, p(C), 4, 0, None) != 0


@pytest.mark.parametrize("B,J,N,L", [(1, 40, 700, 256), (3, 33, 1000, 192), (1, 86, 900, 320)])
def test_wide_gram(hip, B, J, N, L):
    """gf_wide_gram: G_c = sum h h^T / d, m_c = sum h z / d over the rows of every chunk (ragged last chunk,
    non-positive pivots skipped), written into the dense map slots [B][P]."""
    import torch
    lib, p = hip.load(), hip.ptr
    rng = np.random.default_rng(J + N)
    W = 2 * J
    WP, CP = lib.gf_dense_width(W), lib.gf_fused_row_stride(0, J)
    nch = -(-N // L)
    P = 1 << max(0, (nch - 1).bit_length())
    h = np.zeros((B * N + 2, CP))
    h[:B * N, :W] = rng.normal(size=(B * N, W))
    d = rng.uniform(0.5, 2.0, size=B * N + 2)
    d[rng.integers(0, B * N, 5)] = -1.0                             # failed pivots contribute nothing
    z = rng.normal(size=B * N + 2)
    hd, dd, zd = (torch.as_tensor(x).cuda() for x in (h, d, z))
    G = torch.full((B * P, WP, WP), np.nan, dtype=torch.float64, device="cuda")
    m = torch.full((B * P, WP), np.nan, dtype=torch.float64, device="cuda")
    first, count = (1, nch - 2) if nch > 3 else (0, nch)           # a sub-range of the chunks, as the combine asks
    rc = lib.gf_wide_gram(B, N, L, nch, first, count, P, J, p(hd), p(dd), p(zd), p(G), p(m), None)
    hip.check(rc, "gf_wide_gram")
    G, m = G.cpu().numpy(), m.cpu().numpy()
    for b in range(B):
        for c in range(nch):
            if not first <= c < first + count:
                assert np.all(np.isnan(G[b * P + c])) and np.all(np.isnan(m[b * P + c]))   # untouched
                continue
            r0, r1 = b * N + c * L, b * N + min(N, (c + 1) * L)
            s = np.where(d[r0:r1] > 0, 1.0 / d[r0:r1], 0.0)
            Gr = (h[r0:r1, :W] * s[:, None]).T @ h[r0:r1, :W]
            mr = (h[r0:r1, :W] * s[:, None]).T @ z[r0:r1]
            g = G[b * P + c]
            assert np.abs(g[:W, :W] - Gr).max() <= 1e-12 * np.abs(Gr).max() * L
            assert np.abs(m[b * P + c, :W] - mr).max() <= 1e-12 * np.abs(mr).max() * L
            assert np.array_equal(g, g.T) and not g[W:].any() and not m[b * P + c, W:].any()


@pytest.mark.parametrize("n", [5, 64, 80, 172, 176])
def test_dense_solve_logdet(hip, n):
    """gf_dense_solve_logdet: log det A where det A > 0 (pivots' signs x parity of the row permutation), NaN
    where det A <= 0 or A is singular; the solution itself as gf_dense_solve."""
    import torch
    lib, p = hip.load(), hip.ptr
    rng = np.random.default_rng(n)
    batch, nrhs = 12, 16
    A = np.eye(n) + 0.4 * rng.normal(size=(batch, n, n)) / np.sqrt(n)
    A[1] = np.eye(n)[rng.permutation(n)] + 0.05 * rng.normal(size=(n, n))     # row exchanges: parity matters
    A[2, :, 0] *= -1.0                                                          # flips the sign of det
    A[3] = 0.0                                                                  # singular
    A[4, :, 3] = np.nan
    B = rng.normal(size=(batch, n, nrhs))
    Ad, Bd = torch.as_tensor(A).cuda(), torch.as_tensor(B).cuda()
    ld = torch.full((batch,), 123.0, dtype=torch.float64, device="cuda")
    hip.check(lib.gf_dense_solve_logdet(batch, n, nrhs, p(Ad), p(Bd), p(ld), None), "gf_dense_solve_logdet")
    got, X = ld.cpu().numpy(), Bd.cpu().numpy()
    for b in range(batch):
        if b in (3, 4):
            assert np.isnan(got[b]), b
            continue
        sign, ref = np.linalg.slogdet(A[b])
        if sign > 0:
            assert abs(got[b] - ref) <= 1e-10 * max(1.0, abs(ref)), (b, got[b], ref)
        else:
            assert np.isnan(got[b]), (b, sign, got[b])
        assert np.abs(X[b] - np.linalg.solve(A[b], B[b])).max() <= 1e-9 * np.abs(X[b]).max()
