// gadfly_dense.hip -- the DENSE part of the time-parallel factorisation of wide kernels (64 <= W <= 176)
//
// A chunk of the time axis maps its start state (X, Y) to its end state by a linear-fractional
// transformation M = (Phi, G, Xbar, Ybar, m):
//     X+ = Xbar + Phi K Phi^T,  K = (I - X G)^-1 X ;   Y+ = Ybar + Phi (I - X G)^-1 (Y - X m)
// and two maps compose into one (M2 after M1, D = (I - X1 G2)^-1):
//     Phi = Phi2 D Phi1 ;  X = X2 + Phi2 D X1 Phi2^T ;  G = G1 + Phi1^T G2 D Phi1
//     v = D (Y1 - X1 m2) ;  Y = Y2 + Phi2 v ;  m = m1 + Phi1^T (m2 - G2 v)
// so the true start states of all chunks come from an exclusive scan over the chunk maps.  For wide
// kernels the maps are W x W matrices and every step is GEMM-shaped -- the one place of the path where
// the FP64 matrix pipe pays.  This file holds that scan as hand-written kernels:
//   k_jobs        a launch = up to 8 "jobs" over a batch of map pairs: 64 x 64 output tiles of batched
//                 GEMMs on v_mfma_f64_16x16x4 (operands staged through LDS, K-steps of 16, register
//                 double buffering; epilogues I - AB, D + AB, and D + AB computed on the upper
//                 triangle and mirrored -- exactly symmetric results at half the tiles), the scan's
//                 mat-vecs, and strided copies
//   k_wide_gram   G_c = sum_n h_n h_n^T / d_n, m_c = sum_n h_n z_n / d_n of a chunk: one workgroup per
//                 chunk, the upper-triangular 16 x 16 tiles of G and the strips of m dealt evenly over its
//                 4 or 8 waves, the rows h_n read ONCE through LDS
//   k_lft_pack / k_lft_unpack   sweep-state layout ([column][row], k_factorw / k_phiw) <-> dense maps
// The one solve per tree level is gf_dense_solve (k_dense_solve, gadfly_hip.hip).
//
// Reference being replaced: celerite2's sequential `factor` reached from gadfly/gp.py:202 (`compute`)
// with the reference's default kernel (gadfly/core.py:430-461, 86 terms, W = 172).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/gadfly_hip.h"
#include "gf_internal.h"
#include "gf_wave.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
#define GF_MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// v_mfma_f64_16x16x4 operand layout (tools/microbench/mfma_f64_layout.hip): lane (i, k) = (lane & 15,
// lane >> 4) supplies A[i][k] and B[k][i]; accumulator register r of that lane is D[k + 4 r][i].

constexpr int GT_LD = 80;               // LDS row stride of a 16 x 64 operand tile: rows k, k + 1 land 32
                                        // banks apart, so one ds_read_b64 of 32 lanes is conflict-free
constexpr int GT_TILE = 16 * GT_LD;     // doubles per operand tile

__device__ __forceinline__ double wave_sum_x(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// One K-step (16 deep) of a 64-wide operand strip into registers, 4 doubles per thread (256 threads).
//   KCONTIG: element (x, k) at p[x * ld + k]  (thread: x = tid / 4, four consecutive k)
//   else   : element (x, k) at p[k * ld + x]  (thread: k = tid / 16, four consecutive x)
// Out-of-range elements are zero.
template <bool KCONTIG>
__device__ __forceinline__ void gt_fetch(double (&v)[4], const double *__restrict__ p, const int ld,
                                         const int x0, const int xlim, const int kb, const int K,
                                         const int tid) {
    const int x = KCONTIG ? x0 + (tid >> 2) : x0 + 4 * (tid & 15);
    const int k = KCONTIG ? kb + 4 * (tid & 3) : kb + (tid >> 4);
    const double *q = p + (size_t)(KCONTIG ? x : k) * ld + (KCONTIG ? k : x);
    const bool all = KCONTIG ? (x < xlim && k + 3 < K) : (k < K && x + 3 < xlim);
    if (all && ((reinterpret_cast<uintptr_t>(q) & 15) == 0)) {
        const double2 a = reinterpret_cast<const double2 *>(q)[0], b = reinterpret_cast<const double2 *>(q)[1];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = KCONTIG ? (x < xlim && k + e < K) : (k < K && x + e < xlim);
            v[e] = ok ? q[e] : 0.0;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void gt_stash(double *S, const double (&v)[4], const int tid) {
    if (KCONTIG) {
        const int x = tid >> 2, k = 4 * (tid & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(k + e) * GT_LD + x] = v[e];
    } else {
        const int k = tid >> 4, x = 4 * (tid & 15);
        reinterpret_cast<double2 *>(S + k * GT_LD + x)[0] = double2{v[0], v[1]};
        reinterpret_cast<double2 *>(S + k * GT_LD + x)[1] = double2{v[2], v[3]};
    }
}

// C tile (rows i0 .. i0 + 63, columns j0 .. j0 + 63) of op(A) (M x K) op(B) (K x N), row-major operands:
// op(A)(i, k) = TA ? A[k lda + i] : A[i lda + k];  op(B)(k, j) = TB ? B[j ldb + k] : B[k ldb + j].
// 256 threads: wave w owns rows i0 + 16 w .. + 15 as four 16 x 16 accumulator tiles.  epi(row, col, value)
// is called for every in-range element.
template <bool TA, bool TB, class Epi>
__device__ __forceinline__ void gemm_tile(const double *__restrict__ A, const int lda,
                                          const double *__restrict__ B, const int ldb, const int M,
                                          const int N, const int K, const int i0, const int j0,
                                          double *lds, Epi epi) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    double *As = lds, *Bs = lds + 2 * GT_TILE;
    const int left = (N - j0 + 15) >> 4;
    const int nct = left < 4 ? left : 4;            // live 16-column tiles
    const bool rows_live = i0 + 16 * w < M;
    d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    double va[4], vb[4];
    gt_fetch<!TA>(va, A, lda, i0, M, 0, K, tid);
    gt_fetch<TB>(vb, B, ldb, j0, N, 0, K, tid);
    gt_stash<!TA>(As, va, tid);
    gt_stash<TB>(Bs, vb, tid);
    __syncthreads();
    int cur = 0;
    for (int kb = 0; kb < K; kb += 16) {
        const bool more = kb + 16 < K;
        if (more) {                                 // the next step's operands travel during the MFMAs
            gt_fetch<!TA>(va, A, lda, i0, M, kb + 16, K, tid);
            gt_fetch<TB>(vb, B, ldb, j0, N, kb + 16, K, tid);
        }
        if (rows_live) {
            const double *a = As + cur * GT_TILE + 16 * w + li, *b = Bs + cur * GT_TILE + li;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double av = a[(4 * ks + lk) * GT_LD];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < nct) acc[q] = GF_MFMA64(av, b[(4 * ks + lk) * GT_LD + 16 * q], acc[q]);
            }
        }
        if (more) {
            gt_stash<!TA>(As + (cur ^ 1) * GT_TILE, va, tid);
            gt_stash<TB>(Bs + (cur ^ 1) * GT_TILE, vb, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (rows_live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nct) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 16 * w + lk + 4 * r, col = j0 + 16 * q + li;
                if (row < M && col < N) epi(row, col, acc[q][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// job launches
// ------------------------------------------------------------------------------------------------
enum { JOB_GEMM = 0, JOB_MATVEC = 1, JOB_COPY = 2 };
enum { EPI_PLAIN = 0,       // C = AB
       EPI_IMINUS = 1,      // C = I - AB
       EPI_ADD = 2,         // C = D + AB
       EPI_ADD_SYMU = 3 };  // C = D + AB on the upper triangle, mirrored (tiles with tn >= tm only)

struct Job {
    const double *A, *B, *D;
    double *C;
    long long sA, sB, sC, sD;           // batch strides (elements)
    double s;
    int type, first;                    // `first`: first blockIdx.x of the job
    int M, N, K, lda, ldb, ldc, ldd;
    int ta, tb, epi, tn;                // tn: tiles per dimension (GEMM) / blocks (COPY)
};
constexpr int MAX_JOBS = 8;
struct JobTable {
    int njobs, nblocks;
    Job j[MAX_JOBS];
};

template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {              // DPP move inside each 16-lane row
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over each 16-lane row, every lane of the row holds it (quad swaps, then the two mirrors)
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_row<0xB1>(v);
    v += dpp_row<0x4E>(v);
    v += dpp_row<0x141>(v);
    v += dpp_row<0x140>(v);
    return v;
}

// y[i] = a[i] + s sum_k Mat(i, k) x[k inc_x],  i < n <= 192, k < kn <= 192;
// Mat(i, k) = trans ? Mt[k ld + i] : Mt[i ld + k].  One workgroup; every load is a coalesced row piece and
// independent of the others (the scan's mat-vecs sit on the critical path of every level: a dependent
// chain of loads and wave reductions per row made them the longest job of their launches).
__device__ __forceinline__ void job_matvec(const Job &J, const double *__restrict__ Mt,
                                           const double *__restrict__ x, const double *__restrict__ a,
                                           double *__restrict__ y, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = J.M, kn = J.K, ld = J.lda, incx = J.ldb, incy = J.ldc;
    double *part = lds;                                 // [4][192]
    if (J.ta) {                         // lanes along i, the k range split over the four waves
        const int kq = (kn + 3) >> 2, k0 = w * kq, k1 = (k0 + kq < kn) ? k0 + kq : kn;
        const int i0 = (lane < n) ? lane : n - 1, i1 = (lane + 64 < n) ? lane + 64 : n - 1,
                  i2 = (lane + 128 < n) ? lane + 128 : n - 1;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
        int k = k0;
        for (; k + 1 < k1; k += 2) {
            const double xa = x[(size_t)k * incx], xb = x[(size_t)(k + 1) * incx];
            const double *ra = Mt + (size_t)k * ld, *rb = ra + ld;
            p0 = fma(ra[i0], xa, p0); p1 = fma(ra[i1], xa, p1); p2 = fma(ra[i2], xa, p2);
            q0 = fma(rb[i0], xb, q0); q1 = fma(rb[i1], xb, q1); q2 = fma(rb[i2], xb, q2);
        }
        if (k < k1) {
            const double xa = x[(size_t)k * incx];
            const double *ra = Mt + (size_t)k * ld;
            p0 = fma(ra[i0], xa, p0); p1 = fma(ra[i1], xa, p1); p2 = fma(ra[i2], xa, p2);
        }
        part[w * 192 + lane] = p0 + q0;
        part[w * 192 + lane + 64] = p1 + q1;
        part[w * 192 + lane + 128] = p2 + q2;
    } else {                            // lanes along k; wave w: rows w, w + 4, ..., four at a time
        const int c0 = (lane < kn) ? lane : kn - 1, c1 = (lane + 64 < kn) ? lane + 64 : kn - 1,
                  c2 = (lane + 128 < kn) ? lane + 128 : kn - 1;
        const double x0 = (lane < kn) ? x[(size_t)c0 * incx] : 0.0;
        const double x1 = (lane + 64 < kn) ? x[(size_t)c1 * incx] : 0.0;
        const double x2 = (lane + 128 < kn) ? x[(size_t)c2 * incx] : 0.0;
        for (int i = w; i < n; i += 16) {
            double p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = (i + 4 * u < n) ? i + 4 * u : n - 1;
                const double *row = Mt + (size_t)r * ld;
                p[u] = fma(row[c0], x0, fma(row[c1], x1, row[c2] * x2));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = row16_sum(p[u]);
            if ((lane & 15) == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + 4 * u < n) part[(lane >> 4) * 192 + i + 4 * u] = p[u];
            }
        }
    }
    __syncthreads();
    if (tid < n) {
        const double v = (part[tid] + part[192 + tid]) + (part[384 + tid] + part[576 + tid]);
        y[(size_t)tid * incy] = fma(J.s, v, a ? a[tid] : 0.0);
    }
}

__global__ void __launch_bounds__(256) k_jobs(const JobTable T) {
    __shared__ __attribute__((aligned(16))) double lds[4 * GT_TILE];
    Job J = T.j[0];
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)              // (constant indices: the table stays in the kernarg segment)
        if (q < T.njobs && (int)blockIdx.x >= T.j[q].first) J = T.j[q];
    const int t = (int)blockIdx.x - J.first;
    const long long bi = blockIdx.y;
    const double *A = J.A ? J.A + bi * J.sA : nullptr;
    const double *B = J.B ? J.B + bi * J.sB : nullptr;
    const double *D = J.D ? J.D + bi * J.sD : nullptr;
    double *C = J.C + bi * J.sC;
    if (J.type == JOB_MATVEC) {
        job_matvec(J, A, B, D, C, lds);
        return;
    }
    if (J.type == JOB_COPY) {                       // rows M x columns N; A == nullptr: zeros
        const long long tot = (long long)J.M * J.N;
        for (long long e = (long long)t * 256 + threadIdx.x; e < tot; e += (long long)J.tn * 256) {
            const int r = (int)(e / J.N), c = (int)(e - (long long)r * J.N);
            C[(size_t)r * J.ldc + c] = A ? A[(size_t)r * J.lda + c] : 0.0;
        }
        return;
    }
    int tm, tn;
    if (J.epi == EPI_ADD_SYMU) {                    // upper-triangular tiles, row by row
        tm = 0;
        int rem = t;
        while (rem >= J.tn - tm) { rem -= J.tn - tm; ++tm; }
        tn = tm + rem;
    } else {
        tm = t / J.tn;
        tn = t - tm * J.tn;
    }
    const int epi = J.epi, ldc = J.ldc, ldd = J.ldd;
    const bool diag_tile = tm == tn;
    auto store = [&](const int row, const int col, const double v) {
        if (epi == EPI_PLAIN) C[(size_t)row * ldc + col] = v;
        else if (epi == EPI_IMINUS) C[(size_t)row * ldc + col] = ((row == col) ? 1.0 : 0.0) - v;
        else if (epi == EPI_ADD) C[(size_t)row * ldc + col] = D[(size_t)row * ldd + col] + v;
        else if (!diag_tile || col >= row) {
            const double o = D[(size_t)row * ldd + col] + v;
            C[(size_t)row * ldc + col] = o;
            C[(size_t)col * ldc + row] = o;
        }
    };
    if (!J.ta && !J.tb) gemm_tile<false, false>(A, J.lda, B, J.ldb, J.M, J.N, J.K, 64 * tm, 64 * tn, lds, store);
    else if (!J.ta && J.tb) gemm_tile<false, true>(A, J.lda, B, J.ldb, J.M, J.N, J.K, 64 * tm, 64 * tn, lds, store);
    else if (J.ta && !J.tb) gemm_tile<true, false>(A, J.lda, B, J.ldb, J.M, J.N, J.K, 64 * tm, 64 * tn, lds, store);
    else gemm_tile<true, true>(A, J.lda, B, J.ldb, J.M, J.N, J.K, 64 * tm, 64 * tn, lds, store);
}

struct JobBuilder {
    JobTable T;
    JobBuilder() { T.njobs = 0; T.nblocks = 0; }
    Job &add(int type, int nblocks) {
        Job &j = T.j[T.njobs++];
        memset(&j, 0, sizeof(j));
        j.type = type;
        j.first = T.nblocks;
        T.nblocks += nblocks;
        return j;
    }
    // C = epi(D, op(A) op(B)),  C: M x N, inner dimension K
    void gemm(int ta, int tb, int epi, int M, int N, int K, const double *A, int lda, long long sA,
              const double *B, int ldb, long long sB, const double *D, int ldd, long long sD, double *C,
              int ldc, long long sC) {
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        Job &j = add(JOB_GEMM, epi == EPI_ADD_SYMU ? tm * (tm + 1) / 2 : tm * tn);
        j.ta = ta; j.tb = tb; j.epi = epi; j.M = M; j.N = N; j.K = K; j.tn = tn;
        j.A = A; j.lda = lda; j.sA = sA; j.B = B; j.ldb = ldb; j.sB = sB;
        j.D = D; j.ldd = ldd; j.sD = sD; j.C = C; j.ldc = ldc; j.sC = sC;
    }
    // y = a + s op(Mt) x   (n outputs, kn terms; a may be null)
    void matvec(int trans, int n, int kn, double s, const double *Mt, int ld, long long sM, const double *x,
                int incx, long long sx, const double *a, long long sa, double *y, int incy, long long sy) {
        Job &j = add(JOB_MATVEC, 1);
        j.ta = trans; j.M = n; j.K = kn; j.s = s;
        j.A = Mt; j.lda = ld; j.sA = sM; j.B = x; j.ldb = incx; j.sB = sx;
        j.D = a; j.sD = sa; j.C = y; j.ldc = incy; j.sC = sy;
    }
    // dst (rows x cols) = src, or zeros when src is null
    void copy(int rows, int cols, const double *src, int lds_, long long ssrc, double *dst, int ldd_,
              long long sdst) {
        long long nb = ((long long)rows * cols + 2047) / 2048;
        if (nb < 1) nb = 1;
        if (nb > 16) nb = 16;
        Job &j = add(JOB_COPY, (int)nb);
        j.M = rows; j.N = cols; j.tn = (int)nb;
        j.A = src; j.lda = lds_; j.sA = ssrc; j.C = dst; j.ldc = ldd_; j.sC = sdst;
    }
    void launch(long long batch, hipStream_t st) {
        if (T.njobs == 0 || batch < 1) return;
        hipLaunchKernelGGL(k_jobs, dim3((unsigned)T.nblocks, (unsigned)batch), dim3(256), 0, st, T);
    }
};

// ------------------------------------------------------------------------------------------------
// Gram sums of a chunk: G = sum_n h_n h_n^T / d_n (upper triangle, mirrored), m = sum_n h_n z_n / d_n
// ------------------------------------------------------------------------------------------------
// The work of a chunk is NT (NT + 1) / 2 upper-triangular 16 x 16 tiles of G plus one 16 x 1 strip of m per row
// strip: NT (NT + 3) / 2 units of one MFMA per 4 rows each.  They are dealt out EVENLY, in row-strip order, over
// the waves of the workgroup(s) of a chunk (NW waves per workgroup, gridDim.y workgroups): with one wave per row
// strip the first wave carried NT + 1 units and the last 2, and the step time was the first wave's (cfg4's
// shard: 24 of a workgroup's 80 MFMAs per K-step on one SIMD, 1.8 TB/s).
template <int NT>
struct GramPlan {
    static constexpr int WP = 16 * NT, UNITS = NT * (NT + 3) / 2, NW = (NT <= 5) ? 4 : 8;
};

template <int NT, int SPLIT>            // WP = 16 NT; SPLIT workgroups share a chunk (few chunks on 256 CUs)
__global__ void __launch_bounds__(64 * GramPlan<NT>::NW)
k_wide_gram(const int64_t N, const int64_t L, const int nch, const int ch0, const int nsel, const int P,
            const int CP, const double *__restrict__ h_, const double *__restrict__ dbar_, const double *__restrict__ zbar_,
            double *__restrict__ G_out, double *__restrict__ m_out) {
    using Plan = GramPlan<NT>;
    constexpr int WP = Plan::WP, NW = Plan::NW, NTH = 64 * NW;
    constexpr int UPW = (Plan::UNITS + NW * SPLIT - 1) / (NW * SPLIT);     // units per wave (at most)
    constexpr int HLD = (WP + 31) / 32 * 32 + 16;   // rows k, k + 1 land 32 banks apart
    constexpr int NV = (8 * WP + NTH - 1) / NTH;    // double2 loads per thread and K-step
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)(tid >> 6) + NW * (int)blockIdx.y);   // wave among NW SPLIT
    const int li = lane & 15, lk = lane >> 4;
    const int sel = blockIdx.x, pr = sel / nsel, ch = ch0 + (sel - pr * nsel);      // chunks ch0 .. ch0 + nsel - 1
    const int64_t c0 = (int64_t)ch * L;
    const int rows = (int)((N - c0 < L) ? (N - c0) : L);
    const size_t pb = (size_t)pr * N + c0;
    const double *__restrict__ H = h_ + pb * CP;
    const double *__restrict__ dg = dbar_ + pb;
    const double *__restrict__ zg = zbar_ + pb;
    __shared__ __attribute__((aligned(16))) double Hs[2][16 * HLD];
    __shared__ double Ss[2][16], Zs[2][16];
    // this wave's units: u = gw UPW + s in row-strip order (strip r: its tiles q = r .. NT - 1, then q = NT: m)
    int ur[UPW], uq[UPW];
    {
        int u0 = gw * UPW, r = 0, base = 0;         // base = first unit of strip r
        while (r < NT && u0 >= base + (NT - r + 1)) { base += NT - r + 1; ++r; }
#pragma unroll
        for (int s = 0; s < UPW; ++s) {
            const int u = u0 + s;
            while (r < NT && u >= base + (NT - r + 1)) { base += NT - r + 1; ++r; }
            ur[s] = (r < NT) ? r : -1;              // (-1: no such unit)
            uq[s] = (r < NT) ? r + (u - base) : 0;
        }
    }
    d4 acc[UPW];
#pragma unroll
    for (int s = 0; s < UPW; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    double2 v[NV];
    double sv = 0.0, zv = 0.0;
    auto fetch = [&](const int kb) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = tid + q * NTH, row = e / (WP / 2), c2 = e - row * (WP / 2);
            const bool ok = e < 8 * WP && kb + row < rows;
            v[q] = ok ? *reinterpret_cast<const double2 *>(H + (size_t)(kb + row) * CP + 2 * c2) : double2{0.0, 0.0};
        }
        if (tid < 16) {
            const bool ok = kb + tid < rows;
            const double d = ok ? dg[kb + tid] : 0.0;
            sv = (d > 0.0) ? 1.0 / d : 0.0;
            zv = ok ? zg[kb + tid] : 0.0;
        }
    };
    auto stash = [&](const int buf) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = tid + q * NTH, row = e / (WP / 2), c2 = e - row * (WP / 2);
            if (e < 8 * WP) *reinterpret_cast<double2 *>(&Hs[buf][row * HLD + 2 * c2]) = v[q];
        }
        if (tid < 16) { Ss[buf][tid] = sv; Zs[buf][tid] = zv; }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int kb = 0; kb < rows; kb += 16) {
        const bool more = kb + 16 < rows;
        if (more) fetch(kb + 16);
        const double *hrow = &Hs[cur][0];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + lk;
            const double sc = Ss[cur][k];
#pragma unroll
            for (int s = 0; s < UPW; ++s) {
                if (ur[s] < 0) continue;            // (wave-uniform)
                const double av = hrow[k * HLD + 16 * ur[s] + li] * sc;
                const double bv = (uq[s] < NT) ? hrow[k * HLD + 16 * uq[s] + li] : ((li == 0) ? Zs[cur][k] : 0.0);
                acc[s] = GF_MFMA64(av, bv, acc[s]);
            }
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const size_t mp = (size_t)pr * P + ch;          // dense map index (the scan pads nch to P)
    double *__restrict__ Gd = G_out + mp * ((size_t)WP * WP);
#pragma unroll
    for (int s = 0; s < UPW; ++s) {
        const int r = ur[s], q = uq[s];
        if (r < 0) continue;
        if (q == NT) {                              // the strip of m
            if (li == 0) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) m_out[mp * WP + 16 * r + lk + 4 * rr] = acc[s][rr];
            }
            continue;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 16 * r + lk + 4 * rr, col = 16 * q + li;
            if (q > r || col >= row) {
                Gd[(size_t)row * WP + col] = acc[s][rr];
                Gd[(size_t)col * WP + row] = acc[s][rr];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sweep-state layout <-> dense maps
//   state slot (k_factorw / k_phiw): [column j][row i] with RP rows per column, CP columns; the forward
//   solve (Y) rides in column CP - 1.  dense: row-major WP x WP, zero pads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lft_pack(const int nch, const int P, const int W, const int WP, const int CP, const int RP,
           const double *__restrict__ S_state, const double *__restrict__ Phi_state,
           double *__restrict__ Ph, double *__restrict__ X, double *__restrict__ Y,
           double *__restrict__ G, double *__restrict__ m) {
    const int mp = blockIdx.x, pr = mp / P, ch = mp - pr * P;
    const int band = blockIdx.y;                    // dense rows 32 band .. 32 band + 31
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const size_t msz = (size_t)WP * WP;
    double *Pd = Ph + mp * msz, *Xd = X + mp * msz, *Yd = Y + (size_t)mp * WP;
    const int i0 = 32 * band;
    // The start states never depend on the LAST chunk's map, and on the FIRST chunk's only through its end
    // state from a zero start (Xbar, Ybar): the last chunk becomes one more identity map, the first keeps
    // Xbar, Ybar with Phi = G = m = 0 -- so callers need not sweep transitions or Gram sums for either.
    if (ch >= nch - 1) {                            // identity map pads the scan
        double *Gd = G + mp * msz, *md = m + (size_t)mp * WP;
        for (int e = tid; e < 32 * WP; e += 256) {
            const int i = i0 + e / WP, j = e % WP;
            if (i < WP) {
                Pd[(size_t)i * WP + j] = (i == j) ? 1.0 : 0.0;
                Xd[(size_t)i * WP + j] = 0.0;
                Gd[(size_t)i * WP + j] = 0.0;
            }
        }
        if (band == 0)
            for (int e = tid; e < WP; e += 256) { Yd[e] = 0.0; md[e] = 0.0; }
        return;
    }
    const size_t slot = (size_t)pr * nch + ch;
    const double *__restrict__ Ss = S_state + slot * ((size_t)CP * RP);
    const double *__restrict__ Ps = Phi_state + slot * ((size_t)CP * RP);
    __shared__ double tP[32][33], tS[32][33];
    const int nt = (WP + 31) / 32;
    for (int t = 0; t < nt; ++t) {
        const int j0 = 32 * t;
        // column j0 + a, rows i0 + tx (coalesced along the rows of a state column)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = ty + 8 * q, j = j0 + a, i = i0 + tx;
            const bool ok = i < W && j < W;
            tP[a][tx] = ok ? Ps[(size_t)j * RP + i] : 0.0;          // Phi(i, j)
            tS[a][tx] = ok ? Ss[(size_t)j * RP + i] : 0.0;          // S(i, j)
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = ty + 8 * q, i = i0 + a, j = j0 + tx;      // dense row i, column j
            if (i < WP && j < WP) {
                Pd[(size_t)i * WP + j] = (ch == 0) ? 0.0 : tP[tx][a];
                if (ch == 0) G[mp * msz + (size_t)i * WP + j] = 0.0;
                // X symmetrised: (S(i, j) + S(j, i)) / 2, S(j, i) read straight (row i of the dense
                // matrix = column i of the state: coalesced along j)
                const double sji = (i < W && j < W) ? Ss[(size_t)i * RP + j] : 0.0;
                Xd[(size_t)i * WP + j] = 0.5 * (tS[tx][a] + sji);
            }
        }
        __syncthreads();
    }
    if (band == 0)
        for (int e = tid; e < WP; e += 256) {
            Yd[e] = (e < W) ? Ss[(size_t)(CP - 1) * RP + e] : 0.0;
            if (ch == 0) m[(size_t)mp * WP + e] = 0.0;
        }
}

__global__ void __launch_bounds__(256)
k_lft_unpack(const int nch, const int P, const int W, const int WP, const int CP, const int RP,
             const double *__restrict__ Xs, const double *__restrict__ Ys, double *__restrict__ S_state) {
    const int slot = blockIdx.x, pr = slot / nch, ch = slot - pr * nch;
    const size_t mp = (size_t)pr * P + ch;
    const double *__restrict__ Xd = Xs + mp * ((size_t)WP * WP);
    const double *__restrict__ Yd = Ys + mp * WP;
    double *__restrict__ Sd = S_state + (size_t)slot * ((size_t)CP * RP);
    for (int e = blockIdx.y * 256 + threadIdx.x; e < CP * RP; e += 256 * gridDim.y) {
        const int col = e / RP, row = e - col * RP;
        double v = 0.0;
        if (row < W) {
            if (col < W) v = Xd[(size_t)col * WP + row];            // X symmetric: S(row, col) = X[col][row]
            else if (col == CP - 1) v = Yd[row];
        }
        Sd[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// the scan
// ------------------------------------------------------------------------------------------------
struct Maps {                           // one level: n maps, dense
    double *Ph, *G, *X, *Y, *m;
};

inline int ilog2(int P) { int l = 0; while ((1 << l) < P) ++l; return l; }

struct TreePlan {
    int B, P, WP, nlev, NR;
    size_t msz, total;
    // offsets into the workspace (doubles)
    size_t lev_maps[32], lev_state[32], scrA, scrR, scrT1, scrT2, scrU;
    TreePlan(int B_, int P_, int WP_) : B(B_), P(P_), WP(WP_) {
        nlev = ilog2(P);
        NR = 2 * WP + 16;
        msz = (size_t)WP * WP;
        size_t off = 0;
        for (int l = 1; l < nlev; ++l) {            // level 0 = the caller's arrays
            const size_t n = (size_t)B * (P >> l);
            lev_maps[l] = off;  off += n * (3 * msz + 2 * WP);
            lev_state[l] = off; off += n * (msz + WP);
        }
        const size_t pairs = (size_t)B * (P > 1 ? P / 2 : 1);
        scrA = off;  off += pairs * msz;
        scrR = off;  off += pairs * (size_t)WP * NR;
        scrT1 = off; off += pairs * msz;
        scrT2 = off; off += pairs * msz;
        scrU = off;  off += pairs * WP;
        total = off;
    }
    Maps maps(int l, double *work) const {
        const size_t n = (size_t)B * (P >> l);
        double *p = work + lev_maps[l];
        Maps M;
        M.Ph = p; M.G = p + n * msz; M.X = p + 2 * n * msz; M.Y = p + 3 * n * msz; M.m = M.Y + n * WP;
        return M;
    }
};

int tree_scan(const TreePlan &T, const Maps &M0, double *Xs0, double *Ys0, double *work, hipStream_t st) {
    const int WP = T.WP, NR = T.NR, nlev = T.nlev;
    const long long msz = (long long)T.msz;
    const size_t nmaps0 = (size_t)T.B * T.P;
    if (nlev == 0) {                                // one chunk per problem: the start state is zero
        (void)hipMemsetAsync(Xs0, 0, nmaps0 * T.msz * sizeof(double), st);
        (void)hipMemsetAsync(Ys0, 0, nmaps0 * WP * sizeof(double), st);
        return gf_internal_check_launch("gf_lft_tree_scan");
    }
    double *A = work + T.scrA, *R = work + T.scrR, *T1 = work + T.scrT1, *T2 = work + T.scrT2, *U = work + T.scrU;
    const long long sR = (long long)WP * NR;
    auto level_maps = [&](int l) { return l == 0 ? M0 : T.maps(l, work); };
    auto state_X = [&](int l) { return l == 0 ? Xs0 : work + T.lev_state[l]; };
    auto state_Y = [&](int l) { return l == 0 ? Ys0 : work + T.lev_state[l] + (size_t)T.B * (T.P >> l) * T.msz; };
    // ---- up-sweep: level l pairs (2k, 2k + 1) -> level l + 1 map k (the root is never needed) ----
    for (int l = 0; l + 1 < nlev; ++l) {
        const Maps S = level_maps(l), Dn = T.maps(l + 1, work);
        const long long pairs = (long long)T.B * (T.P >> (l + 1));
        const double *PL = S.Ph, *PR = S.Ph + msz, *GL = S.G, *GR = S.G + msz, *XL = S.X, *XR = S.X + msz;
        const double *YL = S.Y, *YR = S.Y + WP, *mL = S.m, *mR = S.m + WP;
        const long long s2 = 2 * msz, v2 = 2 * WP;
        {   // A = I - X_L G_R ; rhs = [Y_L - X_L m_R, 0 ... | Phi_L | X_L]
            JobBuilder jb;
            jb.gemm(0, 0, EPI_IMINUS, WP, WP, WP, XL, WP, s2, GR, WP, s2, nullptr, 0, 0, A, WP, msz);
            jb.matvec(1, WP, WP, -1.0, XL, WP, s2, mR, 1, v2, YL, v2, R, NR, sR);      // X symmetric: T form
            jb.copy(WP, 15, nullptr, 0, 0, R + 1, NR, sR);
            jb.copy(WP, WP, PL, WP, s2, R + 16, NR, sR);
            jb.copy(WP, WP, XL, WP, s2, R + 16 + WP, NR, sR);
            jb.launch(pairs, st);
        }
        if (gf_dense_solve((int)pairs, WP, NR, A, R, st)) return -1;
        {   // Phi = Phi_R (D Phi_L) ; T1 = Phi_R (D X_L) ; T2 = G_R (D Phi_L) ; Y = Y_R + Phi_R v ; u = m_R - G_R v
            JobBuilder jb;
            jb.gemm(0, 0, EPI_PLAIN, WP, WP, WP, PR, WP, s2, R + 16, NR, sR, nullptr, 0, 0, Dn.Ph, WP, msz);
            jb.gemm(0, 0, EPI_PLAIN, WP, WP, WP, PR, WP, s2, R + 16 + WP, NR, sR, nullptr, 0, 0, T1, WP, msz);
            jb.gemm(0, 0, EPI_PLAIN, WP, WP, WP, GR, WP, s2, R + 16, NR, sR, nullptr, 0, 0, T2, WP, msz);
            jb.matvec(0, WP, WP, 1.0, PR, WP, s2, R, NR, sR, YR, v2, Dn.Y, 1, WP);
            jb.matvec(1, WP, WP, -1.0, GR, WP, s2, R, NR, sR, mR, v2, U, 1, WP);        // G symmetric
            jb.launch(pairs, st);
        }
        {   // X = X_R + T1 Phi_R^T ; G = G_L + Phi_L^T T2 (both symmetric) ; m = m_L + Phi_L^T u
            JobBuilder jb;
            jb.gemm(0, 1, EPI_ADD_SYMU, WP, WP, WP, T1, WP, msz, PR, WP, s2, XR, WP, s2, Dn.X, WP, msz);
            jb.gemm(1, 0, EPI_ADD_SYMU, WP, WP, WP, PL, WP, s2, T2, WP, msz, GL, WP, s2, Dn.G, WP, msz);
            jb.matvec(1, WP, WP, 1.0, PL, WP, s2, U, 1, WP, mL, v2, Dn.m, 1, WP);
            jb.launch(pairs, st);
        }
    }
    // ---- down-sweep: parent state (level l + 1, block k) -> children (level l, blocks 2k, 2k + 1) ----
    {   // top: the parent state is zero, so child 0 starts from zero and child 1 from (Xbar, Ybar) of map 0
        const int l = nlev - 1;
        const Maps S = level_maps(l);
        double *Xc = state_X(l), *Yc = state_Y(l);
        JobBuilder jb;
        jb.copy(WP, WP, nullptr, 0, 0, Xc, WP, 2 * msz);
        jb.copy(1, WP, nullptr, 0, 0, Yc, WP, 2 * WP);
        jb.copy(WP, WP, S.X, WP, 2 * msz, Xc + msz, WP, 2 * msz);
        jb.copy(1, WP, S.Y, WP, 2 * WP, Yc + WP, WP, 2 * WP);
        jb.launch(T.B, st);
    }
    const int NRa = WP + 16;
    const long long sRa = (long long)WP * NRa;
    for (int l = nlev - 2; l >= 0; --l) {
        const Maps S = level_maps(l);               // the LEFT map of every pair is applied
        const long long pairs = (long long)T.B * (T.P >> (l + 1));
        const double *Xp = state_X(l + 1), *Yp = state_Y(l + 1);
        double *Xc = state_X(l), *Yc = state_Y(l);
        const long long s2 = 2 * msz, v2 = 2 * WP;
        {   // A = I - X G ; rhs = [Y - X m, 0 ... | X] ; left child inherits the parent's state
            JobBuilder jb;
            jb.gemm(0, 0, EPI_IMINUS, WP, WP, WP, Xp, WP, msz, S.G, WP, s2, nullptr, 0, 0, A, WP, msz);
            jb.matvec(1, WP, WP, -1.0, Xp, WP, msz, S.m, 1, v2, Yp, WP, R, NRa, sRa);
            jb.copy(WP, 15, nullptr, 0, 0, R + 1, NRa, sRa);
            jb.copy(WP, WP, Xp, WP, msz, R + 16, NRa, sRa);
            jb.copy(WP, WP, Xp, WP, msz, Xc, WP, s2);
            jb.copy(1, WP, Yp, WP, WP, Yc, WP, v2);
            jb.launch(pairs, st);
        }
        if (gf_dense_solve((int)pairs, WP, NRa, A, R, st)) return -1;
        {   // T1 = Phi K ; Y+ = Ybar + Phi v
            JobBuilder jb;
            jb.gemm(0, 0, EPI_PLAIN, WP, WP, WP, S.Ph, WP, s2, R + 16, NRa, sRa, nullptr, 0, 0, T1, WP, msz);
            jb.matvec(0, WP, WP, 1.0, S.Ph, WP, s2, R, NRa, sRa, S.Y, v2, Yc + WP, 1, v2);
            jb.launch(pairs, st);
        }
        {   // X+ = Xbar + T1 Phi^T (symmetric)
            JobBuilder jb;
            jb.gemm(0, 1, EPI_ADD_SYMU, WP, WP, WP, T1, WP, msz, S.Ph, WP, s2, S.X, WP, s2, Xc + msz, WP, s2);
            jb.launch(pairs, st);
        }
    }
    return gf_internal_check_launch("gf_lft_tree_scan");
}

// ------------------------------------------------------------------------------------------------
// Is every pivot of a chunk positive?  det(I - X G) > 0 only says that the NUMBER of non-positive pivots of the
// chunk is even.  With X (the true start state) and G (the nominal pass' Gram sums) symmetric positive
// semi-definite and X = R R^T, the non-zero eigenvalues of X G are those of the symmetric R^T G R, and the number
// of them >= 1 IS the number of non-positive pivots of the chunk (each row is a rank-one step of G and
// d_n / dbar_n the ratio of successive determinants).  So:  all pivots positive  <=>  M = I - R^T G R positive
// definite, which an unpivoted Cholesky attempt decides (k_spd_check; M's eigenvalues 1 - mu_i lie in (0, 1] for a
// healthy chunk: a well-scaled test, pivots compared with their own diagonal entry).
//   k_pchol      R by Cholesky with diagonal pivoting, stopped where the largest remaining diagonal entry has
//                fallen to PCHOL_TOL of its original value: X is a covariance of state ESTIMATES and numerically
//                rank-deficient whenever fewer combinations of the data are informative than there are states (an
//                un-pivoted attempt on the congruent form (I - X G) X fails on healthy chunks for that reason);
//                the factor is written transposed (Rt[k][i] = R[i][k], zero rows past the rank) for the GEMM tiles
//   two GEMM jobs    T = G Rt^T,  M = I - Rt T
// A failed attempt marks the chunk's log det correction NaN: the caller repeats the evaluation with a final pass,
// which names the failing row the way celerite2 does (/root/reference/gadfly/gp.py:188-192).  A rounding-level
// false alarm costs that repeat, never a wrong value.  One workgroup per map, the lower triangle packed by rows in
// (dynamic) LDS: n (n + 1) / 2 + 2 n doubles.
// ------------------------------------------------------------------------------------------------
constexpr int SPD_THREADS = 512;
constexpr double SPD_TOL = 1e-12;       // ~ 30 n eps at n = 176: pivots this far inside rounding count as zero
constexpr double PCHOL_TOL = 1e-11;     // relative size of a remaining diagonal entry of X that still gets a column: below
                                        // it sits what the combine's rounding left (either sign), not variance

__global__ void __launch_bounds__(SPD_THREADS)
k_pchol(const int P, const int first, const int count, const int n, const double *__restrict__ X_,
        double *__restrict__ Rt_) {
    const int pr = blockIdx.x / count, c = first + (blockIdx.x - pr * count);
    const size_t mp = (size_t)pr * P + c;
    const double *__restrict__ X = X_ + mp * (size_t)n * n;
    double *__restrict__ Rt = Rt_ + mp * (size_t)n * n;
    extern __shared__ __attribute__((aligned(16))) double spd_lds[];
    double *T = spd_lds;                            // T[i (i + 1) / 2 + j], j <= i
    double *d0 = spd_lds + (size_t)n * (n + 1) / 2; // 1 / original diagonal (0: row not in play)
    double *col = d0 + n;                           // the current column of R
    __shared__ double s_val[SPD_THREADS / 64];
    __shared__ int s_idx[SPD_THREADS / 64];
    __shared__ int s_done[192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NWV = blockDim.x >> 6;
    for (int i = wave; i < n; i += NWV)
        for (int j = lane; j <= i; j += 64) T[i * (i + 1) / 2 + j] = X[(size_t)i * n + j];
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const double v = T[i * (i + 1) / 2 + i];
        const bool live = v > 0.0;                  // (zero: padding / a zero state; negative: rounding of a null direction)
        d0[i] = live ? 1.0 / v : 0.0;
        s_done[i] = live ? 0 : 1;
    }
    __syncthreads();
    int k = 0;
    for (; k < n; ++k) {
        // largest remaining diagonal entry relative to its original value
        if (tid < 192) {
            double v = -1.0;
            int ix = tid;
            if (tid < n && !s_done[tid]) v = T[tid * (tid + 1) / 2 + tid] * d0[tid];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double ov = __shfl_xor(v, m);
                const int oi = __shfl_xor(ix, m);
                if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            if (lane == 0) { s_val[wave] = v; s_idx[wave] = ix; }
        }
        __syncthreads();
        double best = s_val[0];
        int p = s_idx[0];
#pragma unroll
        for (int w = 1; w < 3; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < p)) { best = s_val[w]; p = s_idx[w]; }
        if (!(best > PCHOL_TOL)) break;             // (uniform) numerical rank reached
        const double piv = T[p * (p + 1) / 2 + p];
        const double r = 1.0 / sqrt(piv);
        for (int i = tid; i < n; i += blockDim.x) {
            double v = 0.0;
            if (!s_done[i]) v = (i >= p ? T[i * (i + 1) / 2 + p] : T[p * (p + 1) / 2 + i]) * r;
            col[i] = v;
            Rt[(size_t)k * n + i] = v;
        }
        __syncthreads();
        if (tid == 0) s_done[p] = 1;
        // trailing update of the rows still in play (row p itself is finished: its column stays in `col` this step)
        for (int i = wave; i < n; i += NWV) {
            if (s_done[i] || i == p) continue;      // (wave-uniform)
            const double ci = col[i];
            double *row = T + i * (i + 1) / 2;
            for (int j = lane; j <= i; j += 64) row[j] = fma(-ci, col[j], row[j]);
        }
        __syncthreads();
    }
    for (int e = k * n + tid; e < n * n; e += blockDim.x) Rt[e] = 0.0;     // rows past the numerical rank
}

__global__ void __launch_bounds__(SPD_THREADS)
k_spd_check(const int P, const int first, const int count, const int n, const double *__restrict__ M_,
            double *__restrict__ ld) {
    const int pr = blockIdx.x / count, c = first + (blockIdx.x - pr * count);
    const size_t mp = (size_t)pr * P + c;
    const double *__restrict__ S = M_ + mp * (size_t)n * n;
    extern __shared__ __attribute__((aligned(16))) double spd_lds[];
    double *T = spd_lds;                            // T[i (i + 1) / 2 + j], j <= i
    double *d0 = spd_lds + (size_t)n * (n + 1) / 2; // original diagonal
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NWV = blockDim.x >> 6;
    if (tid == 0) s_fail = 0;
    for (int i = wave; i < n; i += NWV)             // rows: coalesced pieces of the lower triangle
        for (int j = lane; j <= i; j += 64) T[i * (i + 1) / 2 + j] = S[(size_t)i * n + j];
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const double v = T[i * (i + 1) / 2 + i];
        d0[i] = v;
        if (!(v > 0.0)) s_fail = 1;                 // non-positive (or NaN) diagonal: not positive definite
    }
    __syncthreads();
    for (int k = 0; k < n && !s_fail; ++k) {
        const double piv = T[k * (k + 1) / 2 + k];
        if (!(piv > SPD_TOL * d0[k])) {             // (uniform: every thread reads the same two values)
            __syncthreads();
            if (tid == 0) s_fail = 1;
            __syncthreads();
            break;
        }
        const double r = 1.0 / sqrt(piv);
        __syncthreads();                            // everyone has read the pivot
        for (int i = k + 1 + tid; i < n; i += blockDim.x) T[i * (i + 1) / 2 + k] *= r;      // column k of L
        __syncthreads();
        // trailing update: rows dealt over the waves, a row's columns over the lanes; the lane's entries of
        // column k are kept in registers (j = k + 1 + lane + 64 q)
        double lk[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = k + 1 + lane + 64 * q;
            lk[q] = (j < n) ? T[j * (j + 1) / 2 + k] : 0.0;
        }
        for (int i = k + 1 + wave; i < n; i += NWV) {
            const double lik = T[i * (i + 1) / 2 + k];
            double *row = T + i * (i + 1) / 2;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int j = k + 1 + lane + 64 * q;
                if (j <= i) row[j] = fma(-lik, lk[q], row[j]);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0 && s_fail) ld[mp] = __longlong_as_double(0x7ff8000000000000LL);
}

// ------------------------------------------------------------------------------------------------
// Log-likelihood of a chunked series WITHOUT a final pass.  With (X, Y) the true start state of a chunk and
// (G, m) its Gram sums from the nominal pass (zero start), the chunk's sums over its rows satisfy
//     sum log d_n     = sum log dbar_n        + log det(I - X G)
//     sum z_n^2 / d_n = sum zbar_n^2 / dbar_n + e^T G v - 2 m^T e - m^T X m,   e = Y - X m,  v = (I - X G)^-1 e
// (row by row: d = dbar det(I - Delta u u^T / dbar); the determinants telescope through the closed-loop
// transitions exactly as the Gram sums do -- verified against the sequential recurrence in 80-bit numpy and
// by tests/test_gpu_configs.py).  det(I - X G) <= 0 means a non-positive pivot somewhere in the chunk: the
// correction is then NaN and the caller repeats the evaluation with a final pass.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_corr_finish(const int P, const int first, const int count, const int WP, const double *__restrict__ Y,
              const double *__restrict__ m, const double *__restrict__ e, const double *__restrict__ u1,
              const double *__restrict__ ld, double *__restrict__ acc) {
    const int pr = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[2][256];
    double sl = 0.0, sq = 0.0;
    for (int c = first; c < first + count; ++c) {
        const size_t mp = (size_t)pr * P + c;
        for (int i = tid; i < WP; i += 256) {
            const size_t k = mp * WP + i;
            sq += e[k] * u1[k] - 2.0 * m[k] * e[k] - m[k] * (Y[k] - e[k]);
        }
        if (tid == 0) sl += ld[mp];
    }
    red[0][tid] = sl;
    red[1][tid] = sq;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {         // fixed-shape tree: deterministic
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        acc[3 * pr + 0] += red[0][0];
        acc[3 * pr + 1] += red[1][0];
    }
}

// ------------------------------------------------------------------------------------------------
// The corrections of a chunk whose maps live in 64 x 64 slots (celerite width W <= 63), in ONE kernel:
//   the sign of every pivot, through the symmetric form:  X = R R^T  (diagonal pivoting, numerical rank r),
//       M = I_r - R^T G R  eliminated without pivoting (fails <=> some pivot of the chunk is not positive -> NaN);
//   the values, through A = I - X G, Gauss-Jordan with partial pivoting:  log det(I - X G),  v = A^-1 e
//       (e = Y - X m),  e^T G v - 2 m^T e - m^T X m = (G e)^T v - 2 m^T e - m^T (Y - e).
// One workgroup of 256 threads per map.  The three eliminations run on REGISTERS (the layout of the tree combine's
// Gauss-Jordan: lanes = rows, wave w owns the columns c = w mod 4, 16 registers per lane + the right-hand side):
// per pivot ONE barrier -- the wave that owns the pivot column publishes the multipliers, every wave updates its
// columns with the pivot row's entries fetched by readlane.  Kept in LDS, with rows of n + 1 doubles (n = W rounded
// up to 4), the steps took 5500-6700 cycles per pivot (three barriers and dependent LDS round trips; 88 % of the
// kernel, which at N = 1e6, W = 60 cost more than the final sweep it replaces); in registers < 1500.  The products
// (T = G R, M, A) stay 4 x 4 register tiles over two n x (n + 1) LDS buffers.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_corr_small(const int P, const int first, const int count, const int n, const double *__restrict__ X_,
             const double *__restrict__ Y_, const double *__restrict__ G_, const double *__restrict__ m_,
             double *__restrict__ ld_out, double *__restrict__ quad_out) {
    const int pr = blockIdx.x / count, c = first + (blockIdx.x - pr * count);
    const size_t mp = (size_t)pr * P + c;
    const double *__restrict__ X = X_ + mp * 4096, *__restrict__ G = G_ + mp * 4096;
    const double *__restrict__ Y = Y_ + mp * 64, *__restrict__ mv = m_ + mp * 64;
    const int LD = n + 1;
    extern __shared__ __attribute__((aligned(16))) double spd_lds[];
    double *A = spd_lds, *Rm = A + n * LD;
    double *ve = Rm + n * LD, *vw1 = ve + 64, *vy = vw1 + 64, *vm = vy + 64, *fbuf = vm + 64 /* [2][64] */,
           *pinvbuf = fbuf + 128 /* [2] */;
    int *pvbuf = reinterpret_cast<int *>(pinvbuf + 2);          // [2]: pivot lane (elimination of A), ok flag (of M)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // ---- X -> A, Y, m; R := 0
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        A[i * LD + j] = X[i * 64 + j];
        Rm[i * LD + j] = 0.0;
    }
    if (tid < 64) { vy[tid] = (tid < n) ? Y[tid] : 0.0; vm[tid] = (tid < n) ? mv[tid] : 0.0; }
    __syncthreads();
    // ---- e = Y - X m (X symmetric, full rows are there)
    if (tid < 64) {
        double acc = vy[tid];
        if (tid < n)
            for (int j = 0; j < n; ++j) acc = fma(-A[tid * LD + j], vm[j], acc);
        ve[tid] = acc;
    }
    // ---- X = R R^T with diagonal pivoting, on registers: Rg[lc] = element (lane, 4 lc + w); every wave follows the
    // diagonal by itself (dg: the same numbers in all four, so all four choose the same pivot without talking)
    double Rg[17];
#pragma unroll
    for (int lc = 0; lc < 16; ++lc) {
        const int cc = 4 * lc + w;
        Rg[lc] = (lane < n && cc < n) ? A[lane * LD + cc] : 0.0;
    }
    Rg[16] = 0.0;
    double dg = (lane < n) ? A[lane * LD + lane] : 0.0;
    const double d0 = (dg > 0.0) ? 1.0 / dg : 0.0;
    bool used = !(dg > 0.0);
    int rank = 0;
    for (int k = 0; k < n; ++k) {
        const double cand = used ? -1.0 : dg * d0;
        const double vmx = wave_max(cand);
        if (!(vmx > PCHOL_TOL)) break;              // (uniform over the workgroup) numerical rank reached
        const unsigned long long hit = __ballot(cand == vmx);
        const int p = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hit) - 1);
        const int buf = k & 1;
        if (w == (p & 3)) {                         // this wave owns column p
            const int lp = p >> 2;
            double cval = 0.0;
#pragma unroll
            for (int lc = 0; lc < 16; ++lc) cval = (lc == lp) ? Rg[lc] : cval;
            const double r = 1.0 / sqrt(read_lane(cval, p));
            const double cl = used ? 0.0 : cval * r;
            fbuf[buf * 64 + lane] = cl;
            if (lane < n) Rm[lane * LD + k] = cl;
        }
        __syncthreads();
        const double cl = fbuf[buf * 64 + lane];
#pragma unroll
        for (int lc = 0; lc < 16; ++lc) Rg[lc] = fma(-cl, read_lane(cl, 4 * lc + w), Rg[lc]);
        dg = fma(-cl, cl, dg);
        if (lane == p) used = true;
        rank = k + 1;
    }
    __syncthreads();
    // ---- G -> A (whole rows), w1 = G e
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        A[i * LD + j] = G[i * 64 + j];
    }
    __syncthreads();
    if (tid < n) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc = fma(A[tid * LD + j], ve[j], acc);
        vw1[tid] = acc;
    }
    // ---- T = G R as 4 x 4 register tiles (tile (ti, tj): rows 4 ti.., columns 4 tj.. < rank), then T -> A
    const int nt = n >> 2, ti = tid >> 4, tj = tid & 15;
    const bool tile = ti < nt && tj < nt;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    if (tile && 4 * tj < rank) {
        for (int l = 0; l < n; ++l) {
            double ga[4], rb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) ga[a] = A[(4 * ti + a) * LD + l];
#pragma unroll
            for (int b = 0; b < 4; ++b) rb[b] = Rm[l * LD + 4 * tj + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fma(ga[a], rb[b], acc[a][b]);
        }
    }
    __syncthreads();
    if (tile) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) A[(4 * ti + a) * LD + 4 * tj + b] = acc[a][b];
    }
    __syncthreads();
    // ---- M = I - R^T T (rank x rank; identity beyond), then M -> A
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    if (tile && 4 * ti < rank && 4 * tj < rank) {
        for (int l = 0; l < n; ++l) {
            double ra[4], tb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) ra[a] = Rm[l * LD + 4 * ti + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) tb[b] = A[l * LD + 4 * tj + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fma(ra[a], tb[b], acc[a][b]);
        }
    }
    __syncthreads();
    if (tile) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                A[(4 * ti + a) * LD + 4 * tj + b] = ((4 * ti + a == 4 * tj + b) ? 1.0 : 0.0) - acc[a][b];
    }
    __syncthreads();
    // ---- M eliminated without pivoting (the leading `rank` rows / columns): its pivots are those of M = L L^T
    // squared; each against its own diagonal entry.  Rows below the pivot only.
#pragma unroll
    for (int lc = 0; lc < 16; ++lc) {
        const int cc = 4 * lc + w;
        Rg[lc] = (lane < rank && cc < rank) ? A[lane * LD + cc] : 0.0;
    }
    const double dm = (lane < rank) ? A[lane * LD + lane] : 1.0;
    bool fail = false;
    static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value, wo = k & 3, lk = k >> 2, buf = k & 1;
        if (k >= rank || fail) return;              // (uniform over the workgroup)
        if (w == wo) {
            const double cval = Rg[lk];
            const double piv = read_lane(cval, k);
            const bool ok = piv > SPD_TOL * read_lane(dm, k);
            fbuf[buf * 64 + lane] = (ok && lane > k) ? cval * fast_rcp(piv) : 0.0;
            if (lane == 0) pvbuf[buf] = ok ? 1 : 0;
        }
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(pvbuf[buf])) { fail = true; return; }
        const double f = fbuf[buf * 64 + lane];
#pragma unroll
        for (int lc = lk; lc < 16; ++lc) Rg[lc] = fma(-f, read_lane(Rg[lc], k), Rg[lc]);
    }, std::make_integer_sequence<int, 64>{});
    if (fail) {                                     // (uniform) some pivot of the chunk is not positive
        if (tid == 0) { ld_out[mp] = nan; quad_out[mp] = nan; }
        return;
    }
    __syncthreads();
    // ---- the VALUES come from an elimination of A = I - X G with partial pivoting, not from M: the start states
    // reach this kernel through the combine's products, and what rounding leaves in the numerically null directions
    // of X (1e-13 of the diagonal, either sign) meets entries of G that are as large as X's are small there.  In
    // det(I - X G) the signed residue cancels to first order; R keeps its positive part only, and log det M came
    // out 1e-4 off on ill-scaled problems (22 of 800 random seeds beyond the 1e-8 bar, tools/random_sweep.py).
    // X, G again (the buffers were reused), A = I - X G as 4 x 4 tiles; Gauss-Jordan with implicit row pivoting on
    // [A | e]; log |det| from the pivots, its sign from their signs and the parity of the row -> variable map.
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        A[i * LD + j] = X[i * 64 + j];
        Rm[i * LD + j] = G[i * 64 + j];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    if (tile) {
        for (int l = 0; l < n; ++l) {
            double xa[4], gb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xa[a] = A[(4 * ti + a) * LD + l];
#pragma unroll
            for (int b = 0; b < 4; ++b) gb[b] = Rm[l * LD + 4 * tj + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fma(xa[a], gb[b], acc[a][b]);
        }
    }
    __syncthreads();
    if (tile) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                A[(4 * ti + a) * LD + 4 * tj + b] = ((4 * ti + a == 4 * tj + b) ? 1.0 : 0.0) - acc[a][b];
    }
    __syncthreads();
#pragma unroll
    for (int lc = 0; lc < 16; ++lc) {
        const int cc = 4 * lc + w;
        Rg[lc] = (lane < n && cc < n) ? A[lane * LD + cc] : 0.0;
    }
    Rg[16] = (w == 0) ? ve[lane] : 0.0;             // column 64: the right-hand side (ve is zero beyond n)
    bool pivoted = lane >= n;                       // row `lane` has served as a pivot (or is not one of the system's)
    int myk = lane;                                 // ... of which variable
    double mypinv = 1.0;                            // ... with which 1 / pivot
    // (the search one step ahead of the elimination, as in the tree combine's cb_gauss_jordan)
    if (w == 0) gj_search(Rg[0], pivoted, lane, fbuf, pvbuf, pinvbuf);
    static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value, lk = k >> 2, buf = k & 1;
        constexpr int k1 = k + 1, w1 = k1 & 3, lk1 = k1 >> 2, buf1 = k1 & 1;
        if (k >= n) return;
        __syncthreads();
        const double f = fbuf[buf * 64 + lane];
        const int pv = __builtin_amdgcn_readfirstlane(pvbuf[buf]);
        if (lane == pv) { pivoted = true; myk = k; mypinv = pinvbuf[buf]; }
        if (k1 < 64 && k1 < n && w == w1) {                     // (wave-uniform) the next column's owner
            Rg[lk1] = fma(-f, read_lane(Rg[lk1], pv), Rg[lk1]);
            gj_search(Rg[lk1], pivoted, lane, fbuf + buf1 * 64, pvbuf + buf1, pinvbuf + buf1);
#pragma unroll
            for (int lc = lk; lc < 17; ++lc)
                if (lc != lk1) Rg[lc] = fma(-f, read_lane(Rg[lc], pv), Rg[lc]);
        } else {
#pragma unroll
            for (int lc = lk; lc < 17; ++lc) Rg[lc] = fma(-f, read_lane(Rg[lc], pv), Rg[lc]);
        }
    }, std::make_integer_sequence<int, 64>{});
    // ---- row `lane` solved variable myk: v(myk) = (its right-hand side) / pivot; then the quadratic form and the
    // determinant (wave 0 holds column 64)
    if (w == 0) {
        const double v = Rg[16] * mypinv;
        double q = 0.0, lg = 0.0;
        if (lane < n) {
            q = vw1[myk] * v - 2.0 * vm[lane] * ve[lane] - vm[lane] * (vy[lane] - ve[lane]);
            lg = -log(fabs(mypinv));
        }
        q = wave_sum_x(q);
        lg = wave_sum_x(lg);
        // sign: negative pivots, and the inversions of lane -> myk (identity beyond n)
        int inv = 0;
#pragma unroll
        for (int j = 0; j < 63; ++j) {
            const int oj = __builtin_amdgcn_readlane(myk, j);
            inv += (lane > j && oj > myk) ? 1 : 0;
        }
        const int odd = (__popcll(__ballot(mypinv < 0.0)) + __popcll(__ballot(inv & 1))) & 1;
        if (lane == 0) {
            const bool positive = !odd && fabs(lg) < 1.0e300;       // (a column without a pivot: inf / NaN)
            ld_out[mp] = positive ? lg : nan;
            quad_out[mp] = positive ? q : nan;
        }
    }
}

__global__ void __launch_bounds__(256)
k_corr_finish_small(const int P, const int first, const int count, const double *__restrict__ ld,
                    const double *__restrict__ quad, double *__restrict__ acc) {
    const int pr = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[2][256];
    double sl = 0.0, sq = 0.0;
    for (int c = first + tid; c < first + count; c += 256) {
        sl += ld[(size_t)pr * P + c];
        sq += quad[(size_t)pr * P + c];
    }
    red[0][tid] = sl;
    red[1][tid] = sq;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {         // fixed-shape tree: deterministic
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        acc[3 * pr + 0] += red[0][0];
        acc[3 * pr + 1] += red[1][0];
    }
}

// corrections of the chunks first .. first + count - 1 of every problem, added to acc [B][3] (sum log d,
// sum z^2 / d, min d: the accumulators of gf_reduce_tile); maps [B * P][WP x WP] / [B * P][WP]; work:
// n (3 WP^2 + 18 WP + 1) doubles for n = B P
size_t corrections_work(size_t n, int WP) { return n * (3 * (size_t)WP * WP + 18 * (size_t)WP + 1); }

// W <= 63 (64 x 64 slots): everything in k_corr_small
int chunk_corrections_small(int B, int P, int W, int first, int count, const double *X, const double *Y,
                            const double *G, const double *m, double *acc, double *work, hipStream_t st) {
    if (count < 1) return 0;
    const long long n = (long long)B * P;
    double *ld = work, *quad = work + n;
    const int na = (W + 3) / 4 * 4;
    const size_t lds = sizeof(double) * (2 * (size_t)na * (na + 1) + 7 * 64);
    const void *fn[1] = {(const void *)k_corr_small};
    if (lds > 64 * 1024 && !gf_internal_lds_opt_in(3, st, fn, 1, sizeof(double) * (2 * 64 * 65 + 7 * 64)))
        return gf_internal_error(-1, "chunk corrections: cannot opt in to %lld bytes of LDS", (long long)lds);
    hipLaunchKernelGGL(k_corr_small, dim3((unsigned)(B * count)), dim3(256), lds, st, P, first, count, na, X, Y, G, m,
                       ld, quad);
    hipLaunchKernelGGL(k_corr_finish_small, dim3(B), dim3(256), 0, st, P, first, count, ld, quad, acc);
    return 0;
}

int chunk_corrections(int B, int P, int WP, int first, int count, const double *X, const double *Y,
                      const double *G, const double *m, double *acc, double *work, hipStream_t st) {
    if (count < 1) return 0;
    const long long n = (long long)B * P, msz = (long long)WP * WP;
    double *A = work, *R = A + n * msz, *ev = R + n * WP * 16, *u1 = ev + n * WP, *ld = u1 + n * WP, *Rt = ld + n, *Mx = Rt + n * msz;
    const long long sR = (long long)WP * 16;
    {   // A = I - X G ; e = Y - X m (X symmetric: the coalesced mat-vec form), once as the right-hand side
        JobBuilder jb;
        jb.gemm(0, 0, EPI_IMINUS, WP, WP, WP, X, WP, msz, G, WP, msz, nullptr, 0, 0, A, WP, msz);
        jb.matvec(1, WP, WP, -1.0, X, WP, msz, m, 1, WP, Y, WP, R, 16, sR);
        jb.matvec(1, WP, WP, -1.0, X, WP, msz, m, 1, WP, Y, WP, ev, 1, WP);
        jb.copy(WP, 15, nullptr, 0, 0, R + 1, 16, sR);
        jb.launch(n, st);
    }
    if (gf_dense_solve_logdet((int)n, WP, 16, A, R, ld, st)) return -1;
    {   // u1 = G v  (G symmetric)
        JobBuilder jb;
        jb.matvec(1, WP, WP, 1.0, G, WP, msz, R, 16, sR, nullptr, 0, u1, 1, WP);
        jb.launch(n, st);
    }
    {   // every pivot of the chunk positive?  X = R R^T (rank-revealing), M = I - R^T G R, Cholesky attempt on M;
        // a failed attempt turns the chunk's log det correction into NaN  (T reuses A: the solve is done with it)
        double *Tm = A;
        const size_t lds = sizeof(double) * ((size_t)WP * (WP + 1) / 2 + 2 * (size_t)WP);
        const void *fn[2] = {(const void *)k_pchol, (const void *)k_spd_check};
        if (lds > 64 * 1024 && !gf_internal_lds_opt_in(2, st, fn, 2, sizeof(double) * (192 * 193 / 2 + 2 * 192)))     // (the widest map: 151 KB)
            return gf_internal_error(-1, "chunk corrections: cannot opt in to %lld bytes of LDS", (long long)lds);
        const int threads = WP <= 64 ? 256 : SPD_THREADS;
        hipLaunchKernelGGL(k_pchol, dim3((unsigned)(B * count)), dim3(threads), lds, st, P, first, count, WP, X, Rt);
        {
            JobBuilder jb;
            jb.gemm(0, 1, EPI_PLAIN, WP, WP, WP, G, WP, msz, Rt, WP, msz, nullptr, 0, 0, Tm, WP, msz);
            jb.launch(n, st);
        }
        {
            JobBuilder jb;
            jb.gemm(0, 0, EPI_IMINUS, WP, WP, WP, Rt, WP, msz, Tm, WP, msz, nullptr, 0, 0, Mx, WP, msz);
            jb.launch(n, st);
        }
        hipLaunchKernelGGL(k_spd_check, dim3((unsigned)(B * count)), dim3(threads), lds, st, P, first, count, WP, Mx, ld);
    }
    hipLaunchKernelGGL(k_corr_finish, dim3(B), dim3(256), 0, st, P, first, count, WP, Y, m, ev, u1, ld, acc);
    return 0;
}

bool pow2(int x) { return x >= 1 && (x & (x - 1)) == 0; }

template <int NT>
void launch_gram(int B, int64_t N, int64_t L, int nch, int ch0, int nsel, int P, int CP, const double *h,
                 const double *dbar, const double *zbar, double *G, double *m, hipStream_t st) {
    // few chunks: two workgroups per chunk (each stages all rows of h, the units are dealt over both)
    const int slots = B * nsel;
    constexpr int NW = GramPlan<NT>::NW;
    if (slots < 200 && NT >= 4)
        hipLaunchKernelGGL((k_wide_gram<NT, 2>), dim3(slots, 2), dim3(64 * NW), 0, st, N, L, nch, ch0, nsel, P, CP,
                           h, dbar, zbar, G, m);
    else
        hipLaunchKernelGGL((k_wide_gram<NT, 1>), dim3(slots, 1), dim3(64 * NW), 0, st, N, L, nch, ch0, nsel, P, CP,
                           h, dbar, zbar, G, m);
}

int dispatch_gram(int WP, int B, int64_t N, int64_t L, int nch, int ch0, int nsel, int P, int CP, const double *h,
                  const double *dbar, const double *zbar, double *G, double *m, hipStream_t st) {
    if (nsel < 1) return 0;
    switch (WP / 16) {
#define GF_GR(n) case n: launch_gram<n>(B, N, L, nch, ch0, nsel, P, CP, h, dbar, zbar, G, m, st); return 0;
        GF_GR(4) GF_GR(5) GF_GR(6) GF_GR(7) GF_GR(8) GF_GR(9) GF_GR(10) GF_GR(11)
#undef GF_GR
    }
    return -1;
}

}  // namespace

// ==================================================================================================
// C-ABI
// ==================================================================================================
extern "C" {

int gf_bgemm(int batch, int trans_a, int trans_b, int M, int N, int K,
             const double *A, int lda, int64_t stride_a, const double *B, int ldb, int64_t stride_b,
             const double *D, int ldd, int64_t stride_d, double *C, int ldc, int64_t stride_c, void *stream) {
    if (batch < 1 || M < 1 || N < 1 || K < 1)
        return gf_internal_error(-1, "gf_bgemm: empty problem (batch=%d, M=%d, N=%d, K=%d)", batch, M, N, K);
    if (!A || !B || !C) return gf_internal_error(-1, "gf_bgemm: null pointer");
    if (batch > 65535) return gf_internal_error(-1, "gf_bgemm: batch=%d unsupported (max 65535)", batch);
    JobBuilder jb;
    jb.gemm(trans_a ? 1 : 0, trans_b ? 1 : 0, D ? EPI_ADD : EPI_PLAIN, M, N, K, A, lda, stride_a, B, ldb, stride_b,
            D, ldd, stride_d, C, ldc, stride_c);
    jb.launch(batch, (hipStream_t)stream);
    return gf_internal_check_launch("gf_bgemm");
}

int gf_dense_width(int W) { return (W + 15) / 16 * 16; }

int64_t gf_lft_tree_work(int B, int P, int WP) {
    if (B < 1 || !pow2(P) || WP < 16 || WP > 192 || (WP & 15)) return -1;
    return (int64_t)TreePlan(B, P, WP).total;
}

int gf_lft_tree_scan(int B, int P, int WP, const double *Phi, const double *G, const double *Xbar,
                     const double *Ybar, const double *m, double *X_start, double *Y_start, double *work,
                     void *stream) {
    if (B < 1 || !pow2(P)) return gf_internal_error(-1, "gf_lft_tree_scan: P=%d must be a power of two (B=%d)", P, B);
    if (WP < 16 || WP > 192 || (WP & 15))
        return gf_internal_error(-1, "gf_lft_tree_scan: WP=%d must be a multiple of 16 in 16..192", WP);
    if ((long long)B * P / 2 > 65535) return gf_internal_error(-1, "gf_lft_tree_scan: too many maps (B*P=%lld)", (long long)B * P);
    if (!Phi || !G || !Xbar || !Ybar || !m || !X_start || !Y_start || (P > 2 && !work))
        return gf_internal_error(-1, "gf_lft_tree_scan: null pointer");
    TreePlan T(B, P, WP);
    Maps M0;
    M0.Ph = const_cast<double *>(Phi); M0.G = const_cast<double *>(G); M0.X = const_cast<double *>(Xbar);
    M0.Y = const_cast<double *>(Ybar); M0.m = const_cast<double *>(m);
    return tree_scan(T, M0, X_start, Y_start, work, (hipStream_t)stream);
}

int gf_wide_gram(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count, int P, int Jc,
                 const double *h, const double *dbar, const double *zbar, double *G_out, double *m_out, void *stream) {
    const int W = 2 * Jc, WP = gf_dense_width(W);
    const int CP = gf_fused_row_stride(0, Jc);
    if (B < 1 || N < 1) return gf_internal_error(-1, "gf_wide_gram: empty problem (N=%lld)", (long long)N);
    if (W <= 63 || CP < 0 || WP > 176) return gf_internal_error(-1, "gf_wide_gram: width %d unsupported (64..176)", W);
    if (nch < 1 || chunk_len < 1 || (int64_t)nch * chunk_len < N || (int64_t)(nch - 1) * chunk_len >= N || P < nch)
        return gf_internal_error(-1, "gf_wide_gram: bad chunking (chunk_len=%lld, nch=%d, P=%d)", (long long)chunk_len, nch, P);
    if (chunk_first < 0 || chunk_count < 0 || chunk_first + chunk_count > nch)
        return gf_internal_error(-1, "gf_wide_gram: bad chunk range (first=%d, count=%d)", chunk_first, chunk_count);
    if (!h || !dbar || !zbar || !G_out || !m_out) return gf_internal_error(-1, "gf_wide_gram: null pointer");
    if (dispatch_gram(WP, B, N, chunk_len, nch, chunk_first, chunk_count, P, CP, h, dbar, zbar, G_out, m_out,
                      (hipStream_t)stream))
        return gf_internal_error(-1, "gf_wide_gram: internal dispatch error");
    return gf_internal_check_launch("gf_wide_gram");
}

static int tree_P(int nch) { int P = 1; while (P < nch) P *= 2; return P; }

int64_t gf_wide_combine_work(int B, int nch, int Jc) {
    const int W = 2 * Jc, WP = gf_dense_width(W);
    if (B < 1 || nch < 1 || W <= 63 || WP > 176) return -1;
    const int P = tree_P(nch);
    const size_t n = (size_t)B * P, msz = (size_t)WP * WP;
    const size_t tree = TreePlan(B, P, WP).total, corr = corrections_work(n, WP);
    return (int64_t)(n * (3 * msz + 2 * WP) + n * (msz + WP) + (tree > corr ? tree : corr));
}

int64_t gf_chunk_corrections_work(int B, int nch) { return (B < 1 || nch < 1) ? -1 : (int64_t)corrections_work((size_t)B * nch, 64); }

int gf_chunk_corrections(int B, int nch, int W, int chunk_first, int chunk_count, const double *S_state,
                         const double *F_state, const double *G, const double *m, double *acc, double *work,
                         void *stream) {
    if (B < 1 || nch < 1) return gf_internal_error(-1, "gf_chunk_corrections: empty problem (B=%d, nch=%d)", B, nch);
    if (W < 1 || W > 63) return gf_internal_error(-1, "gf_chunk_corrections: width %d unsupported (1..63)", W);
    if (chunk_first < 0 || chunk_count < 0 || chunk_first + chunk_count > nch)
        return gf_internal_error(-1, "gf_chunk_corrections: bad chunk range (first=%d, count=%d)", chunk_first, chunk_count);
    if ((long long)B * nch > 65535) return gf_internal_error(-1, "gf_chunk_corrections: too many chunks (B*nch=%lld)", (long long)B * nch);
    if (!S_state || !F_state || !G || !m || !acc || !work) return gf_internal_error(-1, "gf_chunk_corrections: null pointer");
    // the 64 x 64 state slots are stored [column][row]: X and G are symmetric, so they ARE dense row-major maps
    if (chunk_corrections_small(B, nch, W, chunk_first, chunk_count, S_state, F_state, G, m, acc, work, (hipStream_t)stream))
        return -1;
    return gf_internal_check_launch("gf_chunk_corrections");
}

int gf_wide_combine(int B, int64_t N, int64_t chunk_len, int nch, int Jc, const double *h, const double *dbar,
                    const double *zbar, const double *Phi_state, double *S_state, double *acc, double *work,
                    void *stream) {
    const int W = 2 * Jc, WP = gf_dense_width(W);
    const int CP = gf_fused_row_stride(0, Jc);
    if (B < 1 || N < 1) return gf_internal_error(-1, "gf_wide_combine: empty problem (N=%lld)", (long long)N);
    if (W <= 63 || CP < 0 || WP > 176) return gf_internal_error(-1, "gf_wide_combine: width %d unsupported (64..176)", W);
    if (nch < 2 || chunk_len < 1 || (int64_t)nch * chunk_len < N || (int64_t)(nch - 1) * chunk_len >= N)
        return gf_internal_error(-1, "gf_wide_combine: bad chunking (chunk_len=%lld, nch=%d)", (long long)chunk_len, nch);
    if (!h || !dbar || !zbar || !Phi_state || !S_state || !work) return gf_internal_error(-1, "gf_wide_combine: null pointer");
    const int P = tree_P(nch);
    if ((long long)B * P / 2 > 65535) return gf_internal_error(-1, "gf_wide_combine: too many chunks (B*P=%lld)", (long long)B * P);
    const int RP = (int)(gf_fused_state_size(0, Jc) / CP);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * P, msz = (size_t)WP * WP;
    Maps M0;
    M0.Ph = work; M0.G = work + n * msz; M0.X = work + 2 * n * msz; M0.Y = work + 3 * n * msz; M0.m = M0.Y + n * WP;
    double *Xs = M0.m + n * WP, *Ys = Xs + n * msz, *tw = Ys + n * WP;
    hipLaunchKernelGGL(k_lft_pack, dim3((unsigned)n, (unsigned)((WP + 31) / 32)), dim3(256), 0, st, nch, P, W, WP, CP, RP, S_state, Phi_state,
                       M0.Ph, M0.X, M0.Y, M0.G, M0.m);
    // (the maps of the first and the last chunk are not needed for the start states: see k_lft_pack; with
    // `acc` the last chunk's Gram sums are: its log-likelihood correction uses them)
    if (dispatch_gram(WP, B, N, chunk_len, nch, 1, acc ? nch - 1 : nch - 2, P, CP, h, dbar, zbar, M0.G, M0.m, st))
        return gf_internal_error(-1, "gf_wide_combine: internal dispatch error");
    TreePlan T(B, P, WP);
    if (tree_scan(T, M0, Xs, Ys, tw, st)) return -1;
    if (acc) {      // log-likelihood without a final pass: corrections of the chunks 1 .. nch - 1 (chunk 0 starts from zero)
        if ((long long)B * P > 65535) return gf_internal_error(-1, "gf_wide_combine: too many chunks for the corrections (B*P=%lld)", (long long)B * P);
        if (chunk_corrections(B, P, WP, 1, nch - 1, Xs, Ys, M0.G, M0.m, acc, tw, st)) return -1;
    }
    hipLaunchKernelGGL(k_lft_unpack, dim3((unsigned)(B * nch), 8), dim3(256), 0, st, nch, P, W, WP, CP, RP, Xs, Ys, S_state);
    return gf_internal_check_launch("gf_wide_combine");
}

}  // extern "C"
