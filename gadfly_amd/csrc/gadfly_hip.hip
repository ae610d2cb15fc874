// gadfly_hip.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4) + the C-ABI of
// include/gadfly_hip.h.  float64 throughout; wave = 64 lanes.
//
// What is replaced: the celerite2 C++ driver routines behind gadfly's GaussianProcess
// (/root/reference/gadfly/gp.py:202, :232, :327, :350, :370, :391).
// The arithmetic follows SURVEY.md Appendix A.4-A.8.
//
// Kernel inventory
//   k_build        A.4 generator rows U, V (+ a, + propagator rows P)         HBM/VALU (sincos, exp)
//   k_factor       A.5 LDL^T recurrence, state S (W x W) in VGPRs, rows split over NW waves,
//                  optional fused forward solve (A.6) for the log-likelihood path
//   k_reduce_*     sum log d, sum z^2/d (deterministic fixed-shape tree)
//   k_solve_vec    A.6/A.7 sweeps, one right-hand side, state F lane-resident
//   k_solve_rhs    A.6/A.7 sweeps, one lane per right-hand side, generator rows wave-uniform
//   k_gmm_*        A.8 conditional mean at new times
//
// No CUDA compatibility layer, no Triton, no CPU fallback.

#include <hip/hip_runtime.h>
#include <atomic>
#include <utility>
#include <tuple>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "../../include/gadfly_hip.h"
#include "gf_internal.h"

// sincos / exp with <= 1 ulp error for the argument ranges of k_build2 (validated against
// long-double libm by oracle/fastmath_check.c); ~4x fewer instructions than the general OCML
// routines, which carry Payne-Hanek reduction for arbitrary arguments.
#define FM_INLINE __device__ __forceinline__
#include "fastmath.h"
#include "gf_wave.h"

// fm_exp with its coefficients formed where they are used.  Written as literals, hipcc hoists the fourteen FP64
// coefficients out of any loop that contains a call -- also when the call sits on a reset row, one row in 16 to
// 64 -- and holds them in 28 VGPRs for the whole sweep (k_phi7<60> then spills).  Each coefficient's two words
// XOR an opaque zero taken once per call, so it is not loop-invariant; same bits, same result.  For the sweeps
// whose only transcendental is this one (the transition sweeps); the factor sweeps are faster with the hoisted
// literals (measured: DESIGN.md 2.1e).
template <unsigned long long BITS>
__device__ __forceinline__ double fm_local_constant(const int z) {
    return __hiloint2double((int)(unsigned)(BITS >> 32) ^ z, (int)(unsigned)(BITS & 0xffffffffull) ^ z);
}
#define GF_K(x) fm_local_constant<__builtin_bit_cast(unsigned long long, (double)(x))>(fm_z)
__device__ __forceinline__ double fm_exp_local(const double x) {
    int fm_z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(fm_z));
    if (x < -745.0) return 0.0;
    const double k = rint(x * GF_K(1.44269504088896338700e+00));
    const double r = fma(-k, GF_K(1.90821492927058770002e-10), fma(-k, GF_K(6.93147180369123816490e-01), x));
    double p = GF_K(1.0 / 6227020800.0);
    p = fma(p, r, GF_K(1.0 / 479001600.0));
    p = fma(p, r, GF_K(1.0 / 39916800.0));
    p = fma(p, r, GF_K(1.0 / 3628800.0));
    p = fma(p, r, GF_K(1.0 / 362880.0));
    p = fma(p, r, GF_K(1.0 / 40320.0));
    p = fma(p, r, GF_K(1.0 / 5040.0));
    p = fma(p, r, GF_K(1.0 / 720.0));
    p = fma(p, r, GF_K(1.0 / 120.0));
    p = fma(p, r, GF_K(1.0 / 24.0));
    p = fma(p, r, GF_K(1.0 / 6.0));
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
#undef GF_K

namespace {

thread_local char g_err[512] = "";

int set_err_impl(const char *fmt, const char *a = "", long long x = 0, long long y = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return -1;
}

// conversions in a format literal; set_err() refuses to compile when they do not match the arguments
// (a missing argument would be read from the defaulted `x` as a char*)
constexpr int fmt_conversions(const char *f) {
    int n = 0;
    for (; *f; ++f)
        if (*f == '%') { if (f[1] == '%') ++f; else ++n; }
    return n;
}
#define set_err(fmt, ...)                                                                              \
    (static_cast<void>(sizeof(char[fmt_conversions(fmt) ==                                             \
                                   (int)std::tuple_size<decltype(std::make_tuple(__VA_ARGS__))>::value \
                                   ? 1 : -1])),                                                        \
     set_err_impl(fmt, __VA_ARGS__))

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}
}  // namespace

// the same two services for the library's other translation units (gf_internal.h)
int gf_internal_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int gf_internal_check_launch(const char *what) { return check_launch(what); }

namespace {

// ------------------------------------------------------------------------------------
// wave-level primitives (64 lanes, DPP; no LDS traffic)
// ------------------------------------------------------------------------------------
// A wave-uniform index made opaque to the optimiser (it stays in SGPRs, at no cost): `p[opaque(i)]`
// with a loop-invariant per-lane pointer p is then addressed as p + i inside the branch that uses
// it, instead of becoming one more per-lane 64-bit pointer that loop strength reduction advances
// with a vector add on EVERY iteration, taken or not.
__device__ __forceinline__ size_t opaque_uniform(size_t i) {
    asm volatile("" : "+s"(i));
    return i;
}

// Fixed-shape tree sum over the 64 lanes, result broadcast to every lane (deterministic).
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_get<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);   // row_mirror       -> every lane holds its 16-lane row sum
    // the two broadcasts only matter for row 3 (rows 1, 3 and rows 2, 3 are what LLVM's own
    // reductions enable with row_mask 0xa / 0xc); with all rows enabled and bound_ctrl every lane is
    // written, so no zero-initialised destination (2 v_mov + s_nop per stage) is needed.  The other
    // rows end up with sums nobody reads.
    v += dpp_get<0x142, 0xf>(v);   // row_bcast15: rows 1, 3 += total of rows 0, 2
    v += dpp_get<0x143, 0xf>(v);   // row_bcast31: row 3 += total of rows 0 + 1 -> lane 63 holds the sum
    return read_lane(v, 63);
}

// Two independent sums interleaved (same latency chain as one).
__device__ __forceinline__ void wave_sum2(double &a, double &b) {
    a += dpp_get<0xB1, 0xf>(a);   b += dpp_get<0xB1, 0xf>(b);
    a += dpp_get<0x4E, 0xf>(a);   b += dpp_get<0x4E, 0xf>(b);
    a += dpp_get<0x141, 0xf>(a);  b += dpp_get<0x141, 0xf>(b);
    a += dpp_get<0x140, 0xf>(a);  b += dpp_get<0x140, 0xf>(b);
    a += dpp_get<0x142, 0xf>(a);  b += dpp_get<0x142, 0xf>(b);
    a += dpp_get<0x143, 0xf>(a);  b += dpp_get<0x143, 0xf>(b);
    a = read_lane(a, 63);
    b = read_lane(b, 63);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// queue (s_waitcnt vmcnt(0) before s_barrier), which turns a software prefetch issued before the
// barrier into a wait for HBM at the barrier; kernels whose waves exchange data through LDS alone use
// this instead and keep their global loads in flight.
__device__ __forceinline__ void wg_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Order LDS traffic of ONE wave: lane-form writes before uniform (broadcast) reads.
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in issue order, so wavefront scope (a pure
    // compiler fence, no s_waitcnt) is enough and leaves global prefetches in flight.
    // The fences name the LDS address space only, so global loads stay provably unclobbered
    // (which is what lets hipcc turn wave-uniform row loads into s_load).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// ------------------------------------------------------------------------------------
// K0: generator rows + propagator rows
// ------------------------------------------------------------------------------------
struct BuildArgs {
    int64_t N, n_first;             // rows built: global n_first .. n_first + N - 1
    int Jr, Jc, ld, units;          // units per row = Jr + Jc + (ld - W)
    const double *ar, *cr, *ac, *bc, *cc, *dc, *diag_add;
    const double *t; int64_t t_bs;
    const double *diag; int64_t diag_bs;
    double *a, *U, *V, *P;
};

__global__ void __launch_bounds__(256) k_build(const BuildArgs A) {
    const int b = blockIdx.y;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = id / A.units;
    if (n >= A.N) return;
    const int q = (int)(id - n * A.units);
    const int W = A.Jr + 2 * A.Jc;
    const double *t = A.t + (size_t)b * A.t_bs;
    const int64_t ng = A.n_first + n;                // global row (t, diag are global arrays)
    const double tn = t[ng];
    const double dt = (ng > 0) ? (t[ng - 1] - tn) : 0.0;
    const size_t row = ((size_t)b * A.N + n) * A.ld;
    if (q == 0 && A.a) {
        const double dg = A.diag ? A.diag[(size_t)b * A.diag_bs + ng] : 0.0;
        A.a[(size_t)b * A.N + n] = dg + A.diag_add[b];
    }
    if (q < A.Jr) {                                  // real term: one column
        const double ar = A.ar[(size_t)b * A.Jr + q], cr = A.cr[(size_t)b * A.Jr + q];
        A.U[row + q] = ar;
        A.V[row + q] = 1.0;
        if (A.P) A.P[row + q] = (ng > 0) ? exp(cr * dt) : 1.0;
    } else if (q < A.Jr + A.Jc) {                    // complex term: two columns
        const int k = q - A.Jr;
        const size_t ck = (size_t)b * A.Jc + k;
        const double ac = A.ac[ck], bc = A.bc[ck], cc = A.cc[ck], dc = A.dc[ck];
        const double arg = dc * tn;                  // ONE rounded multiply (parity hazard i)
        double si, co;
        sincos(arg, &si, &co);
        const int j = A.Jr + 2 * k;
        A.U[row + j]     = ac * co + bc * si;
        A.U[row + j + 1] = ac * si - bc * co;
        A.V[row + j]     = co;
        A.V[row + j + 1] = si;
        if (A.P) {
            const double p = (ng > 0) ? exp(cc * dt) : 1.0;
            A.P[row + j] = p;
            A.P[row + j + 1] = p;
        }
    } else {                                         // pad column
        const int j = W + (q - A.Jr - A.Jc);
        A.U[row + j] = 0.0;
        A.V[row + j] = 0.0;
        if (A.P) A.P[row + j] = 1.0;
    }
}

// ------------------------------------------------------------------------------------
// K1: factor (+ fused forward solve).  One workgroup of NW waves per (problem, chunk).
//   thread (wave w, lane l) holds S[i][j] for rows i = w*RB + r (r < RB) and
//   columns j = c*64 + l (c < CT).  Per row n:
//     S   <- P_n (S + d_{n-1} w_{n-1} w_{n-1}^T) P_n      (fused with the mat-vec below)
//     tmp  = u_n^T S      (partial over this wave's rows; reduced across waves through LDS)
//     d_n  = a_n - tmp . u_n        (DPP tree)
//     w_n  = (v_n - tmp) / d_n
//   Row-indexed operands (u_i, p_i, w_i) are wave-uniform: each wave stages the lane-form
//   vectors in its private LDS slice and reads them back as broadcasts.
// ------------------------------------------------------------------------------------
struct FactorArgs {
    int64_t N, chunk_len, n_first;  // n_first: global index of row 0 (tile streaming), for info
    int W, ld;
    const double *a, *U, *V, *P;
    const double *y; int64_t y_bs;
    double *d, *Wm, *z;
    const double *S_in, *F_in;      // per (problem, chunk) state, or NULL = zero
    double *S_out, *F_out;          // or NULL
    int32_t *info;
};

template <int RB, int CT, int NW>
__global__ void __launch_bounds__(64 * NW) k_factor(const FactorArgs A) {
    constexpr int WPC = CT * 64;
    constexpr int RT = RB * NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.y, ch = blockIdx.x;
    const int nch = gridDim.x;
    const int64_t n0 = (int64_t)ch * A.chunk_len;
    const int64_t n1 = (n0 + A.chunk_len < A.N) ? (n0 + A.chunk_len) : A.N;
    const int ld = A.ld;
    const size_t pb = (size_t)b * A.N;
    const double *__restrict__ Ug = A.U + pb * ld;
    const double *__restrict__ Vg = A.V + pb * ld;
    const double *__restrict__ Pg = A.P + pb * ld;
    const double *__restrict__ ag = A.a + pb;
    const double *__restrict__ yg = A.y ? (A.y + (size_t)b * A.y_bs) : nullptr;

    __shared__ double s_part[2][NW][WPC];
    __shared__ double s_vec[NW][3][WPC];
    double *sv_u = s_vec[wave][0], *sv_p = s_vec[wave][1], *sv_w = s_vec[wave][2];

    double S[RB][CT];
    double Fv[CT], wv[CT];
    bool colok[CT];
    int colc[CT];                       // pad lanes read a clamped (valid) column and are masked at use
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        colok[c] = (c * 64 + lane) < ld;
        colc[c] = colok[c] ? (c * 64 + lane) : (ld - 1);
        wv[c] = 0.0;
        Fv[c] = 0.0;
    }
    const size_t sidx = (size_t)b * nch + ch;
    if (A.info[sidx] != 0) return;      // an earlier tile of this problem already failed
    if (A.S_in) {
        const double *Si = A.S_in + sidx * RT * WPC;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) S[r][c] = Si[(size_t)(wave * RB + r) * WPC + c * 64 + lane];
        if (A.F_in) {
#pragma unroll
            for (int c = 0; c < CT; ++c) Fv[c] = A.F_in[sidx * WPC + c * 64 + lane];
        }
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c) S[r][c] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) sv_w[c * 64 + lane] = 0.0;

    double dprev = 0.0, zprev = 0.0;
    int32_t fail = 0;

    // software prefetch of row n0: plain unconditional loads that nothing touches before the row
    // is processed (a `colok ? load : 0` here makes the wave wait for HBM right where it issued the load)
    double un[CT], vn[CT], pn[CT], an, yn;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const size_t o = (size_t)n0 * ld + colc[c];
        un[c] = Ug[o];
        vn[c] = Vg[o];
        pn[c] = Pg[o];
    }
    // the per-row scalars travel in the vector-load queue too (opaque zero lane offset): a scalar load
    // would be waited for at the next LDS access, which shares its counter
    const int vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);
    const double *__restrict__ ys = yg ? yg : ag;          // always a valid address; dropped below
    an = ag[n0 + vz];
    yn = ys[n0 + vz];

    for (int64_t n = n0; n < n1; ++n) {
        double u[CT], v[CT], p[CT];
        const double a_n = an, y_n = yg ? yn : 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            u[c] = colok[c] ? un[c] : 0.0; v[c] = colok[c] ? vn[c] : 0.0; p[c] = colok[c] ? pn[c] : 1.0;
            sv_u[c * 64 + lane] = u[c];
            sv_p[c * 64 + lane] = p[c];
        }
        // prefetch the next row (clamped: the last iteration re-reads its own row)
        const int64_t nn = (n + 1 < n1) ? (n + 1) : n;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const size_t o = (size_t)nn * ld + colc[c];
            un[c] = Ug[o];
            vn[c] = Vg[o];
            pn[c] = Pg[o];
        }
        an = ag[nn + vz];
        yn = ys[nn + vz];

        wave_lds_fence();

        // fused: pending rank-1 update + decay, then the mat-vec partial for this wave's rows
        double acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.0;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = wave * RB + r;
            const double ui = sv_u[i], pi = sv_p[i];
            const double wid = sv_w[i] * dprev;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const double s = (S[r][c] + wid * wv[c]) * (pi * p[c]);
                S[r][c] = s;
                acc[c] = fma(ui, s, acc[c]);
            }
        }
        double tmp[CT];
        if constexpr (NW > 1) {
            const int buf = (int)(n & 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) s_part[buf][wave][c * 64 + lane] = acc[c];
            wg_lds_barrier();           // (not __syncthreads: the next row's prefetch stays in flight)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                double s = s_part[buf][0][c * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) s += s_part[buf][w2][c * 64 + lane];
                tmp[c] = s;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) tmp[c] = acc[c];
        }
        // F_n = P_n (F_{n-1} + w_{n-1} z_{n-1});  two dot products in one DPP tree
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            Fv[c] = p[c] * fma(wv[c], zprev, Fv[c]);
            s1 = fma(u[c], tmp[c], s1);
            s2 = fma(u[c], Fv[c], s2);
        }
        wave_sum2(s1, s2);
        const double dn = a_n - s1;
        const double zn = y_n - s2;
        if (!(dn > 0.0)) {              // uniform across the whole workgroup
            const int64_t ng = A.n_first + n + 1;
            fail = (int32_t)(ng > 0x7fffffff ? 0x7fffffff : ng);
            break;
        }
        const double inv_dn = fast_rcp(dn);     // one reciprocal (<= 1 ulp) instead of CT IEEE divisions
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            wv[c] = (v[c] - tmp[c]) * inv_dn;
            sv_w[c * 64 + lane] = wv[c];
        }
        if (wave == 0) {
            if (A.Wm) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if (colok[c]) A.Wm[(pb + n) * ld + c * 64 + lane] = wv[c];
            }
            if (lane == 0) {
                A.d[pb + n] = dn;
                if (A.z) A.z[pb + n] = zn;
            }
        }
        dprev = dn;
        zprev = zn;
    }

    if (fail && wave == 0 && lane == 0) A.info[sidx] = fail;
    if (fail) return;

    if (A.S_out) {                      // state handed to the next chunk: pending update, no decay
        wave_lds_fence();
        double *So = A.S_out + sidx * RT * WPC;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = wave * RB + r;
            const double wid = sv_w[i] * dprev;
#pragma unroll
            for (int c = 0; c < CT; ++c) So[(size_t)i * WPC + c * 64 + lane] = fma(wid, wv[c], S[r][c]);
        }
        if (A.F_out && wave == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) A.F_out[sidx * WPC + c * 64 + lane] = fma(wv[c], zprev, Fv[c]);
        }
    }
}

// ------------------------------------------------------------------------------------
// v2 log-likelihood path (W <= 64): block-scaled coordinates, one wave per problem.
//
// Within a block of rows [r, r') (r = "reset row") the state is kept in coordinates scaled by
// rho_n = exp(-c (t_n - t_r)):   S_n = rho_n T_n rho_n,  u~_n = u_n o rho_n,  v~_n = v_n / rho_n,
// w~_n = w_n / rho_n.  In these coordinates the celerite recurrence has NO per-row decay:
//     T   <- T + d_{n-1} w~_{n-1} w~_{n-1}^T
//     tmp  = T u~_n ;  d_n = a_n - u~_n . tmp ;  w~_n = (v~_n - tmp) / d_n
//     F~  <- F~ + w~_{n-1} z_{n-1} ;  z_n = y_n - u~_n . F~
// i.e. 2 FMAs per element of T per row instead of an update, two decays and a mat-vec.  At a
// reset row the accumulated decay E = exp(-c (t_r' - t_r)) is applied once (T <- E T E).  This is
// the un-preconditioned celerite (2017) recurrence made safe by bounding the block span:
// reset(n) = (n % 8 == 0) or cmax (t_n - t_{n-1}) > 4  =>  |log rho| <= 28 inside a block.
// k_build2 writes u~, v~ (and E at reset rows); k_factor2 reads the row operands u~_i as
// wave-uniform scalar loads (SGPR operands of v_fma_f64), so only w~ needs an LDS broadcast.
// ------------------------------------------------------------------------------------
// reset(n) = (n % block == 0) or cmax (t_n - t_{n-1}) > gap, with (block - 1) * gap <= 28
constexpr double SC_SPAN = 28.0;

struct Build2Args {
    int64_t N, n_first;
    int Jr, Jc, ld, units, block;   // block: power of two, rows between forced resets
    double gap;                     // cmax * dt above which a row is a reset of its own
    const double *ar, *cr, *ac, *bc, *cc, *dc, *diag_add, *cmax;
    const double *t; int64_t t_bs;
    const double *diag; int64_t diag_bs;
    double *a, *Ut, *Vt, *de;       // de[n] = t_ref(n) - t_ref(n-1) at reset rows, -1 elsewhere
};

__device__ __forceinline__ bool sc_is_reset(const double *t, int64_t g, double cmax, int block,
                                            double gap) {
    return (g & (block - 1)) == 0 || cmax * (t[g] - t[g - 1]) > gap;
}

// latest reset row <= g (bounded walk: g - g % block is always a reset)
__device__ __forceinline__ int64_t sc_ref(const double *t, int64_t g, double cmax, int block,
                                          double gap) {
    while (!sc_is_reset(t, g, cmax, block, gap)) --g;
    return g;
}

constexpr int B2_ROWS = 32;         // rows per workgroup of k_build2

__global__ void __launch_bounds__(256) k_build2(const Build2Args A) {
    const int b = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * B2_ROWS;          // first tile-local row of the block
    const int nrows = (int)((A.N - n0 < B2_ROWS) ? (A.N - n0) : B2_ROWS);
    const int W = A.Jr + 2 * A.Jc;
    const double *t = A.t + (size_t)b * A.t_bs;
    const double cmax = A.cmax[b];

    // phase 1: one thread per row -- reset rule, distance to the block reference, block decay
    __shared__ double s_t[B2_ROWS], s_ds[B2_ROWS];
    if ((int)threadIdx.x < nrows) {
        const int r = threadIdx.x;
        const int64_t g = A.n_first + n0 + r;
        const double tn = t[g];
        const bool reset = sc_is_reset(t, g, cmax, A.block, A.gap);
        const int64_t ref = reset ? g : sc_ref(t, g, cmax, A.block, A.gap);
        const double de = reset ? ((g > 0) ? (tn - t[sc_ref(t, g - 1, cmax, A.block, A.gap)]) : 0.0)
                                : -1.0;
        s_t[r] = tn;
        s_ds[r] = tn - t[ref];                                   // >= 0, 0 at a reset row
        const size_t o = (size_t)b * A.N + n0 + r;
        const double dg = A.diag ? A.diag[(size_t)b * A.diag_bs + g] : 0.0;
        A.a[o] = dg + A.diag_add[b];
        A.de[o] = de;
    }
    __syncthreads();

    // phase 2: one thread per (row, term); the first (ld - W) units of a row also zero a pad column
    const unsigned units = (unsigned)(A.Jr + A.Jc);
    const unsigned total = (unsigned)nrows * units;
    const int npad = A.ld - W;
    for (unsigned idx = threadIdx.x; idx < total; idx += 256) {
        const unsigned r = idx / units;
        const int q = (int)(idx - r * units);
        const double tn = s_t[r], ds = s_ds[r];
        const size_t row = ((size_t)b * A.N + n0 + r) * A.ld;
        if (q < A.Jr) {
            const double ar = A.ar[(size_t)b * A.Jr + q], cr = A.cr[(size_t)b * A.Jr + q];
            const double rho = fm_exp(-cr * ds);
            A.Ut[row + q] = ar * rho;
            A.Vt[row + q] = 1.0 / rho;
        } else {
            const int k = q - A.Jr;
            const size_t ck = (size_t)b * A.Jc + k;
            const double ac = A.ac[ck], bc = A.bc[ck], cc = A.cc[ck], dc = A.dc[ck];
            const double arg = dc * tn;              // ONE rounded multiply (parity hazard i)
            double si, co;
            if (fabs(arg) < FM_SINCOS_RANGE) fm_sincos(arg, &si, &co);
            else sincos(arg, &si, &co);                        // e.g. JD-based time axes
            const double rho = fm_exp(-cc * ds), irho = 1.0 / rho;
            const int j = A.Jr + 2 * k;
            double2 u2, v2;
            u2.x = (ac * co + bc * si) * rho;
            u2.y = (ac * si - bc * co) * rho;
            v2.x = co * irho;
            v2.y = si * irho;
            if ((j & 1) == 0) {                      // 16-byte aligned pair
                *reinterpret_cast<double2 *>(A.Ut + row + j) = u2;
                *reinterpret_cast<double2 *>(A.Vt + row + j) = v2;
            } else {
                A.Ut[row + j] = u2.x; A.Ut[row + j + 1] = u2.y;
                A.Vt[row + j] = v2.x; A.Vt[row + j + 1] = v2.y;
            }
        }
        if (q < npad) {
            A.Ut[row + W + q] = 0.0;
            A.Vt[row + W + q] = 0.0;
        }
    }
    // more pad columns than units (tiny kernels only)
    if (npad > (int)units) {
        for (unsigned idx = threadIdx.x; idx < (unsigned)nrows * (unsigned)npad; idx += 256) {
            const unsigned r = idx / (unsigned)npad;
            const int q = (int)(idx - r * (unsigned)npad);
            const size_t row = ((size_t)b * A.N + n0 + r) * A.ld;
            A.Ut[row + W + q] = 0.0;
            A.Vt[row + W + q] = 0.0;
        }
    }
}

// Sweeps over the ROWS register-resident rows of T with wave-uniform row operands read from
// LDS (xa[i], xw[i]), software-pipelined: batches of 4 rows (2 + 2 ds_read_b128), SW_AHEAD
// batches in flight ahead of the FMAs, so the LDS latency is paid once per sweep and not once
// per read (hipcc otherwise parks every read right in front of its use).
constexpr int SW_BR = 4, SW_AHEAD = 1;
#ifndef GF_CHAIN_PRIO
#define GF_CHAIN_PRIO 1
#endif

// issue the reads of the first SW_AHEAD batches (done early, under the reduction of the
// previous row: the operands do not depend on it)
template <int ROWS>
__device__ __forceinline__ void sweep_preload(double (&ab)[SW_AHEAD + 1][SW_BR],
                                              double (&wb)[SW_AHEAD + 1][SW_BR],
                                              const double *xa, const double *xw) {
    constexpr int NB = ROWS / SW_BR;
#pragma unroll
    for (int k = 0; k < SW_AHEAD && k < NB; ++k) {
#pragma unroll
        for (int r = 0; r < SW_BR; ++r) { ab[k][r] = xa[k * SW_BR + r]; wb[k][r] = xw[k * SW_BR + r]; }
    }
}

//   RESET = false:  T_i += xw_i * q ;  acc += xa_i * T_i            (update + mat-vec)
//   RESET = true :  T_i  = (T_i + xw_i * q) * (xa_i * el)          (fold pending, decay)
template <int ROWS, bool RESET>
__device__ __forceinline__ double sweep_run(double (&T)[ROWS], double (&ab)[SW_AHEAD + 1][SW_BR],
                                            double (&wb)[SW_AHEAD + 1][SW_BR], const double *xa,
                                            const double *xw, const double q, const double el) {
    constexpr int BR = SW_BR, NB = ROWS / BR, AHEAD = SW_AHEAD;
    static_assert(ROWS % BR == 0, "ROWS must be a multiple of 4");
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (k + AHEAD < NB) {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                ab[(k + AHEAD) % (AHEAD + 1)][r] = xa[(k + AHEAD) * BR + r];
                wb[(k + AHEAD) % (AHEAD + 1)][r] = xw[(k + AHEAD) * BR + r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int c = k % (AHEAD + 1);
        if constexpr (RESET) {
#pragma unroll
            for (int r = 0; r < BR; ++r)
                T[k * BR + r] = fma(wb[c][r], q, T[k * BR + r]) * (ab[c][r] * el);
        } else {
            T[k * BR + 0] = fma(wb[c][0], q, T[k * BR + 0]);
            T[k * BR + 1] = fma(wb[c][1], q, T[k * BR + 1]);
            T[k * BR + 2] = fma(wb[c][2], q, T[k * BR + 2]);
            T[k * BR + 3] = fma(wb[c][3], q, T[k * BR + 3]);
            acc0 = fma(ab[c][0], T[k * BR + 0], acc0);
            acc1 = fma(ab[c][1], T[k * BR + 1], acc1);
            acc2 = fma(ab[c][2], T[k * BR + 2], acc2);
            acc3 = fma(ab[c][3], T[k * BR + 3], acc3);
            // pin the partial sums to this batch: LLVM otherwise sinks all the mat-vec FMAs
            // to the end of the sweep and keeps every row operand alive (register blow-up)
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return (acc0 + acc2) + (acc1 + acc3);
}

// Per row n (scaled coordinates; r = v~ - tmp, so w~ = r/d and d w~ = r):
//     sweep :  T_i += r_{n-1,i} * q_{n-1}  (q = r/d, lane-form) ;  tmp = sum_i u~_{n,i} T_i
//     r_n = v~_n - tmp ;  stage r_n and u~_{n+1} in LDS, start reading the next row's operands
//     d_n = a_n - u~_n . tmp ;  z_n = y_n - u~_n . F~ ;  q_n = r_n / d_n
// The division and the DPP reduction sit between two sweeps; staging r (not w~) lets the LDS
// traffic of the next sweep start before they finish.
// Forward solve for free (W < 64): lane 63 is a pad column of T; loading it with F~ (as
// T[i][63] = F~_i) and giving it the multiplier q_63 = z/d makes the sweep's rank-1 update
// perform F~ += r z/d and its mat-vec return u~ . F~ in lane 63's partial sum, so the
// solve needs neither its own recurrence nor a second DPP reduction.
template <int ROWS>
__global__ void __launch_bounds__(64, 2)     // 2 waves/SIMD: one wave alone gets half the VALU issue rate
k_factor2(const int64_t N, const int64_t n_first, const int ld, const int W,
          const double *__restrict__ c_,
          const double *__restrict__ a_, const double *__restrict__ Ut_,
          const double *__restrict__ Vt_, const double *__restrict__ de_,
          const double *__restrict__ y_, const int64_t y_bs,
          double *__restrict__ d_, double *__restrict__ z_,
          double *__restrict__ S_state, double *__restrict__ F_state,
          int32_t *__restrict__ info) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (info[b] != 0) return;
    const size_t pb = (size_t)b * N;
    const double *__restrict__ ag = a_ + pb;
    const double *__restrict__ Ug = Ut_ + pb * ld;
    const double *__restrict__ Vg = Vt_ + pb * ld;
    const double *__restrict__ eg = de_ + pb;
    const double *__restrict__ yg = y_ + (size_t)b * y_bs;
    double *__restrict__ dg = d_ + pb;
    double *__restrict__ zg = z_ + pb;
    // state layout [lane][64 rows]: one base address per lane, immediate offsets per row
    double *__restrict__ Sg = S_state + (size_t)b * (64 * 64) + (size_t)lane * 64;
    double *__restrict__ Fg = F_state + (size_t)b * 64;

    __shared__ double s_w[64];      // r_{n-1}  (pending rank-1 update, row form)
    __shared__ double s_u[64];      // u~_n
    __shared__ double s_e[64];      // block decay E at reset rows
    const bool colok = lane < ld;
    const double cj = (lane < W) ? c_[(size_t)b * W + lane] : 0.0;

    // PADF: the forward solve rides in pad lane 63 (needs W < 64, i.e. ld <= 64 and lane 63 unused)
    const bool PADF = (ROWS < 64) || (W < 64);   // compile-time true except for ROWS == 64
    const bool fl = PADF && lane == 63;
    double T[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) T[i] = Sg[i];
    double Fv = Fg[lane];                       // only used when !PADF
    if (PADF) {
        // F~ (row form in F_state) becomes lane 63's column of T
#pragma unroll
        for (int i = 0; i < ROWS; ++i) T[i] = fl ? Fg[i] : T[i];
    }
    double rprev = 0.0, q = 0.0, zq = 0.0;      // pending: T += r r^T/d, F~ += r z/d
    int32_t fail = 0;

    // Everything a row needs is prefetched two rows ahead, unconditionally (the caller pads
    // every buffer, y included, by two rows/elements): a clamped index makes hipcc give up its
    // scalarisation of uniform loads, and an un-prefetched scalar costs a full memory round
    // trip per row.  Pad lanes (>= ld) re-read column ld-1 and are masked in registers.
    const int lanec = colok ? lane : (ld - 1);
    double ut_n = Ug[lanec], ut_n2 = Ug[ld + lanec];
    double vt_n = Vg[lanec], vt_n2 = Vg[ld + lanec];
    double a_nx = ag[0], y_nx = yg[0], e_nx = eg[0];
    double a_nx2 = ag[1], y_nx2 = yg[1], e_nx2 = eg[1];

    double ab[SW_AHEAD + 1][SW_BR], wb[SW_AHEAD + 1][SW_BR];
    s_w[lane] = 0.0;
    s_u[lane] = colok ? ut_n : 0.0;
    wave_lds_fence();
    sweep_preload<ROWS>(ab, wb, s_u, s_w);

    for (int64_t n = 0; n < N; ++n) {
        const double ut = colok ? ut_n : 0.0, vt = colok ? vt_n : 0.0;
        const double a_n = a_nx, y_n = y_nx, e_n = e_nx;
        ut_n = ut_n2; vt_n = vt_n2; a_nx = a_nx2; y_nx = y_nx2; e_nx = e_nx2;
        ut_n2 = Ug[(size_t)(n + 2) * ld + lanec];
        vt_n2 = Vg[(size_t)(n + 2) * ld + lanec];
        a_nx2 = ag[n + 2];
        y_nx2 = yg[n + 2];
        e_nx2 = eg[n + 2];
        if (e_n >= 0.0) {                   // wave-uniform: reset row
            // fold the pending update, then decay by E = exp(-c * de) (de = t_ref - t_ref_prev)
            const double el = colok ? exp(-cj * e_n) : 1.0;
            s_e[lane] = el;
            wave_lds_fence();
            sweep_preload<ROWS>(ab, wb, s_e, s_w);      // ring reused, re-fetched below
            (void)sweep_run<ROWS, true>(T, ab, wb, s_e, s_w, q, el);
            sweep_preload<ROWS>(ab, wb, s_u, s_w);
            Fv = el * fma(rprev, zq, Fv);
            q = 0.0;
            zq = 0.0;
        }
        const double tmp = sweep_run<ROWS, false>(T, ab, wb, s_u, s_w, q, 0.0);
        const double r = fl ? 0.0 : (vt - tmp); // pad lanes: 0 - 0; lane 63 carries u~ . F~
        Fv = fma(rprev, zq, Fv);
        // stage the next row's operands and start fetching them under the reduction
        wave_lds_fence();
        s_w[lane] = r;
        s_u[lane] = colok ? ut_n : 0.0;
        wave_lds_fence();
        sweep_preload<ROWS>(ab, wb, s_u, s_w);
        double s1 = ut * tmp, s2;
        if (PADF) {                             // wave-uniform
            s1 = wave_sum(s1);                  // ut = 0 in lane 63
            s2 = read_lane(tmp, 63);
        } else {
            s2 = ut * Fv;
            wave_sum2(s1, s2);
        }
        const double dn = a_n - s1;
        const double zn = y_n - s2;
        if (!(dn > 0.0)) {
            const int64_t gg = n_first + n + 1;
            fail = (int32_t)(gg > 0x7fffffff ? 0x7fffffff : gg);
            break;
        }
        const double inv = 1.0 / dn;
        zq = zn * inv;
        q = fl ? zq : r * inv;
        rprev = r;
        if (lane == 0) { dg[n] = dn; zg[n] = zn; }
    }
    if (fail) {
        if (lane == 0) info[b] = fail;
        return;
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const double v = fma(s_w[i], q, T[i]);
        if (fl) Fg[i] = v; else Sg[i] = v;
    }
    if (!PADF) Fg[lane] = fma(rprev, zq, Fv);
}

// ------------------------------------------------------------------------------------
// k_factor3: the fully fused log-likelihood sweep (W < 64).  Same recurrence as k_factor2, but
// the generator rows are never materialised: every row's u~, v~ are produced in registers
// from t_n (theta = d t_n as one rounded multiply, fm_sincos, running products for rho), so an
// evaluation reads only t, y, diag (24 B/row) and writes d, z (16 B/row).  Each lane computes
// sin AND cos of its own phase (lanes 2k, 2k+1 share theta_k), which removes any cross-lane
// exchange:  u~ = (uA * own + uB * other) * rho,  v~ = own / rho  with
//   even column: own = cos, uA = a, uB = +b ;  odd column: own = sin, uA = a, uB = -b ;
//   real column: d = 0 (cos = 1), uA = a_r, uB = 0.
// Between reset rows (anchors: exact fm_sincos, rho = 1) the rho-scaled phasor advances by a cached
// one-cadence rotation with a second-order correction for cadence jitter (RowGen::next).
// Valid while |d t| < 1e12 (fm_sincos's range, FM_SINCOS_RANGE): the caller checks and otherwise
// uses the k_build2 + k_factor2 pair.
//
// Chunk mode (time-parallel evaluation of one series, see k_phi / k_combine): the grid is
// (problem, chunk); block b handles rows [c L, (c+1) L) of problem b / nch with its own state
// slot b, so the same kernel serves the nominal pass (zero start state, r~ rows stored for
// k_phi) and the final pass (true start states from k_combine).
// ------------------------------------------------------------------------------------
struct RowGen {
    // per-lane column constants: u~ = (k1 cu + k2 su), v~ = own / rho^2 with (cu, su) =
    // rho (cos, sin)(d t); lanes 2k, 2k+1 carry the same (cu, su)
    double cj, dj, k1, k2;
    // own-component selectors of v~ as exact 0/1 factors (pad lanes: both 0): two FMAs-worth of
    // VALU work instead of two 64-bit selects, and no lane masks held in SGPRs
    double sel_c, sel_s;
    // wave-uniform thresholds of the row tests, divided out once: a row is a reset of its own when
    // dt > gthr (= gap / cmax), its spacing counts as the cached one when |ddt| < jthr (= 2e-6 / wmax)
    double gthr, jthr, jthr1;       // jthr1 = 1.4e-9 / wmax: below it the step's correction is first order
    int block, sub_mask;            // scaling block length; sub-anchor period - 1
    // far from t = 0 the rotation steps follow celerite2's ROUNDED phases theta_n = fl(d t_n) (see step())
    bool qmode;
    // running state: rho-scaled phasor, 1 / rho^2, cached one-cadence multipliers
    double cu, su, irho2, Er, Ei, G2, dt_ref, dt_last, tref, t_m1;

    __device__ __forceinline__ void init(int lane, int b, int Jr, int Jc, int block_sub, double gap_,
                                         const double *ar_, const double *cr_, const double *ac_,
                                         const double *bc_, const double *cc_, const double *dc_,
                                         const double *cmax_, const double *tg, int64_t n_first) {
        const int W = Jr + 2 * Jc;
        const bool colok = lane < W;
        bool is_sin = false;
        cj = 0.0; dj = 0.0; k1 = 0.0; k2 = 0.0;
        if (lane < Jr) {
            cj = cr_[(size_t)b * Jr + lane];
            k1 = ar_[(size_t)b * Jr + lane];
        } else if (colok) {
            const int k = (lane - Jr) >> 1;
            const size_t ck = (size_t)b * Jc + k;
            is_sin = ((lane - Jr) & 1) != 0;
            cj = cc_[ck];
            dj = dc_[ck];
            // cos column: a cos + b sin ;  sin column: a sin - b cos
            k1 = is_sin ? -bc_[ck] : ac_[ck];
            k2 = is_sin ? ac_[ck] : bc_[ck];
        }
        sel_c = (colok && !is_sin) ? 1.0 : 0.0;
        sel_s = (colok && is_sin) ? 1.0 : 0.0;
        const double cmax = cmax_[b];
        const double wmax = read_lane(wave_max(fmax(cj, fabs(dj))), 0);
        gthr = read_lane(gap_ / cmax, 0);                       // uniform: live in SGPRs
        jthr = read_lane(2e-6 / wmax, 0);
        // phases beyond 4e6 rad at the first row (a time axis that does not start near zero: 5e9 rad on a JD-based
        // one): their rounding, half an ulp of theta, is no longer negligible against the accuracy target
        qmode = __builtin_amdgcn_readfirstlane((int)(wmax * fabs(tg[0]) > 4.0e6)) != 0;
        jthr1 = qmode ? 0.0 : read_lane(1.4e-9 / wmax, 0);      // (the first-order step assumes |x| < 1.4e-9)
        block = block_sub & 0xff;           // (block <= 64) | (sub-anchor period << 8)
        sub_mask = ((block_sub >> 8) & 0xff) - 1;  // (bit 30: zero start, see the sweeps)
        cu = 1.0; su = 0.0; irho2 = 1.0; Er = 1.0; Ei = 0.0; G2 = 1.0; dt_ref = -1.0; dt_last = -2.0;
        // reference time of the block that precedes the first row (for its decay); tg points
        // at the first row, earlier rows are at negative indices
        tref = tg[0];
        if (n_first > 0) {
            int64_t g = -1;
            while (!(((n_first + g) & (block - 1)) == 0 || (tg[g] - tg[g - 1]) > gthr)) --g;
            tref = tg[g];
        }
        t_m1 = (n_first > 0) ? tg[-1] : tg[0];
    }

    // generate global row g at time tn: u~, v~, reset flag and (for reset rows) the decay span.
    // Reset rows (every `block` <= 64 rows, and after gaps) are anchors: rho = 1 and the phase is
    // theta = d t_n as ONE rounded multiply through fm_sincos.  Between anchors the phasor
    // advances by the cached one-cadence multiplier exp((-c + i d) dt_ref), corrected to second
    // order for the deviation of this row's spacing from dt_ref (rounding-level jitter of a
    // regular cadence: |(c, d) ddt| < 2e-6 => truncation < 2e-18 per row); any other spacing
    // recomputes the phasor exactly (and re-caches the multiplier once the new spacing repeats).
    // Per-row rounding accumulates over at most period - 1 rows (sub-anchors).
    // The pieces of next(): kind of the row at time tn (0 = plain rotation step, 1 = the cached
    // multipliers must be refreshed first, 2 = reset row / anchor); wave-uniform.
    __device__ __forceinline__ int peek(const double tn, const int64_t g, double &dt, double &ddt) const {
        dt = tn - t_m1;
        ddt = dt - dt_ref;
        if (((g & (block - 1)) == 0) || (dt > gthr)) return 2;
        if ((g & sub_mask) == 0) return 3;                      // exact phasor again, same scaling
        return (fabs(ddt) < jthr) ? 0 : 1;
    }
    // Sub-anchor: every `period` rows (gf_set_generator_period: 1, 2, 4, ... 64) the phasor is
    // recomputed exactly (theta = d t_n as one rounded multiply, rho = exp(-c (t_n - t_ref))), so
    // the rotation's rounding -- coherent within a segment, the cached multiplier carries one
    // fixed 0.5-ulp error -- accumulates over at most period - 1 steps.  Measured against the
    // 80-bit recurrence (tests/test_gpu_random.py seeds, condition 4e5): period 1 = exact
    // generation every row 2e-10 (the float64 class), 4: 2e-9, 16: 1e-8, 64: several 1e-8.
    __device__ __forceinline__ void subanchor(const double tn) {
        t_m1 = tn;
        double si, co;
        fm_sincos(dj * tn, &si, &co);
        const double rho = fm_exp(-cj * (tn - tref));
        cu = rho * co;
        su = rho * si;
        irho2 = fast_rcp(rho * rho);
    }
    __device__ __forceinline__ void anchor(const double tn, const int64_t g, double &de) {
        de = (g > 0) ? (tn - tref) : 0.0;
        tref = tn;
        t_m1 = tn;
        fm_sincos(dj * tn, &su, &cu);               // real columns: d = 0 -> (1, 0)
        irho2 = 1.0;
    }
    __device__ __forceinline__ void refresh(const double dt) {
        double si, co;
        fm_sincos(dj * dt, &si, &co);
        const double pr = fm_exp(-cj * dt);
        Er = pr * co;
        Ei = pr * si;
        G2 = fast_rcp(pr * pr);
        dt_ref = read_lane(dt, 0);                  // uniform: keep it in SGPRs
    }
    __device__ __forceinline__ void step(const double tn, const double ddt) {
        const double tp = t_m1;
        t_m1 = tn;
        const double xr = -cj * ddt;
        double xi;
        if (qmode) {
            // celerite2 takes cos / sin of theta_n = fl(d t_n): at 5e9 rad that is the true phase +- 5e-7 rad,
            // row by row.  The step must land on THAT phase: the phasor stands at theta_{n-1}, the cached
            // multiplier turns it by fl(d dt_ref) (what refresh() took the sincos of), so the correction is the
            // difference of the two ROUNDED products -- exact (neighbouring values) -- minus that angle.  (The
            // decay needs nothing of the kind: celerite2's exp(-c (t_n - t_{n-1})) sees the time difference too.)
#pragma clang fp contract(off)
            const double th_n = dj * tn, th_p = dj * tp;        // two rounded products, as celerite2 forms them
            xi = (th_n - th_p) - dj * dt_ref;
        } else {
            xi = dj * ddt;
        }
        double Mr, Mi, g2;
        if (fabs(ddt) < jthr1) {
            // |x| < 1.4e-9 (the rounding jitter of a regular cadence): exp(x) = 1 + x to 1e-18
            Mr = fma(-Ei, xi, fma(Er, xr, Er));                     // E (1 + x)
            Mi = fma(Ei, xr, fma(Er, xi, Ei));
            g2 = fma(G2 * -2.0, xr, G2);                            // G2 (1 - 2 xr)
        } else {
            const double qr = 1.0 + fma(0.5, fma(xr, xr, -xi * xi), xr);    // exp(xr + i xi), 2nd order
            const double qi = fma(xr, xi, xi);
            Mr = fma(Er, qr, -Ei * qi);
            Mi = fma(Er, qi, Ei * qr);
            const double x2 = -2.0 * xr;
            g2 = G2 * (1.0 + fma(0.5 * x2, x2, x2));
        }
        const double c2 = fma(cu, Mr, -su * Mi);
        su = fma(cu, Mi, su * Mr);
        cu = c2;
        irho2 *= g2;
    }
    __device__ __forceinline__ void emit(double &ut, double &vt) const {
        ut = fma(k1, cu, k2 * su);                  // pad lanes: k1 = k2 = 0
        vt = fma(sel_s, su, sel_c * cu) * irho2;
    }
    __device__ __forceinline__ void next(const double tn, const int64_t g, double &ut, double &vt,
                                         bool &rst, double &de) {
        advance(tn, g, rst, de);
        emit(ut, vt);
    }
    __device__ __forceinline__ void advance(const double tn, const int64_t g, bool &rst, double &de) {
        // two wave-uniform tests on the common path (reset? plain step?); the rare cases are told
        // apart inside the rare branch
        const double dt = tn - t_m1;
        double ddt = dt - dt_ref;
        rst = ((g & (block - 1)) == 0) || (dt > gthr);
        de = -1.0;
        if (__builtin_expect(rst, 0)) {
            anchor(tn, g, de);
        } else if (__builtin_expect(((g & sub_mask) != 0) && (fabs(ddt) < jthr), 1)) {
            step(tn, ddt);
        } else {
            // sub-anchor row, or a spacing that differs from the cached one: exact phasor (an
            // irregular cadence thus never accumulates rotation steps: same cost as refreshing
            // the multiplier, exact accuracy); the cached multipliers move to a new spacing
            // only once it repeats (a lasting change of cadence, not a single odd row)
            subanchor(tn);
            if (!(fabs(ddt) < jthr)) {
                if (fabs(dt - dt_last) < jthr) refresh(dt);
                dt_last = read_lane(dt, 0);
            }
        }
    }

    // Park / restore the ten per-lane doubles in LDS (buf[10][64]): a kernel that needs its
    // registers while no row is being generated spills them here instead of to scratch.
    __device__ __forceinline__ void park(double *buf, int lane) const {
        buf[0 * 64 + lane] = cj; buf[1 * 64 + lane] = dj; buf[2 * 64 + lane] = k1; buf[3 * 64 + lane] = k2;
        buf[4 * 64 + lane] = cu; buf[5 * 64 + lane] = su; buf[6 * 64 + lane] = irho2;
        buf[7 * 64 + lane] = Er; buf[8 * 64 + lane] = Ei; buf[9 * 64 + lane] = G2;
    }
    __device__ __forceinline__ void park_state(double *buf, int lane) const {       // what next() changes
        buf[4 * 64 + lane] = cu; buf[5 * 64 + lane] = su; buf[6 * 64 + lane] = irho2;
        buf[7 * 64 + lane] = Er; buf[8 * 64 + lane] = Ei; buf[9 * 64 + lane] = G2;
    }
    __device__ __forceinline__ void unpark(const double *buf, int lane) {
        cj = buf[0 * 64 + lane]; dj = buf[1 * 64 + lane]; k1 = buf[2 * 64 + lane]; k2 = buf[3 * 64 + lane];
        cu = buf[4 * 64 + lane]; su = buf[5 * 64 + lane]; irho2 = buf[6 * 64 + lane];
        Er = buf[7 * 64 + lane]; Ei = buf[8 * 64 + lane]; G2 = buf[9 * 64 + lane];
    }
};

template <int ROWS>
__global__ void __launch_bounds__(64, 2)
k_factor3(const int64_t N, const int64_t n_first, const int64_t chunk_len, const int nch,
          const int ch0, const int nsel, const int Jr, const int Jc, const int block_sub, const double gap,
          const double *__restrict__ ar_, const double *__restrict__ cr_,
          const double *__restrict__ ac_, const double *__restrict__ bc_,
          const double *__restrict__ cc_, const double *__restrict__ dc_,
          const double *__restrict__ diag_add_, const double *__restrict__ cmax_,
          const double *__restrict__ t_, const int64_t t_bs,
          const double *__restrict__ diag_, const int64_t diag_bs,
          const double *__restrict__ y_, const int64_t y_bs,
          double *__restrict__ d_, double *__restrict__ z_, double *__restrict__ r_out,
          double *__restrict__ Ut_out, double *__restrict__ Wt_out, double *__restrict__ de_out,
          double *__restrict__ S_state, double *__restrict__ F_state,
          int32_t *__restrict__ info) {
    const int lane = threadIdx.x;
    const int sel = blockIdx.x;                     // chunks ch0 .. ch0 + nsel - 1 of every problem
    const int pr = sel / nsel, ch = ch0 + (sel - pr * nsel);
    const int b = pr * nch + ch;                    // state slot = problem * nch + chunk
    if (info[b] != 0) return;
    const int64_t c0 = (int64_t)ch * chunk_len;     // first row of the chunk within this call
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const int64_t g0 = n_first + c0;                // global index of the chunk's first row
    const size_t pb = (size_t)pr * N + c0;
    const double *__restrict__ tg = t_ + (size_t)pr * t_bs + g0;
    const double *__restrict__ yg = y_ + (size_t)pr * y_bs + g0;
    // per-row diagonal: always read (through an address that is valid either way, so the loads stay
    // unconditional scalar loads) and dropped by a scalar select when there is none
    const bool has_g = diag_ != nullptr;
    const double *__restrict__ gg = has_g ? diag_ + (size_t)pr * diag_bs + g0 : yg;
    double *__restrict__ dg = d_ + pb;
    double *__restrict__ zg = z_ + pb;
    // chunk-mode row stores: loop-invariant per-lane pointers, indexed with opaque_uniform(row)
    double *__restrict__ rg = r_out ? r_out + pb * 64 + lane : nullptr;
    double *__restrict__ ug = Ut_out ? Ut_out + pb * 64 + lane : nullptr;   // stored factor:
    double *__restrict__ wg = Wt_out ? Wt_out + pb * 64 + lane : nullptr;   // u~, w~ = r/d rows
    double *__restrict__ eg = de_out ? de_out + pb : nullptr;               // reset spans (-1: none)
    double *__restrict__ Sg = S_state + (size_t)b * (64 * 64) + (size_t)lane * 64;
    double *__restrict__ Fg = F_state + (size_t)b * 64;
    const double diag_add = diag_add_[pr];

    RowGen G;
    G.init(lane, pr, Jr, Jc, block_sub, gap, ar_, cr_, ac_, bc_, cc_, dc_, cmax_, tg, g0);
    const double cj = G.cj;

    __shared__ double s_w[64];      // r_{n-1}  (pending rank-1 update, row form)
    __shared__ double s_u[64];      // u~_n
    __shared__ double s_e[64];      // block decay E at reset rows
    const bool fl = lane == 63;     // pad lane carrying the forward solve (W < 64)
    // exact 0/1 factors for "every lane but 63" / "lane 63 only": one multiply (FMA) where a
    // 64-bit select costs two instructions on the row's serial chain
    const double not63 = fl ? 0.0 : 1.0, is63 = fl ? 1.0 : 0.0;

    double T[ROWS];
    const bool zero_start = (block_sub >> 30) & 1;  // nominal pass: the state slots are outputs only
#pragma unroll
    for (int i = 0; i < ROWS; ++i) T[i] = zero_start ? 0.0 : (fl ? Fg[i] : Sg[i]);
    double q = 0.0;
    int32_t fail = 0;

    // rows are generated one row ahead of their sweep; t is prefetched two further rows ahead, y and
    // diag one (the caller pads t, y, diag by three elements; a row takes far longer than a scalar load)
    double t_n1 = tg[1], t_n2 = tg[2];
    double y_n = yg[0], y_n1 = yg[1];
    double g_n = gg[0], g_n1 = gg[1];
    double ut, vt, de;
    bool rst;
    G.next(tg[0], g0, ut, vt, rst, de);

    double ab[SW_AHEAD + 1][SW_BR], wb[SW_AHEAD + 1][SW_BR];
    s_w[lane] = 0.0;
    s_u[lane] = ut;
    wave_lds_fence();
    sweep_preload<ROWS>(ab, wb, s_u, s_w);

    for (int64_t n = 0; n < rows; ++n) {
        const double a_n = (has_g ? g_n : 0.0) + diag_add, yy = y_n;
        const double ut_c = ut, vt_c = vt;
        if (eg && lane == 0) eg[n] = rst ? de : -1.0;
        if (rst) {                          // wave-uniform: fold the pending update, then decay
            const double el = fm_exp(-cj * de);     // pad lanes: cj = 0 -> 1
            s_e[lane] = el;
            wave_lds_fence();
            // the operand ring is reused (its preloaded rows are simply fetched again below):
            // keeping a second ring alive across this branch costs 48 VGPRs on every row
            sweep_preload<ROWS>(ab, wb, s_e, s_w);
            (void)sweep_run<ROWS, true>(T, ab, wb, s_e, s_w, q, el);
            sweep_preload<ROWS>(ab, wb, s_u, s_w);
            q = 0.0;
        }
        __builtin_amdgcn_s_setprio(0);
        const double tmp = sweep_run<ROWS, false>(T, ab, wb, s_u, s_w, q, 0.0);
        // the serial part of the row (reduction, reciprocal, next row's operands) wins the issue
        // arbitration against the partner wave's FMA stream: its instructions are dependent and
        // each lost slot lengthens the chain, the partner's are not
        __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);
        const double r = (vt_c - tmp) * not63;
        // next row's operands: generated here, where the operand ring is dead (register budget)
        G.next(t_n1, g0 + n + 1, ut, vt, rst, de);
        t_n1 = t_n2; y_n = y_n1; g_n = g_n1;
        t_n2 = tg[n + 3];
        y_n1 = yg[n + 2];
        g_n1 = gg[n + 2];
        wave_lds_fence();
        s_w[lane] = r;
        s_u[lane] = ut;
        wave_lds_fence();
        sweep_preload<ROWS>(ab, wb, s_u, s_w);
        const double s1 = wave_sum(ut_c * tmp);     // u~ = 0 in lane 63
        const double s2 = read_lane(tmp, 63);       // u~ . F~
        const double dn = a_n - s1;
        const double zn = yy - s2;
        if (!(dn > 0.0)) {
            const int64_t gf = g0 + n + 1;
            fail = (int32_t)(gf > 0x7fffffff ? 0x7fffffff : gf);
            break;
        }
        const double inv = fast_rcp(dn);
        q = fma(zn, is63, r) * inv;                 // lane 63: z / d (r is 0 there)
        const size_t ro = opaque_uniform((size_t)n * 64);
        if (lane < ROWS || wg) {                    // (nominal passes store the first ROWS columns only: k_factor7)
            if (rg) rg[ro] = r;                     // r~ rows for k_phi (chunk mode)
            if (ug) ug[ro] = ut_c;
            if (wg) wg[ro] = fl ? 0.0 : q;
        }
        if (lane == 0) { dg[n] = dn; zg[n] = zn; }
    }
    if (fail) {
        if (lane == 0) info[b] = fail;
        return;
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const double v = fma(s_w[i], q, T[i]);
        if (fl) Fg[i] = v; else Sg[i] = v;
    }
    if (zero_start) {                               // the slot is an output only: its padding rows too,
        for (int i = ROWS; i < 64; ++i) { if (fl) Fg[i] = 0.0; else Sg[i] = 0.0; }
        if (fl) for (int i = 0; i < 64; ++i) Sg[i] = 0.0;          // and S's own column 63 (lane 63 carries F~)
    }
}

// ------------------------------------------------------------------------------------
// k_factor7: the fused sweep with a 2 x 32 lane tiling (half the LDS operand traffic).
//
// k_factor3 gives every lane one column of T and all ROWS rows, so each FMA needs a wave-uniform row
// operand that reaches the lanes as an LDS broadcast: 1 KiB through the LDS array for 16 useful
// bytes, 60 ds_read_b128 per row.  With two waves on each of the four SIMDs that is 81 % of the
// CU's LDS-array cycles (SQ_LDS_IDX_ACTIVE) -- the resource the kernel actually saturates, ahead of
// vector issue (69 %).  Here lane = (g, c) = (lane >> 5, lane & 31) holds the TWO columns 2c, 2c+1
// (the cos and sin columns of complex term c: one phasor per lane instead of the same phasor in two
// lanes) of HALF the rows (pairs of rows 4k + 2g, 4k + 2g + 1): a ds_read_b128 then delivers a
// different operand pair to each half-wave and feeds 8 FMAs, 30 reads per row instead of 60.  The
// row vectors stay one column per lane (lane (g, c) owns column 2c + g, the generator is
// k_factor3's); two v_permlane32_swap exchanges per row connect the layouts (7 vector
// instructions: the mat-vec's partial sums, then the multipliers q).  Same arguments,
// state layout in memory and results as k_factor3; needs Jr = 0 (every gadfly kernel with
// Q > 1/2) and Jc <= 31 (block 31 carries the pad column 62 and the forward solve in column 63).
// ------------------------------------------------------------------------------------
// The row vectors (u~, v~, r, q) live one column per lane as in k_factor3 -- lane (g, c) owns column
// 2c + g -- while the state is tiled two columns x half the rows per lane.  Two exchanges between
// the half-waves connect the two layouts, each a pair of v_permlane32_swap (which swaps the upper
// half of its first operand with the lower half of its second):
//   own_column_sum(a, b): a, b = this half-wave's partial sums of columns 2c, 2c + 1; returns the
//     full sum of the lane's OWN column (lower half: a_L + a_U, upper half: b_L + b_U);
//   both_halves(x, lo, up): lo = x of lane c, up = x of lane c + 32, in both lanes.
__device__ __forceinline__ double own_column_sum(const double a, const double b) {
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ void both_halves(const double x, double &lo, double &up) {
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    lo = __hiloint2double(h[0], l[0]);
    up = __hiloint2double(h[1], l[1]);
}

// One LDS-DMA instruction (global_load_lds_dwordx4): lane l copies 16 bytes from ITS OWN global address to
// LDS byte `lds_byte` + 16 l -- no VGPR destination, so a deep prefetch costs LDS instead of registers.
// hipcc does not count it: completion is waited for with vm_wait (vector-memory operations retire in
// issue order).  `after` is a value that must exist before the copy may issue (keeps the copy behind
// the reads of the ring slot it overwrites).
__device__ __forceinline__ void glds16(const void *gsrc, const unsigned lds_byte, const int after) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte), "v"(after) : "memory");
}
template <int CNT>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CNT) : "memory"); }

constexpr int S7_AHEAD = 1;         // batches (one row pair = 8 FMAs + 2 reads) of LDS look-ahead

template <int ROWS>
__device__ __forceinline__ void sweep7_preload(double2 (&ub)[S7_AHEAD + 1], double2 (&wb)[S7_AHEAD + 1],
                                               const double2 *pu, const double2 *pw) {
#pragma unroll
    for (int k = 0; k < S7_AHEAD && k < ROWS / 4; ++k) { ub[k] = pu[2 * k]; wb[k] = pw[2 * k]; }
}

// T[2k + s][e] is row 4k + 2g + s, column 2c + e.  pu / pw point at this half-wave's first row pair.
//   RESET = false:  T += w q^T ;  acc += u^T T         (update + mat-vec, acc[e][s])
//   RESET = true :  T  = (T + w q^T) * (e_row el_col)  (fold pending, decay; pu = the row decays)
template <int ROWS, bool RESET>
__device__ __forceinline__ void sweep7_run(double (&T)[ROWS / 2][2], double2 (&ub)[S7_AHEAD + 1],
                                           double2 (&wb)[S7_AHEAD + 1], const double2 *pu,
                                           const double2 *pw, const double q0, const double q1,
                                           const double el0, const double el1, double &o0, double &o1) {
    constexpr int NK = ROWS / 4, AHEAD = S7_AHEAD;
    static_assert(ROWS % 4 == 0, "ROWS must be a multiple of 4");
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + AHEAD < NK) {
            ub[(k + AHEAD) % (AHEAD + 1)] = pu[2 * (k + AHEAD)];
            wb[(k + AHEAD) % (AHEAD + 1)] = pw[2 * (k + AHEAD)];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double2 u = ub[k % (AHEAD + 1)], w = wb[k % (AHEAD + 1)];
        if constexpr (RESET) {
            T[2 * k][0] = fma(w.x, q0, T[2 * k][0]) * (u.x * el0);
            T[2 * k][1] = fma(w.x, q1, T[2 * k][1]) * (u.x * el1);
            T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]) * (u.y * el0);
            T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]) * (u.y * el1);
        } else {
            T[2 * k][0] = fma(w.x, q0, T[2 * k][0]);
            T[2 * k][1] = fma(w.x, q1, T[2 * k][1]);
            T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]);
            T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]);
            a00 = fma(u.x, T[2 * k][0], a00);
            a01 = fma(u.x, T[2 * k][1], a01);
            a10 = fma(u.y, T[2 * k + 1][0], a10);
            a11 = fma(u.y, T[2 * k + 1][1], a11);
            asm volatile("" : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    o0 = a00 + a10;
    o1 = a01 + a11;
}

// ROWSTORE = false: no row is stored (r_out, Ut_out, Wt_out, de_out all null: the streamed log-likelihood
// and the plain final pass) -- their tests, pointer arithmetic and exec-mask switches leave the row loop
template <int ROWS, bool ROWSTORE>
__global__ void __launch_bounds__(64, 2)
k_factor7(const int64_t N, const int64_t n_first, const int64_t chunk_len, const int nch,
          const int ch0, const int nsel, const int Jr, const int Jc, const int block_sub, const double gap,
          const double *__restrict__ ar_, const double *__restrict__ cr_,
          const double *__restrict__ ac_, const double *__restrict__ bc_,
          const double *__restrict__ cc_, const double *__restrict__ dc_,
          const double *__restrict__ diag_add_, const double *__restrict__ cmax_,
          const double *__restrict__ t_, const int64_t t_bs,
          const double *__restrict__ diag_, const int64_t diag_bs,
          const double *__restrict__ y_, const int64_t y_bs,
          double *__restrict__ d_, double *__restrict__ z_, double *__restrict__ r_out,
          double *__restrict__ Ut_out, double *__restrict__ Wt_out, double *__restrict__ de_out,
          double *__restrict__ S_state, double *__restrict__ F_state,
          int32_t *__restrict__ info) {
    const int lane = threadIdx.x;
    const int g = lane >> 5, c = lane & 31;         // row group, column block (columns 2c, 2c + 1)
    const int sel = blockIdx.x;                     // chunks ch0 .. ch0 + nsel - 1 of every problem
    const int pr = sel / nsel, ch = ch0 + (sel - pr * nsel);
    const int b = pr * nch + ch;                    // state slot = problem * nch + chunk
    if (info[b] != 0) return;
    const int64_t c0 = (int64_t)ch * chunk_len;
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const int64_t g0 = n_first + c0;
    const size_t pb = (size_t)pr * N + c0;
    const double *__restrict__ tg = t_ + (size_t)pr * t_bs + g0;
    const double *__restrict__ yg = y_ + (size_t)pr * y_bs + g0;
    const bool has_g = diag_ != nullptr;
    const double *__restrict__ gg = has_g ? diag_ + (size_t)pr * diag_bs + g0 : yg;
    double *__restrict__ dg = d_ + pb;
    double *__restrict__ zg = z_ + pb;
    const int own = 2 * c + g;                      // the column whose row-vector entries this lane carries
    // chunk-mode row stores: loop-invariant per-lane pointers, indexed with opaque_uniform(row)
    double *__restrict__ rg = (ROWSTORE && r_out) ? r_out + pb * 64 + own : nullptr;
    double *__restrict__ ug = (ROWSTORE && Ut_out) ? Ut_out + pb * 64 + own : nullptr;
    double *__restrict__ wg = (ROWSTORE && Wt_out) ? Wt_out + pb * 64 + own : nullptr;
    // a nominal pass (no w~ rows: nothing but the transition sweep reads what it stores) writes the first ROWS
    // columns of its 64-double rows only -- cfg3 (W = 40) moved 34 GB per evaluation, a third of it padding; the
    // stored factor's rows keep their zero padding (the solves read whole rows)
    const bool st_ok = own < ROWS || Wt_out != nullptr;
    double *__restrict__ eg = (ROWSTORE && de_out) ? de_out + pb : nullptr;
    double *__restrict__ Sg = S_state + (size_t)b * (64 * 64);     // [column][row]
    double *__restrict__ Fg = F_state + (size_t)b * 64;
    const double diag_add = diag_add_[pr];

    RowGen G;                                       // this lane generates its own column, as in k_factor3
    G.init(own, pr, Jr, Jc, block_sub, gap, ar_, cr_, ac_, bc_, cc_, dc_, cmax_, tg, g0);
    const double cj = G.cj;

    __shared__ __attribute__((aligned(16))) double s_w[64];     // r_{n-1}  (pending update, row form)
    __shared__ __attribute__((aligned(16))) double s_u[64];     // u~_n
    __shared__ __attribute__((aligned(16))) double s_e[64];     // block decay at reset rows
    const double2 *pw = (const double2 *)s_w + g, *pu = (const double2 *)s_u + g;
    const double2 *pe = (const double2 *)s_e + g;
    const bool f31 = c == 31;                       // block 31: pad column 62, forward solve in 63
    const bool fl = lane == 63;                     // own column 63: the forward solve
    const double not63 = fl ? 0.0 : 1.0, is31 = f31 ? 1.0 : 0.0;

    // this lane's two columns in memory, from its first row on (column 63 lives in F_state)
    double *__restrict__ col0 = Sg + (size_t)(2 * c) * 64 + 2 * g;
    double *__restrict__ col1 = (f31 ? Fg : Sg + (size_t)(2 * c + 1) * 64) + 2 * g;
    double T[ROWS / 2][2];
    const bool zero_start = (block_sub >> 30) & 1;  // nominal pass: the state slots are outputs only
#pragma unroll
    for (int m = 0; m < ROWS / 2; ++m) {
        T[m][0] = zero_start ? 0.0 : col0[4 * (m >> 1) + (m & 1)];
        T[m][1] = zero_start ? 0.0 : col1[4 * (m >> 1) + (m & 1)];
    }
    double q0 = 0.0, q1 = 0.0;
    int32_t fail = 0;

    double t_n1 = tg[1], t_n2 = tg[2];
    double y_n = yg[0], y_n1 = yg[1];
    double g_n = gg[0], g_n1 = gg[1];
    double ut, vt, de;
    bool rst;
    G.next(tg[0], g0, ut, vt, rst, de);

    double2 ub[S7_AHEAD + 1], wb[S7_AHEAD + 1];
    s_w[own] = 0.0;
    s_u[own] = ut;
    wave_lds_fence();
    sweep7_preload<ROWS>(ub, wb, pu, pw);

    for (int64_t n = 0; n < rows; ++n) {
        const double a_n = (has_g ? g_n : 0.0) + diag_add, yy = y_n;
        const double ut_c = ut, vt_c = vt;
        if constexpr (ROWSTORE) { if (eg && lane == 0) eg[n] = rst ? de : -1.0; }
        if (rst) {                          // wave-uniform: fold the pending update, then decay
            const double el = fm_exp(-cj * de);     // pad columns: cj = 0 -> 1
            s_e[own] = el;
            wave_lds_fence();
            double d0, d1, el0, el1;
            both_halves(el, el0, el1);              // decays of this lane's two state columns
            sweep7_preload<ROWS>(ub, wb, pe, pw);
            sweep7_run<ROWS, true>(T, ub, wb, pe, pw, q0, q1, el0, el1, d0, d1);
            sweep7_preload<ROWS>(ub, wb, pu, pw);
            q0 = 0.0;
            q1 = 0.0;
        }
        __builtin_amdgcn_s_setprio(0);
        double acc0, acc1;
        sweep7_run<ROWS, false>(T, ub, wb, pu, pw, q0, q1, 0.0, 0.0, acc0, acc1);
        __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);
        const double tmp = own_column_sum(acc0, acc1);
        const double r = (vt_c - tmp) * not63;
        double r0, r1;
        both_halves(r, r0, r1);                     // r of this lane's two state columns: off the d-chain
        G.next(t_n1, g0 + n + 1, ut, vt, rst, de);
        t_n1 = t_n2; y_n = y_n1; g_n = g_n1;
        t_n2 = tg[n + 3];
        y_n1 = yg[n + 2];
        g_n1 = gg[n + 2];
        wave_lds_fence();
        s_w[own] = r;
        s_u[own] = ut;
        wave_lds_fence();
        sweep7_preload<ROWS>(ub, wb, pu, pw);
        const double s1 = wave_sum(ut_c * tmp);     // u~ = 0 in the pad / forward-solve columns
        const double s2 = read_lane(tmp, 63);       // u~ . F~ (column 63)
        const double dn = a_n - s1;
        const double zn = yy - s2;
        if (!(dn > 0.0)) {
            const int64_t gf = g0 + n + 1;
            fail = (int32_t)(gf > 0x7fffffff ? 0x7fffffff : gf);
            break;
        }
        const double inv = fast_rcp(dn);
        q0 = r0 * inv;                              // multipliers of this lane's two state columns;
        q1 = fma(zn, is31, r1) * inv;               // column 63: z / d (r is 0 there)
        if constexpr (ROWSTORE) {
            const size_t ro = opaque_uniform((size_t)n * 64);
            if (st_ok) {
                if (rg) rg[ro] = r;                 // r~ rows for k_phi (chunk mode)
                if (ug) ug[ro] = ut_c;
                if (wg) wg[ro] = r * inv;
            }
        }
        if (lane == 0) { dg[n] = dn; zg[n] = zn; }
    }
    if (fail) {
        if (lane == 0) info[b] = fail;
        return;
    }
    wave_lds_fence();
#pragma unroll
    for (int m = 0; m < ROWS / 2; ++m) {
        const double w = s_w[2 * g + 4 * (m >> 1) + (m & 1)];
        col0[4 * (m >> 1) + (m & 1)] = fma(w, q0, T[m][0]);
        col1[4 * (m >> 1) + (m & 1)] = fma(w, q1, T[m][1]);
    }
    if (zero_start) {                               // the slot is an output only: its padding rows too
        for (int m = ROWS / 2; m < 32; ++m) {
            col0[4 * (m >> 1) + (m & 1)] = 0.0;
            col1[4 * (m >> 1) + (m & 1)] = 0.0;
        }
        if (f31)                                    // S's own column 63 (these lanes' second column is F~)
            for (int m = 0; m < 32; ++m) Sg[63 * 64 + 2 * g + 4 * (m >> 1) + (m & 1)] = 0.0;
    }
}

#define GF_MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// Closed-loop transition sweeps of the time-parallel evaluation, W <= 63 (k_phi7: k_factor7's lane tiling,
// complex terms only; k_phi: one column per lane, any terms), DESIGN.md 4.3:
//     Phi <- (I - w~ u~^T) Phi  (E Phi at reset rows),  h_n = Phi^T u~_n,
//     G = sum_n h_n h_n^T / dbar_n,   m = sum_n h_n zbar_n / dbar_n
// on the rows the nominal pass stored (u~, r-bar, dbar, zbar, reset spans): no row generator here, and the
// Gram sums are accumulated in the same kernel -- the rows h never leave the chip.
//   * rows arrive through an LDS ring filled by LDS-DMA five rows ahead of their use (PhiRing); a register
//     queue two rows deep left every row waiting on HBM once the row arrays outgrew the caches (cfg3: 8.5 GB);
//   * sixteen rows of h are collected in an LDS tile and folded into the G accumulators as a rank-16
//     update on v_mfma_f64_16x16x4 (PhiGram: NT (NT + 1) / 2 symmetric tiles of 16 x 16, NT = ceil(W / 16)).
// ------------------------------------------------------------------------------------
struct PhiRing {
    static constexpr int PD = 8;                    // slots; rows are fetched PD - 1 ahead, used PD - 3 later
    static constexpr int USL = 72, RSL = 64;        // doubles per slot: u~ row + 3 scalar pairs / r-bar row
    static constexpr int OD = 64, OE = 66, OZ = 68; // dbar / reset span / zbar pair of the row in its u~ slot
    double u[PD][USL];
    double r[PD + 1][RSL];                          // slot PD: zeros ("no pending update" of the first row)
};

// per-lane copy cursors: lanes 0 .. 31 the row's doubles 2 lane, 2 lane + 1; lanes 32 / 33 / 34 of the u~
// copy the aligned pairs that hold dbar[row] / de[row] / zbar[row]; the other lanes stay out (EXEC)
struct PhiCursor {
    uintptr_t cu, cr, mask;
    unsigned long long ustep;
    bool in_u, in_r;
    unsigned lds_u, lds_r;
    // `rows`: the sweep's state rows (ROWS >= W): the copies fetch the row's first `rows` doubles only -- with the
    // nominal pass storing no more than that (see its row stores) the padding of the 64-double rows never travels
    template <int rows = 64>
    __device__ __forceinline__ void init(PhiRing &R, const int lane, const size_t pb, const double *ut_,
                                         const double *rbar_, const double *dbar_, const double *de_,
                                         const double *zbar_) {
        cu = reinterpret_cast<uintptr_t>(ut_ + pb * 64 + 2 * (lane & 31));
        cr = reinterpret_cast<uintptr_t>(rbar_ + pb * 64 + 2 * (lane & 31));
        mask = ~(uintptr_t)0;
        ustep = 64 * 8;
        if (lane >= 32) {
            const double *sp = (lane == 32) ? dbar_ : (lane == 33) ? de_ : zbar_;
            cu = reinterpret_cast<uintptr_t>(sp + pb);
            mask = ~(uintptr_t)15;
            ustep = 8;
        }
        in_u = lane < (rows + 1) / 2 || (lane >= 32 && lane < 35);
        in_r = lane < (rows + 1) / 2;
        lds_u = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(&R.u[0][0]));
        lds_r = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(&R.r[0][0]));
    }
    // row m (the cursors stand at it) -> slot m mod PD; two vector-memory operations
    __device__ __forceinline__ void issue(const int64_t m, const int after) {
        const unsigned sl = (unsigned)(m & (PhiRing::PD - 1));
        if (in_u) glds16(reinterpret_cast<const void *>(cu & mask), lds_u + sl * (PhiRing::USL * 8), after);
        if (in_r) glds16(reinterpret_cast<const void *>(cr), lds_r + sl * (PhiRing::RSL * 8), after);
        cu += ustep;
        cr += 64 * 8;
    }
};

template <int NT>
struct PhiGram {
    static constexpr int LD = 80;                   // tile row stride: rows k, k + 1 land 32 banks apart
    static constexpr int NACC = NT * (NT + 1) / 2;
    double hs[16][LD];                              // h of sixteen rows
    double sv[16];                                  // 1 / dbar of those rows (0: row not there)
};

// fold the tile into the accumulators: acc(it, jt) += sum_k (h_k[16 it + i] / d_k) h_k[16 jt + j]
template <int NT>
__device__ __forceinline__ void phi_gram_flush(PhiGram<NT> &Gm, d4 (&acc)[PhiGram<NT>::NACC], const int lane) {
    const int li = lane & 15, lk = lane >> 4;
    wave_lds_fence();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int k = 4 * ks + lk;
        const double sc = Gm.sv[k];
        double hv[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) hv[it] = Gm.hs[k][16 * it + li];
        int t = 0;
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int jt = it; jt < NT; ++jt, ++t) acc[t] = GF_MFMA64(hv[it] * sc, hv[jt], acc[t]);
    }
    wave_lds_fence();
    if (lane < 16) Gm.sv[lane] = 0.0;               // (a partial last block folds zeros for the missing rows)
}

// G ([column][row], 64 x 64, zero outside the tiles) and m of one chunk
template <int NT>
__device__ __forceinline__ void phi_gram_store(const d4 (&acc)[PhiGram<NT>::NACC], const double macc,
                                               double *__restrict__ Gg, double *__restrict__ mg, const int lane,
                                               const int own) {
    const int li = lane & 15, lk = lane >> 4;
    for (int e = lane; e < 4096; e += 64) {
        const int col = e >> 6, row = e & 63;
        if (col >= 16 * NT || row >= 16 * NT) Gg[e] = 0.0;
    }
    int t = 0;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int jt = it; jt < NT; ++jt, ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * it + lk + 4 * r, col = 16 * jt + li;
                const double v = acc[t][r];
                if (jt > it || col >= row) {        // upper triangle, mirrored: exactly symmetric
                    Gg[(size_t)col * 64 + row] = v;
                    Gg[(size_t)row * 64 + col] = v;
                }
            }
    mg[own] = macc;
}

template <int ROWS>
__global__ void __launch_bounds__(64, 2)
k_phi7(const int64_t N, const int64_t chunk_len, const int nch, const int ch0, const int nsel, const int W,
       const double *__restrict__ c_, const double *__restrict__ de_, const double *__restrict__ dbar_,
       const double *__restrict__ zbar_, const double *__restrict__ rbar_, const double *__restrict__ ut_,
       double *__restrict__ Phi_out, double *__restrict__ G_out, double *__restrict__ m_out) {
    constexpr int NT = (ROWS + 15) / 16, PD = PhiRing::PD;
    const int lane = threadIdx.x;
    const int g = lane >> 5, c = lane & 31, own = 2 * c + g;    // k_factor7's tiling: state columns 2c, 2c + 1
    const int sel = blockIdx.x;                                 // of half the rows; row vectors: column `own`
    const int pr = sel / nsel, ch = ch0 + (sel - pr * nsel);    // chunks ch0 .. ch0 + nsel - 1 of every problem
    const int b = pr * nch + ch;                                // state slot
    const int64_t c0 = (int64_t)ch * chunk_len;
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const size_t pb = (size_t)pr * N + c0;
    double *__restrict__ col0 = Phi_out + (size_t)b * (64 * 64) + (size_t)(2 * c) * 64 + 2 * g;
    double *__restrict__ col1 = col0 + 64;
    const double cj = (own < W) ? c_[(size_t)pr * W + own] : 0.0;

    __shared__ __attribute__((aligned(16))) PhiRing R;
    __shared__ __attribute__((aligned(16))) PhiGram<NT> Gm;
    __shared__ __attribute__((aligned(16))) double s_e[64];
    const double2 *pe = (const double2 *)s_e + g;
    double T[ROWS / 2][2];
#pragma unroll
    for (int m = 0; m < ROWS / 2; ++m) {                        // Phi = I
        const int row = 4 * (m >> 1) + 2 * g + (m & 1);
        T[m][0] = (row == 2 * c) ? 1.0 : 0.0;
        T[m][1] = (row == 2 * c + 1) ? 1.0 : 0.0;
    }
    double q0 = 0.0, q1 = 0.0, macc = 0.0;
    d4 acc[PhiGram<NT>::NACC];
#pragma unroll
    for (int t = 0; t < PhiGram<NT>::NACC; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    R.r[PD][lane] = 0.0;
    for (int e = lane; e < 16 * PhiGram<NT>::LD; e += 64) (&Gm.hs[0][0])[e] = 0.0;
    if (lane < 16) Gm.sv[lane] = 0.0;
    PhiCursor C;
    C.template init<ROWS>(R, lane, pb, ut_, rbar_, dbar_, de_, zbar_);
#pragma unroll
    for (int m = 0; m < PD - 1; ++m) C.issue(m, 0);
    vm_wait<2 * (PD - 2)>();                        // row 0 has landed
    wave_lds_fence();
    double2 ub[S7_AHEAD + 1], wb[S7_AHEAD + 1];
    const double2 *pu = (const double2 *)&R.u[0][0] + g;
    const double2 *pw = (const double2 *)&R.r[PD][0] + g;
    sweep7_preload<ROWS>(ub, wb, pu, pw);

    for (int64_t n = 0; n < rows; ++n) {
        // rows up to n + 1 have landed: behind row n + 1's copies are those of the PD - 3 rows after it
        vm_wait<2 * (PD - 3)>();
        const int sl = (int)(n & (PD - 1)), par = (int)((pb + n) & 1);
        const double dcur = R.u[sl][PhiRing::OD + par], zcur = R.u[sl][PhiRing::OZ + par];
        const double dev = R.u[sl][PhiRing::OE + par];
        const double de = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dev)),
                                           __builtin_amdgcn_readfirstlane(__double2loint(dev)));
        if (de >= 0.0) {                    // reset row: Phi <- E (Phi + pending): row scaling only
            s_e[own] = fm_exp_local(-cj * de);
            wave_lds_fence();
            double d0, d1;
            sweep7_preload<ROWS>(ub, wb, pe, pw);
            sweep7_run<ROWS, true>(T, ub, wb, pe, pw, q0, q1, 1.0, 1.0, d0, d1);
            sweep7_preload<ROWS>(ub, wb, pu, pw);
            q0 = 0.0;
            q1 = 0.0;
        }
        double acc0, acc1;
        sweep7_run<ROWS, false>(T, ub, wb, pu, pw, q0, q1, 0.0, 0.0, acc0, acc1);
        const double h = own_column_sum(acc0, acc1);
        const double rinv = (dcur > 0.0) ? fast_rcp(dcur) : 0.0;
        Gm.hs[n & 15][own] = h;
        if (lane == 0) Gm.sv[n & 15] = rinv;
        macc = fma(h, zcur * rinv, macc);
        // next row's operands: u~ of row n + 1 (landed), the pending r-bar of row n in its ring slot
        pu = (const double2 *)&R.u[(int)((n + 1) & (PD - 1))][0] + g;
        pw = (const double2 *)&R.r[sl][0] + g;
        sweep7_preload<ROWS>(ub, wb, pu, pw);
        both_halves(-h * rinv, q0, q1);     // pending: Phi_i -= (r_i / d) h_j
        // row n + PD - 1 into the slot of row n - 1, whose rows this iteration's sweeps were the last to read
        C.issue(n + PD - 1, __double2loint(h));
        if ((n & 15) == 15) phi_gram_flush<NT>(Gm, acc, lane);
    }
    vm_wait<0>();                           // no copy may outlive the wave's LDS allocation
    if (rows & 15) phi_gram_flush<NT>(Gm, acc, lane);
    wave_lds_fence();
    const double *s_w = &R.r[(int)((rows - 1) & (PD - 1))][0];
#pragma unroll
    for (int m = 0; m < ROWS / 2; ++m) {
        const int ro = 4 * (m >> 1) + (m & 1);
        const double w = s_w[2 * g + ro];
        col0[ro] = fma(w, q0, T[m][0]);
        col1[ro] = fma(w, q1, T[m][1]);
    }
#pragma unroll
    for (int k2 = ROWS / 4; k2 < 16; ++k2) {                    // rows past ROWS: zero
        col0[4 * k2] = 0.0; col0[4 * k2 + 1] = 0.0;
        col1[4 * k2] = 0.0; col1[4 * k2 + 1] = 0.0;
    }
    phi_gram_store<NT>(acc, macc, G_out + (size_t)b * (64 * 64), m_out + (size_t)b * 64, lane, own);
}

// ------------------------------------------------------------------------------------
// k_factorw: the fused sweep (build + factor + forward solve, nothing materialised) for WIDE kernels,
// 64 <= W <= 176, complex terms only (Jr = 0): cfg4's W = 80, the 86-term solar kernel's W = 172.
//
// One workgroup of NW sweep waves + ONE generator wave per (problem, chunk).
// Sweep waves: the state columns are split over the waves and, inside a wave, tiled as in k_factor7 but
// four ways: lane = (g, c) = (lane >> 4, lane & 15) of wave w holds the column pair cb = 16 w + c
// (columns 2cb, 2cb + 1: the cos and sin columns of complex term cb) of the row pairs 8k + 2g,
// 8k + 2g + 1: a ds_read_b128 delivers a different operand pair to each 16-lane row group of the wave and
// feeds 8 FMAs.  EVERY wave holds all rows of its 32 columns, so the mat-vec needs no cross-wave
// reduction: the four row groups of a wave are folded with one v_permlane32_swap pair (reduce-scatter:
// lower half <- column 2cb, upper half <- column 2cb + 1) and one v_permlane16_swap pair -- 6 vector
// instructions.  Lane (g, c) then owns column own = 2cb + (lane >> 5) of the row vectors.
// Generator wave: lane l carries the phasors of complex terms l and l + 64 (k_factor3's RowGen) and
// writes row n + 1's u~, v~ (and, for a reset row, its span and the column decays) to LDS while the sweep
// waves work on row n.  A sweep wave therefore holds nothing but its state, the operand ring and a few
// scalars -- which is what lets W = 176 (T = 176 VGPRs per lane) fit the register file: with the
// generator inside the sweep waves (as in k_factor7) the state of W > 112 spilled.
// What crosses waves, through LDS and ONE workgroup barrier per row: the new r (every wave's next sweep
// needs all of it as row operands), the next row's u~ / v~, and the per-wave partial of u~ . tmp (the
// pivot) -- double-buffered by row parity.  The forward solve rides in the last padded column
// (FCOL = 32 NW - 1) exactly as in k_factor3 / k_factor7.
// Replaces k_build2 + k_factor2w (materialised u~, v~ rows: 2 x 8 ld bytes per row and evaluation
// through HBM, and a sweep that waited on them) on the log-likelihood path.
// ------------------------------------------------------------------------------------
struct FactorWArgs {            // the scalars; the arrays are separate __restrict__ kernel parameters
    int64_t N, n_first, chunk_len, t_bs, diag_bs, y_bs;
    int nch, ch0, nsel, Jc, block_sub;
    double gap;
};

struct FactorWPtrs {
    const double *ac, *bc, *cc, *dc, *diag_add, *cmax, *t, *diag, *y;
    double *d, *z, *r_out, *Ut_out, *Wt_out, *de_out, *S_state;
    int32_t *info;
};

// folds the four 16-lane row groups of a wave: a, b = this lane's partial sums of columns 2cb, 2cb + 1;
// returns the full sum of the lane's OWN column (lanes 0..31: column 2cb, lanes 32..63: 2cb + 1)
__device__ __forceinline__ double own_column_sum4(const double a, const double b) {
    const double h = own_column_sum(a, b);          // lanes l, l ^ 32 folded (reduce-scatter over the halves)
    const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(h), __double2loint(h), false, false);
    const auto u = __builtin_amdgcn_permlane16_swap(__double2hiint(h), __double2hiint(h), false, false);
    return __hiloint2double(u[0], l[0]) + __hiloint2double(u[1], l[1]);         // l, l ^ 16
}

#ifndef GF_WIDE_AHEAD
#define GF_WIDE_AHEAD 1
#endif
// T[2k + s][e] is row 2G k + 2g + s, column 2cb + e (G = row groups per wave: 4 in k_factorw, 8 in
// k_phiw).  pu / pw point at this row group's first row pair.
//   RESET = false:  T += w q^T ;  acc += u^T T         (update + mat-vec)
//   RESET = true :  T  = (T + w q^T) * (e_row el_col)  (fold pending, decay; pu = the row decays)
template <int TR, bool RESET, int G = 4>
__device__ __forceinline__ void sweepw_run(double (&T)[TR][2], const double2 *pu, const double2 *pw,
                                           const double q0, const double q1, const double el0,
                                           const double el1, double &o0, double &o1) {
    constexpr int NK = TR / 2;
    // operand ring: AH row pairs of LDS look-ahead (one batch = 8 FMAs = ~35 cycles of issue covers
    // nothing of the LDS latency when the partner wave is not issuing; the state of a wide kernel leaves
    // room for a deeper ring only while TR is small)
    constexpr int AH = (GF_WIDE_AHEAD < NK) ? GF_WIDE_AHEAD : NK;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    double2 ub[AH + 1], wb[AH + 1];
#pragma unroll
    for (int k = 0; k < AH; ++k) { ub[k] = pu[G * k]; wb[k] = pw[G * k]; }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + AH < NK) { ub[(k + AH) % (AH + 1)] = pu[G * (k + AH)]; wb[(k + AH) % (AH + 1)] = pw[G * (k + AH)]; }
        __builtin_amdgcn_sched_barrier(0);
        const double2 u = ub[k % (AH + 1)], w = wb[k % (AH + 1)];
        if constexpr (RESET) {
            T[2 * k][0] = fma(w.x, q0, T[2 * k][0]) * (u.x * el0);
            T[2 * k][1] = fma(w.x, q1, T[2 * k][1]) * (u.x * el1);
            T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]) * (u.y * el0);
            T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]) * (u.y * el1);
        } else {
            T[2 * k][0] = fma(w.x, q0, T[2 * k][0]);
            T[2 * k][1] = fma(w.x, q1, T[2 * k][1]);
            T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]);
            T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]);
            a00 = fma(u.x, T[2 * k][0], a00);
            a01 = fma(u.x, T[2 * k][1], a01);
            a10 = fma(u.y, T[2 * k + 1][0], a10);
            a11 = fma(u.y, T[2 * k + 1][1], a11);
            asm volatile("" : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    o0 = a00 + a10;
    o1 = a01 + a11;
}

// LDS of one k_factorw workgroup.  The generator runs TWO rows ahead of the sweep (three buffers for
// what it writes), so that everything a sweep wave needs of row n + 1 except the new r is readable before
// row n's barrier; r and the pivot partials alternate between two buffers.
template <int NW>
struct WideShared {
    static constexpr int CP = 32 * NW;
    double w[2][CP];            // r_{n-1}: pending rank-1 update (row form)
    double u[3][CP];            // u~_n
    double v[3][CP];            // v~_n
    double e[3][CP];            // column decays of a reset row
    double f[3][4];             // per row: reset span de (>= 0, or -1 for a plain row), diagonal a_n, y_n
    double p[2][NW][2];         // per-wave partials: u~^T T u~ share, u~ . F~
};

// the update + mat-vec sweep with the w operands (the new r: the only operands behind the row's barrier)
// already in registers -- fetched right after the barrier, in the shadow of the pivot -- and the u~
// operands (readable long before) through a two-deep LDS ring: the FMAs never wait on the critical path
template <int TR>
__device__ __forceinline__ void sweepw_regs(double (&T)[TR][2], const double2 *pu,
                                            const double2 (&wb)[TR / 2], const double q0, const double q1,
                                            double &o0, double &o1) {
    constexpr int NK = TR / 2, AH = (NK > 2) ? 2 : NK;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    double2 ub[AH + 1];
#pragma unroll
    for (int k = 0; k < AH; ++k) ub[k] = pu[4 * k];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + AH < NK) ub[(k + AH) % (AH + 1)] = pu[4 * (k + AH)];
        __builtin_amdgcn_sched_barrier(0);
        const double2 u = ub[k % (AH + 1)], w = wb[k];
        T[2 * k][0] = fma(w.x, q0, T[2 * k][0]);
        T[2 * k][1] = fma(w.x, q1, T[2 * k][1]);
        T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]);
        T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]);
        a00 = fma(u.x, T[2 * k][0], a00);
        a01 = fma(u.x, T[2 * k][1], a01);
        a10 = fma(u.y, T[2 * k + 1][0], a10);
        a11 = fma(u.y, T[2 * k + 1][1], a11);
        asm volatile("" : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11));
        __builtin_amdgcn_sched_barrier(0);
    }
    o0 = a00 + a10;
    o1 = a01 + a11;
}

// rows between looks at the failure flag (a non-positive pivot is recorded without a branch; the rows
// swept after it, at most this many, are thrown away)
constexpr int WIDE_FAIL_CHECK = 64;

// (The arrays are passed as separate __restrict__ parameters, not inside FactorWArgs: only then can
// hipcc prove the wave-uniform t loads unclobbered and issue them as scalar loads.)
template <int TR, int NW>
__global__ void __launch_bounds__(64 * (NW + 1), 2)
k_factorw(const FactorWArgs A,
          const double *__restrict__ ac_, const double *__restrict__ bc_,
          const double *__restrict__ cc_, const double *__restrict__ dc_,
          const double *__restrict__ diag_add_, const double *__restrict__ cmax_,
          const double *__restrict__ t_, const double *__restrict__ diag_,
          const double *__restrict__ y_,
          double *__restrict__ d_, double *__restrict__ z_, double *__restrict__ r_out,
          double *__restrict__ Ut_out, double *__restrict__ Wt_out, double *__restrict__ de_out,
          double *__restrict__ S_state, int32_t *__restrict__ info) {
    static_assert(TR % 2 == 0, "rows per lane come in pairs");
    constexpr int RP = 4 * TR;                      // rows, padded (rows >= W: u~ = r = 0, T stays 0)
    constexpr int CP = 32 * NW;                     // columns, padded; the last one carries the forward solve
    static_assert(RP <= CP, "row operands are read from the column-indexed LDS vectors");
    constexpr int FCOL = CP - 1;                    // forward-solve column: last sweep wave, lanes 47 and 63
    constexpr int NK = TR / 2;
    constexpr bool PRE = TR <= 20;                  // the w operands of a whole row fit in registers (W <= 80)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sel = blockIdx.x;                     // chunks ch0 .. ch0 + nsel - 1 of every problem
    const int pr = sel / A.nsel, ch = A.ch0 + (sel - pr * A.nsel);
    const int b = pr * A.nch + ch;                  // state slot = problem * nch + chunk
    if (info[b] != 0) return;                       // (uniform over the workgroup)
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t rows = (A.N - c0 < A.chunk_len) ? (A.N - c0) : A.chunk_len;
    const int64_t g0 = A.n_first + c0;
    const size_t pb = (size_t)pr * A.N + c0;
    const double *__restrict__ tg = t_ + (size_t)pr * A.t_bs + g0;

    __shared__ __attribute__((aligned(16))) WideShared<NW> sh;

    if (wave == NW) {
        // ------- generator wave: rows u~, v~, reset spans and decays two rows ahead; the pivots -------
        const double *__restrict__ yg = y_ + (size_t)pr * A.y_bs + g0;
        const bool has_g = diag_ != nullptr;
        const double *__restrict__ gg = has_g ? diag_ + (size_t)pr * A.diag_bs + g0 : yg;
        const double diag_add = diag_add_[pr];
        double *__restrict__ dg = d_ + pb;
        double *__restrict__ zg = z_ + pb;
        constexpr int NP = (CP / 2 + 63) / 64;      // phasors per lane (1 up to W + 1 <= 128 columns, else 2)
        RowGen gen[NP];
        bool live[NP];
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const int ph = lane + 64 * h;           // complex term (column pair) of this slot
            live[h] = ph < CP / 2;
            gen[h].init(2 * ph, pr, 0, A.Jc, A.block_sub, A.gap, nullptr, nullptr, ac_, bc_, cc_, dc_,
                        cmax_, tg, g0);
        }
        double *__restrict__ ug = Ut_out ? Ut_out + pb * CP : nullptr;
        double *__restrict__ eg = de_out ? de_out + pb : nullptr;
        auto emit = [&](const double tn, const int64_t gi, const int buf, double *urow, const double an,
                        const double yn) {
            bool rst = false;
            double de = -1.0;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                gen[h].advance(tn, gi, rst, de);            // rst, de: the same for every phasor
                const double uc = fma(gen[h].k1, gen[h].cu, gen[h].k2 * gen[h].su);        // a cos + b sin
                const double us = fma(gen[h].k1, gen[h].su, -gen[h].k2 * gen[h].cu);       // a sin - b cos
                const double vc = gen[h].sel_c * gen[h].cu * gen[h].irho2;                 // pad pairs: 0
                const double vs = gen[h].sel_c * gen[h].su * gen[h].irho2;
                if (live[h]) {
                    const int j = 2 * (lane + 64 * h);
                    *reinterpret_cast<double2 *>(&sh.u[buf][j]) = double2{uc, us};
                    *reinterpret_cast<double2 *>(&sh.v[buf][j]) = double2{vc, vs};
                    if (rst) {
                        const double el = fm_exp(-gen[h].cj * de);      // pad pairs: cj = 0 -> 1
                        *reinterpret_cast<double2 *>(&sh.e[buf][j]) = double2{el, el};
                    }
                    if (urow) *reinterpret_cast<double2 *>(urow + j) = double2{uc, us};
                }
            }
            const double span = rst ? de : -1.0;
            // the row's scalars for the sweep waves (every lane writes the same values: no branch)
            *reinterpret_cast<double2 *>(&sh.f[buf][0]) = double2{span, an};
            sh.f[buf][2] = yn;
            return span;
        };
        // y, diag in the vector-load queue (opaque zero lane offset), three rows ahead: as scalar loads
        // they share the LDS counter, and every LDS wait of the row would wait for them as well
        const int vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);
        auto diag_of = [&](const double gv) { return (has_g ? gv : 0.0) + diag_add; };
        // (t, y, diag) of rows n + 2 and n + 3; rows n, n + 1 keep (a, y) for the pivots
        double t_2 = tg[2 + vz], t_3 = tg[3 + vz], y_2 = yg[2 + vz], y_3 = yg[3 + vz];
        double g_2 = gg[2 + vz], g_3 = gg[3 + vz];
        double a_0 = diag_of(gg[vz]), a_1 = diag_of(gg[1 + vz]), y_0 = yg[vz], y_1 = yg[1 + vz];
        double de_0 = emit(tg[vz], g0, 0, ug, a_0, y_0);
        double de_1 = emit(tg[1 + vz], g0 + 1, 1, (ug && rows > 1) ? ug + CP : nullptr, a_1, y_1);
        wg_lds_barrier();
        int b2 = 2;                                 // buffer of row n + 2
        int32_t fail = 0;
        double dacc = 0.0, zacc = 0.0;
        for (int64_t n = 0; n < rows; ++n) {
            const int nxt = (int)(n & 1) ^ 1;
            const double a_n = a_0, yy = y_0;
            if (eg && lane == 0) eg[n] = de_0;
            const double a_2 = diag_of(g_2);
            de_0 = de_1;
            de_1 = emit(t_2, g0 + n + 2, b2, (ug && n + 2 < rows) ? ug + (size_t)(n + 2) * CP : nullptr, a_2, y_2);
            b2 = (b2 == 2) ? 0 : b2 + 1;
            a_0 = a_1; y_0 = y_1; a_1 = a_2; y_1 = y_2;
            t_2 = t_3; y_2 = y_3; g_2 = g_3;
            t_3 = tg[n + 4 + vz];
            y_3 = yg[n + 4 + vz];
            g_3 = gg[n + 4 + vz];
            wg_lds_barrier();
            double s1 = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) s1 += sh.p[nxt][w2][0];
            const double dn = a_n - s1;
            const double zn = yy - sh.p[nxt][NW - 1][1];
            // d, z leave in coalesced blocks of 64 rows (lane j keeps row 64 m + j): a store per row would
            // sit in the same in-order queue as the y / diag prefetches and make their waits wait for HBM
            // write acknowledgements
            dacc = (lane == (int)(n & 63)) ? dn : dacc;
            zacc = (lane == (int)(n & 63)) ? zn : zacc;
            if ((n & 63) == 63) { dg[n - 63 + lane] = dacc; zg[n - 63 + lane] = zacc; }
            if (!(dn > 0.0) && fail == 0) {
                const int64_t gf = g0 + n + 1;
                fail = (int32_t)(gf > 0x7fffffff ? 0x7fffffff : gf);
            }
            if ((n & (WIDE_FAIL_CHECK - 1)) == WIDE_FAIL_CHECK - 1 && fail) break;
        }
        if (fail) {
            if (lane == 0) info[b] = fail;
        } else if (rows & 63) {                     // the last, partial block of d, z
            const int64_t base = rows - (rows & 63);
            if (lane < (int)(rows & 63)) { dg[base + lane] = dacc; zg[base + lane] = zacc; }
        }
        return;
    }

    // ---------------- sweep waves ------------------------------------------------------------------
    const int g = lane >> 4, c = lane & 15;
    const int cb = wave * 16 + c;                   // column pair (complex term) of this lane
    const int own = 2 * cb + (lane >> 5);           // the column whose row-vector entries this lane carries
    double *__restrict__ rg = r_out ? r_out + pb * CP + own : nullptr;
    double *__restrict__ wg = Wt_out ? Wt_out + pb * CP + own : nullptr;
    double *__restrict__ Sg = S_state + (size_t)b * ((size_t)CP * RP);      // [column][row]
    const double notF = (own == FCOL) ? 0.0 : 1.0;
    const double isF1 = (2 * cb + 1 == FCOL) ? 1.0 : 0.0;   // this lane's second state column is F~
    double *__restrict__ col0 = Sg + (size_t)(2 * cb) * RP + 2 * g;
    double *__restrict__ col1 = col0 + RP;
    double T[TR][2];
    const bool zero_start = (A.block_sub >> 30) & 1;        // nominal pass: the state slots are outputs only
#pragma unroll
    for (int m = 0; m < TR; ++m) {
        T[m][0] = zero_start ? 0.0 : col0[8 * (m >> 1) + (m & 1)];
        T[m][1] = zero_start ? 0.0 : col1[8 * (m >> 1) + (m & 1)];
    }
    double q0 = 0.0, q1 = 0.0;
    sh.w[0][own] = 0.0;
    wg_lds_barrier();
    // row 0's operands, own-column entries and reset span
    double2 wb[PRE ? NK : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < NK; ++k) wb[k] = double2{0.0, 0.0};
    }
    double vt_c = sh.v[0][own];
    double2 u01 = *reinterpret_cast<const double2 *>(&sh.u[0][2 * cb]);     // u~ of the two state columns
    double2 fa = *reinterpret_cast<const double2 *>(&sh.f[0][0]);           // (reset span, a_n)
    double yy = sh.f[0][2];
    double de = read_lane(fa.x, 0);
    int b0 = 0, b1 = 1;                             // buffers of rows n and n + 1
    int32_t fail = 0;

    // Two loops: the outer one runs once per scaling block (its first row is a reset row: fold the pending
    // update, decay), the inner one over the block's rows has NO conditional update of T -- with the reset
    // inside the row loop the register allocator kept T in two register sets and copied all of it twice
    // per row (84 v_mov_b64 next to 80 FMAs).
    int64_t n = 0;
    bool stop = false;
    while (n < rows && !stop) {
    if (de >= 0.0) {                        // workgroup-uniform reset row
        const double2 *pw = (const double2 *)sh.w[n & 1] + g;
        double d0, d1, el0, el1;
        both_halves(sh.e[b0][own], el0, el1);               // decays of this lane's two state columns
        if constexpr (PRE) {                                // (the pending multipliers are in registers)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const double2 e = ((const double2 *)sh.e[b0] + g)[4 * k], w = wb[k];
                T[2 * k][0] = fma(w.x, q0, T[2 * k][0]) * (e.x * el0);
                T[2 * k][1] = fma(w.x, q1, T[2 * k][1]) * (e.x * el1);
                T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]) * (e.y * el0);
                T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]) * (e.y * el1);
            }
            (void)pw; (void)d0; (void)d1;
        } else {
            sweepw_run<TR, true>(T, (const double2 *)sh.e[b0] + g, pw, q0, q1, el0, el1, d0, d1);
        }
        q0 = 0.0;
        q1 = 0.0;
    }
    do {
        const int cur = (int)(n & 1), nxt = cur ^ 1;
        const double2 *pw = (const double2 *)sh.w[cur] + g, *pu = (const double2 *)sh.u[b0] + g;
        __builtin_amdgcn_s_setprio(0);
        double acc0, acc1;
        if constexpr (PRE) sweepw_regs<TR>(T, pu, wb, q0, q1, acc0, acc1);
        else sweepw_run<TR, false>(T, pu, pw, q0, q1, 0.0, 0.0, acc0, acc1);
        // the serial part of the row wins the issue arbitration against the other workgroup's sweep
        __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);
        // what the generator wrote of row n + 1 (two rows ahead) is fetched here, in the shadow of the
        // reductions: this lane's own-column entries, the reset span, the diagonal and y
        const double vt_n = sh.v[b1][own];
        const double2 u01_n = *reinterpret_cast<const double2 *>(&sh.u[b1][2 * cb]);
        const double2 fa_n = *reinterpret_cast<const double2 *>(&sh.f[b1][0]);
        const double yy_n = sh.f[b1][2];
        // this lane's share of the quadratic form u~^T T u~ (its rows x its two columns): the pivot's
        // reduction runs next to -- not after -- the column sums
        const double p1 = wave_sum(fma(acc0, u01.x, acc1 * u01.y));
        const double tmp = own_column_sum4(acc0, acc1);
        const double r = (vt_c - tmp) * notF;
        double r0, r1;
        both_halves(r, r0, r1);                     // r of this lane's two state columns
        sh.w[nxt][own] = r;
        const double p2 = read_lane(tmp, 47);       // column FCOL of the last sweep wave: u~ . F~
        *reinterpret_cast<double2 *>(sh.p[nxt][wave]) = double2{p1, p2};    // (uniform, every lane: no branch)
        wg_lds_barrier();
        double s1 = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) s1 += sh.p[nxt][w2][0];
        const double s2 = sh.p[nxt][NW - 1][1];
        if constexpr (PRE) {                        // next row's w operands: the only reads behind the barrier
#pragma unroll
            for (int k = 0; k < NK; ++k) wb[k] = ((const double2 *)sh.w[nxt] + g)[4 * k];
        }
        const double dn = fa.y - s1;                // every wave forms the pivot itself
        const double zn = yy - s2;
        vt_c = vt_n; u01 = u01_n; fa = fa_n; yy = yy_n;
        de = read_lane(fa.x, 0);
        b0 = b1; b1 = (b1 == 2) ? 0 : b1 + 1;
        if (!(dn > 0.0)) fail = 1;                  // (the generator wave records the row)
        const double inv = fast_rcp(dn);
        q0 = r0 * inv;                              // multipliers of this lane's two state columns;
        q1 = fma(zn, isF1, r1) * inv;               // column FCOL: z / d (r is 0 there)
        const size_t ro = opaque_uniform((size_t)n * CP);
        if (rg) rg[ro] = r;                         // r~ rows (chunk mode)
        if (wg) wg[ro] = r * inv;
        stop = (n & (WIDE_FAIL_CHECK - 1)) == WIDE_FAIL_CHECK - 1 && fail;
        ++n;
    } while (n < rows && !(de >= 0.0) && !stop);
    }
    __builtin_amdgcn_s_setprio(0);
    if (fail) return;                               // (the generator wave records the row)
    const int fin = (int)(rows & 1);                // buffer written by the last row
#pragma unroll
    for (int m = 0; m < TR; ++m) {
        const int ro = 8 * (m >> 1) + (m & 1);
        const double w = sh.w[fin][2 * g + ro];
        col0[ro] = fma(w, q0, T[m][0]);
        col1[ro] = fma(w, q1, T[m][1]);
    }
}

// ------------------------------------------------------------------------------------
// k_phiw: the closed-loop transition sweep of the exact time-parallel evaluation (k_phi / k_phi7) for WIDE
// kernels:  Phi <- (I - w~ u~^T) Phi  (rows scaled by E at reset rows),  h_n = Phi^T u~_n, on the rows the
// nominal pass stored (u~ rows, r~ rows, pivots, reset spans).  Its columns are INDEPENDENT -- the
// multipliers -h_j / d of a column come from that column's own mat-vec -- so every wave sweeps its 16
// columns of Phi with no barrier and no generator: one single-wave workgroup per (problem, chunk, 16
// columns), lane (g, c) = (lane >> 3, lane & 7) holds the column pair 8 w + c of the row pairs
// 16k + 2g, 16k + 2g + 1 (8 row groups: 22 rows per lane at W = 172, so two waves per SIMD fit), row
// operands staged per wave in LDS from plain coalesced row loads two rows ahead.  The Gram sums
// G = sum h h^T / d and the W x W combines of the chunk maps are plain dense GEMMs / solves and run
// as library calls (rocBLAS / hipSOLVER through torch) on these outputs.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double own_column_sum8(const double a, const double b) {
    const double h = own_column_sum(a, b);          // lanes l, l ^ 32 (reduce-scatter over the halves)
    const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(h), __double2loint(h), false, false);
    const auto u = __builtin_amdgcn_permlane16_swap(__double2hiint(h), __double2hiint(h), false, false);
    double s = __hiloint2double(u[0], l[0]) + __hiloint2double(u[1], l[1]);     // l, l ^ 16
    s += dpp_get<0x128, 0xf>(s);                                                  // row_ror:8: l, l ^ 8
    return s;
}

// D ring slots (a power of two) of NI kilobytes per row vector; rows arrive D - 2 rows ahead of their use
// (the u~ row, the pivot and the reset span of a row travel in ONE slot: lanes 63 / 62 of the last piece
// fetch the 16-byte pairs that hold dbar[row] / de[row]).  The register queue this replaces was two rows
// deep: with row arrays beyond the caches (a shard of cfg4: 3 x 9.8 GB) a row cost 0.9 us, all of it
// memory latency at eight waves per CU.
template <int TR, int D, int NI>
__global__ void __launch_bounds__(64, (TR <= 12 && NI == 1) ? 4 : 2)
k_phiw(const int64_t N, const int64_t chunk_len, const int nch, const int ch0, const int nsel, const int W,
       const int CP,
       const double *__restrict__ c_, const double *__restrict__ de_, const double *__restrict__ dbar_,
       const double *__restrict__ rbar_, const double *__restrict__ ut_,
       double *__restrict__ h_out, double *__restrict__ Phi_out) {
    static_assert((D & (D - 1)) == 0 && D >= 4, "ring slots: a power of two");
    constexpr int RP = 8 * TR;                      // state rows, padded
    constexpr int LV = 192;                         // LDS row vectors (CP, RP <= 192)
    constexpr int NL = 3;                           // row entries per lane (lane, lane + 64, lane + 128)
    constexpr int SL = 128 * NI;                    // doubles per ring slot (CP <= SL - 4)
    constexpr int SD = SL - 2, SE = SL - 4;         // the row's pivot pair / reset-span pair
    const int lane = threadIdx.x;
    // waves (16-column groups) per chunk: groups made of pad columns only are not swept (the combine reads
    // 16 ceil(W / 16) columns of h and W columns of Phi; what lies beyond is left unwritten)
    const int nwv = (W + 15) / 16;
    const int sel = blockIdx.x / nwv, wave = blockIdx.x - sel * nwv;
    const int pr = sel / nsel, ch = ch0 + (sel - pr * nsel);       // chunks ch0 .. ch0 + nsel - 1 of every problem
    const int slot = pr * nch + ch;
    const int g = lane >> 3, c = lane & 7;
    const int cb = wave * 8 + c;                    // column pair of this lane
    const int own = 2 * cb + (lane >> 5);           // its own column (h, multipliers)
    const int64_t c0 = (int64_t)ch * chunk_len;
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const size_t pb = (size_t)pr * N + c0;
    double *__restrict__ hg = h_out + pb * CP + own;
    __shared__ __attribute__((aligned(16))) double ring_u[D][SL], ring_r[D + 1][SL], s_e[LV];
    const unsigned lds_u = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(&ring_u[0][0]));
    const unsigned lds_r = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(&ring_r[0][0]));
    const double2 *pe = (const double2 *)s_e + g;
    double ci[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = lane + 64 * k;
        ci[k] = (i < W) ? c_[(size_t)pr * W + i] : 0.0;
        s_e[i] = 1.0;                               // (LV = 192 = 3 * 64: every entry initialised)
    }
    for (int e = lane; e < SL; e += 64) ring_r[D][e] = 0.0;        // "no pending update" for the first row
    // Per-lane source of piece j of a row's u~ slot / r~ slot: address = cursor & mask, the cursor advancing by
    // `step` bytes per row.  Data lanes: the row's doubles 128 j + 2 lane (lanes past the row re-read its
    // start).  Lanes 63 / 62 of the last piece: the aligned 16-byte pair that holds dbar[row] / de[row] (step
    // 8, mask clears bit 3).  Rows are fetched D - 1 ahead, unconditionally: the caller's arrays are readable
    // 8 rows past the end.
    uintptr_t cu[NI], cr[NI], mu[NI];
    unsigned long long su_step[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int e = 128 * j + 2 * lane;
        const int ec = (e < CP) ? e : 0;
        cu[j] = reinterpret_cast<uintptr_t>(ut_ + pb * CP + ec);
        cr[j] = reinterpret_cast<uintptr_t>(rbar_ + pb * CP + ec);
        mu[j] = ~(uintptr_t)0;
        su_step[j] = (unsigned long long)CP * 8;
        if (j == NI - 1 && lane >= 62) {
            cu[j] = reinterpret_cast<uintptr_t>((lane == 63 ? dbar_ : de_) + pb);
            mu[j] = ~(uintptr_t)15;
            su_step[j] = 8;
        }
    }
    const unsigned long long r_step = (unsigned long long)CP * 8;
    auto issue = [&](const int64_t m, const int after) {      // row m (the cursors stand at it) -> slot m mod D
        const unsigned so = (unsigned)(m & (D - 1)) * (SL * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            glds16(reinterpret_cast<const void *>(cu[j] & mu[j]), lds_u + so + 1024 * j, after);
            glds16(reinterpret_cast<const void *>(cr[j]), lds_r + so + 1024 * j, after);
            cu[j] += su_step[j];
            cr[j] += r_step;
        }
    };
#pragma unroll
    for (int m = 0; m < D - 1; ++m) issue(m, 0);
    double T[TR][2];
#pragma unroll
    for (int m2 = 0; m2 < TR; ++m2) {               // Phi = I
        const int row = 16 * (m2 >> 1) + 2 * g + (m2 & 1);
        T[m2][0] = (row == 2 * cb) ? 1.0 : 0.0;
        T[m2][1] = (row == 2 * cb + 1) ? 1.0 : 0.0;
    }
    double q0 = 0.0, q1 = 0.0;
    wave_lds_fence();
    for (int64_t n = 0; n < rows; ++n) {
        // row n has landed: the operations issued after its copies are the (D - 2) rows behind it and, once
        // the loop is in its steady state, one h store per iteration
        if (n < D - 2) vm_wait<(D - 2) * 2 * NI>();
        else vm_wait<(D - 2) * (2 * NI + 1)>();
        const int sl = (int)(n & (D - 1)), par = (int)((pb + n) & 1);
        const double dcur = ring_u[sl][SD + par];
        const double de = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(ring_u[sl][SE + par])),
                                           __builtin_amdgcn_readfirstlane(__double2loint(ring_u[sl][SE + par])));
        const double2 *pu = (const double2 *)&ring_u[sl][0] + g;
        const double2 *pw = (const double2 *)&ring_r[(n == 0) ? D : (int)((n - 1) & (D - 1))][0] + g;
        if (de >= 0.0) {                    // reset row: Phi <- E (Phi + pending): row scaling only
#pragma unroll
            for (int k = 0; k < NL; ++k) s_e[lane + 64 * k] = fm_exp_local(-ci[k] * de);
            wave_lds_fence();
            double d0, d1;
            sweepw_run<TR, true, 8>(T, pe, pw, q0, q1, 1.0, 1.0, d0, d1);
            q0 = 0.0;
            q1 = 0.0;
        }
        double acc0, acc1;
        sweepw_run<TR, false, 8>(T, pu, pw, q0, q1, 0.0, 0.0, acc0, acc1);
        const double h = own_column_sum8(acc0, acc1);
        hg[(size_t)n * CP] = h;             // (the four owner lanes of a column write the same value)
        both_halves(-h * fast_rcp(dcur), q0, q1);      // pending: Phi_i -= (r_i / d) h_j, row n's r from its ring slot
        // row n + D - 1 into the slot of row n - 1, whose r~ this iteration's sweeps were the last to read
        issue(n + D - 1, __double2loint(h));
    }
    vm_wait<0>();                           // no copy may outlive the wave's LDS allocation
    const double *s_w = &ring_r[(int)((rows - 1) & (D - 1))][0];
    double *__restrict__ col0 = Phi_out + (size_t)slot * ((size_t)CP * RP) + (size_t)(2 * cb) * RP + 2 * g;
    double *__restrict__ col1 = col0 + RP;
#pragma unroll
    for (int m2 = 0; m2 < TR; ++m2) {
        const int ro = 16 * (m2 >> 1) + (m2 & 1);
        const double w = s_w[2 * g + ro];
        col0[ro] = fma(w, q0, T[m2][0]);
        col1[ro] = fma(w, q1, T[m2][1]);
    }
}

// (the archived sweep variants k_factor4/5/6 of rounds 1-3 live outside the package: tools/archive/sweeps_456.inc)

// ------------------------------------------------------------------------------------
// Exact time-parallel evaluation of ONE series (Lainiotis-type partitioning; the numpy
// derivation and its verification against the sequential recurrence are in DESIGN.md 4.3).
// The series is cut into nch chunks.  With X = S (+ pending update) handed from chunk to chunk:
//   nominal pass  (k_factor3, zero start)      : Xbar_end, Ybar_end, dbar, zbar, rbar per chunk
//   k_phi         : closed-loop transition      Phi <- (I - w~ u~^T) Phi,  h_n = Phi^T u~_n
//   k_gram        : G = sum h h^T / dbar ,  m = sum h zbar / dbar
//   k_combine     : X+ = Xbar_end + Phi K Phi^T ,  K = (I - X G)^-1 X          (sequential in c)
//                   Y+ = Ybar_end + Phi (I - X G)^-1 (Y - X m)
//   final pass    (k_factor3 from the true X, Y): d, z  -- identical to the sequential run
// k_phi is the same register-resident sweep as the factor: Phi_i += (r~_{n-1,i}) * (-h_{n-1}/d),
// h_n = sum_i u~_{n,i} Phi_i, with no reduction and no division on its chain.
// ------------------------------------------------------------------------------------
template <int ROWS>
__global__ void __launch_bounds__(64, 2)
k_phi(const int64_t N, const int64_t chunk_len, const int nch, const int ch0, const int nsel, const int W,
      const double *__restrict__ c_, const double *__restrict__ de_, const double *__restrict__ dbar_,
      const double *__restrict__ zbar_, const double *__restrict__ rbar_, const double *__restrict__ ut_,
      double *__restrict__ Phi_out, double *__restrict__ G_out, double *__restrict__ m_out) {
    constexpr int NT = (ROWS + 15) / 16, PD = PhiRing::PD;
    const int lane = threadIdx.x;
    const int sel = blockIdx.x;
    const int pr = sel / nsel, ch = ch0 + (sel - pr * nsel);
    const int b = pr * nch + ch;
    const int64_t c0 = (int64_t)ch * chunk_len;
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const size_t pb = (size_t)pr * N + c0;
    double *__restrict__ Pg = Phi_out + (size_t)b * (64 * 64) + (size_t)lane * 64;
    const double cj = (lane < W) ? c_[(size_t)pr * W + lane] : 0.0;

    __shared__ __attribute__((aligned(16))) PhiRing R;
    __shared__ __attribute__((aligned(16))) PhiGram<NT> Gm;
    __shared__ __attribute__((aligned(16))) double s_e[64];
    double T[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) T[i] = (i == lane) ? 1.0 : 0.0;      // Phi = I
    double q = 0.0, macc = 0.0;
    d4 acc[PhiGram<NT>::NACC];
#pragma unroll
    for (int t = 0; t < PhiGram<NT>::NACC; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    R.r[PD][lane] = 0.0;
    for (int e = lane; e < 16 * PhiGram<NT>::LD; e += 64) (&Gm.hs[0][0])[e] = 0.0;
    if (lane < 16) Gm.sv[lane] = 0.0;
    PhiCursor C;
    C.template init<ROWS>(R, lane, pb, ut_, rbar_, dbar_, de_, zbar_);
#pragma unroll
    for (int m = 0; m < PD - 1; ++m) C.issue(m, 0);
    vm_wait<2 * (PD - 2)>();
    wave_lds_fence();
    double ab[SW_AHEAD + 1][SW_BR], wb[SW_AHEAD + 1][SW_BR];
    const double *su = &R.u[0][0], *sw = &R.r[PD][0];
    sweep_preload<ROWS>(ab, wb, su, sw);

    for (int64_t n = 0; n < rows; ++n) {
        vm_wait<2 * (PD - 3)>();
        const int sl = (int)(n & (PD - 1)), par = (int)((pb + n) & 1);
        const double dcur = R.u[sl][PhiRing::OD + par], zcur = R.u[sl][PhiRing::OZ + par];
        const double dev = R.u[sl][PhiRing::OE + par];
        const double de = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dev)),
                                           __builtin_amdgcn_readfirstlane(__double2loint(dev)));
        if (de >= 0.0) {                    // Phi <- E (Phi + pending): row scaling only
            s_e[lane] = fm_exp_local(-cj * de);
            wave_lds_fence();
            sweep_preload<ROWS>(ab, wb, s_e, sw);
            (void)sweep_run<ROWS, true>(T, ab, wb, s_e, sw, q, 1.0);
            sweep_preload<ROWS>(ab, wb, su, sw);
            q = 0.0;
        }
        const double h = sweep_run<ROWS, false>(T, ab, wb, su, sw, q, 0.0);
        const double rinv = (dcur > 0.0) ? fast_rcp(dcur) : 0.0;
        Gm.hs[n & 15][lane] = h;
        if (lane == 0) Gm.sv[n & 15] = rinv;
        macc = fma(h, zcur * rinv, macc);
        su = &R.u[(int)((n + 1) & (PD - 1))][0];
        sw = &R.r[sl][0];
        sweep_preload<ROWS>(ab, wb, su, sw);
        q = -h * rinv;                      // pending: Phi_i -= (r_i / d) h_j
        C.issue(n + PD - 1, __double2loint(h));
        if ((n & 15) == 15) phi_gram_flush<NT>(Gm, acc, lane);
    }
    vm_wait<0>();
    if (rows & 15) phi_gram_flush<NT>(Gm, acc, lane);
    wave_lds_fence();
    const double *s_w = &R.r[(int)((rows - 1) & (PD - 1))][0];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) Pg[i] = fma(s_w[i], q, T[i]);
#pragma unroll
    for (int i = ROWS; i < 64; ++i) Pg[i] = 0.0;
    phi_gram_store<NT>(acc, macc, G_out + (size_t)b * (64 * 64), m_out + (size_t)b * 64, lane, lane);
}

// ---- dense helpers for the chunk combines (one workgroup of 256 threads) ---------------------
// LDS matrices are NS x NS, row-major with the odd leading dimension NS + 1 (conflict-free rows); global
// matrices are 64 x 64 slots stored [j][i] (column-major = the lane-major layout of the sweep states), zero
// beyond the width.  NS = 64, 48 for the widths <= 48, 32 for those <= 32: 135 / 77 / 36 KB of LDS per workgroup,
// one / two / four workgroups per CU -- the levels of the tree scan with more pairs than CUs run in half / a
// quarter of the rounds.
template <int NS> struct CbDims {
    static constexpr int LD = NS + 1;           // a matrix' row stride
    static constexpr int LA = 2 * NS + 2;       // the augmented system's: [A | RHS | one vector]
    static constexpr int VC = 2 * NS;           // the vector's column
    static constexpr int NC = NS / 2 + 1;       // Gauss-Jordan: registers per lane (columns c = 4 lc + wave <= VC)
    static constexpr int R0 = NS / 4;           // ... the first one of the right-hand sides
};

template <int NS = 64>
__device__ __forceinline__ void cb_load(double *dst, const double *__restrict__ src, int tid,
                                        int ld = CbDims<NS>::LD) {
    if (NS < 64 && (tid & 63) >= NS) return;            // (element e = [j][i]: row i = e & 63, column j = e >> 6)
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) { const int e = tid + 256 * q, j = e >> 6, i = e & 63; dst[i * ld + j] = src[e]; }
}

// 64 x 64 x 64 products on v_mfma_f64_16x16x4 (256 threads = 4 waves; wave w owns rows 16w..16w+15
// of the result as four accumulator tiles): acc[q][r] = element (cb_row(ty, r), cb_col(tx, q)) of
// A' B' with A'(row, k) = a_of(row, k), B'(k, col) = b_of(k, col) read from LDS -- two LDS reads per
// 1024 FMAs instead of one per two with 4x4 register blocks (which ran LDS-bound).
__device__ __forceinline__ int cb_row(int ty, int r) { return 16 * (ty >> 2) + (ty & 3) + 4 * r; }
__device__ __forceinline__ int cb_col(int tx, int q) { return 16 * q + tx; }

// nt (wave-uniform, 1 .. 4): only the leading 16 nt rows / columns of the operands are non-zero (states of
// width W <= 16 nt, zero-padded to 64): the other tiles of the result are zeros and are not computed.
template <class FA, class FB>
__device__ __forceinline__ void cb_mm(double (&acc)[4][4], FA a_of, FB b_of, int tx, int ty, const int nt = 4) {
    const int i = tx, k = ty & 3, w = ty >> 2;          // lane = (i, k) of wave w
    d4 C[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) C[q] = d4{0.0, 0.0, 0.0, 0.0};
    if (nt == 4) {
        // the full width as straight-line code: with a loop the accumulators travel between the vector and the
        // accumulation registers on every trip (64 moves per 16 MFMAs)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double av = a_of(16 * w + i, 4 * ks + k);
#pragma unroll
            for (int q = 0; q < 4; ++q) C[q] = GF_MFMA64(av, b_of(4 * ks + k, 16 * q + i), C[q]);
        }
    } else if (nt == 3) {                               // (widths 33 .. 48: cfg3's W = 40)
        if (w < 3) {
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const double av = a_of(16 * w + i, 4 * ks + k);
#pragma unroll
                for (int q = 0; q < 3; ++q) C[q] = GF_MFMA64(av, b_of(4 * ks + k, 16 * q + i), C[q]);
            }
        }
    } else if (w < nt) {
#pragma unroll 4
        for (int ks = 0; ks < 4 * nt; ++ks) {
            const double av = a_of(16 * w + i, 4 * ks + k);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < nt) C[q] = GF_MFMA64(av, b_of(4 * ks + k, 16 * q + i), C[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = C[q][r];
}

// acc = A' B' with A' = A or A^T (TA), B' = B or B^T (TB), row-major LDS matrices
template <bool TA, bool TB>
__device__ __forceinline__ void cb_matmul(double (&acc)[4][4], const double *A, int lda,
                                          const double *B, int ldb, int tx, int ty, const int nt = 4) {
    cb_mm(acc,
          [&](int row, int k) { return TA ? A[k * lda + row] : A[row * lda + k]; },
          [&](int k, int col) { return TB ? B[col * ldb + k] : B[k * ldb + col]; }, tx, ty, nt);
}

template <int NS = 64>
__device__ __forceinline__ void cb_store_lds(double *dst, int ld, const double (&acc)[4][4], int tx, int ty) {
    if (NS < 64 && 16 * (ty >> 2) >= NS) return;        // (wave ty >> 2 holds the rows 16 (ty >> 2) ...)
#pragma unroll
    for (int q = 0; q < NS / 16; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[cb_row(ty, r) * ld + cb_col(tx, q)] = acc[q][r];
}

// Gauss-Jordan with IMPLICIT partial pivoting on the 64 x 129 augmented system Au = [A | RHS],
// held in registers with lanes = rows: wave w (of 4) owns the columns c = w (mod 4), 33 registers
// per lane.  Step k: the wave that owns column k searches the pivot among the unused rows, forms the
// row multipliers f_i = a_ik / a_pk (lane form, f_p = 0) and publishes them with the pivot's lane
// index through LDS; after ONE barrier every wave eliminates its columns, fetching the pivot row's
// entries with readlane (an SGPR operand of the FMA: no LDS traffic in the update).  Rows are never
// swapped or normalised, eliminated columns are not revisited (updating them is harmless and keeps
// the loops compile-time).  The multiplier buffer alternates between two halves so that the next
// owner may write while slower waves still read.
// The search runs one step AHEAD: behind the barrier of step k the owner of column k + 1 updates that
// column first, searches and publishes step k + 1, and only then eliminates the rest of its columns --
// while the other three waves are still eliminating for step k.  And its dependent chain is short: the
// magnitudes compare as 32-bit keys (the high word of |a|: exponent + 20 mantissa bits -- a pivot within
// 1e-6 of the largest is as good), a DPP max per step inside the rows of 16 lanes and scalar maxima
// across them; every lane has the reciprocal of its own entry ready before the pivot is known.  (A step
// took 1290 cycles with the search of a 64-bit maximum, its reciprocal and the whole update in sequence;
// tools/_dbg_tree.py.)
// scratch >= 140 doubles.  On exit Au[:, 64:129] = A^-1 RHS.
// n (uniform, <= 64): the system is the identity beyond its leading n x n block with zero right-hand sides there
// (zero-padded states): the steps k >= n are skipped and those rows of the solution are zeros.
template <int NS = 64>
__device__ __forceinline__ void cb_gauss_jordan(double *Au, double *scratch, int tid, const int n = NS) {
    constexpr int LA = CbDims<NS>::LA, NC = CbDims<NS>::NC, VC = CbDims<NS>::VC, R0 = CbDims<NS>::R0;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *fbuf = scratch;                                     // [2][64] multipliers
    double *pinvbuf = scratch + 128;                            // [2]
    int *pvbuf = reinterpret_cast<int *>(scratch + 130);        // [2]
    const bool dead = NS < 64 && lane >= NS;                    // (no such row)
    double R[NC];
#pragma unroll
    for (int lc = 0; lc < NC; ++lc) {
        const int c = 4 * lc + w;
        R[lc] = (c <= VC && !dead) ? Au[lane * LA + c] : 0.0;
    }
    bool used = dead;                       // row `lane` already served as a pivot (or does not exist)
    int myk = 0;                            // ... of which variable
    double mypinv = 0.0;                    // ... with which 1 / pivot
    if (w == 0 && n > 0) gj_search(R[0], used, lane, fbuf, pvbuf, pinvbuf);
    static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value, lk = k >> 2, buf = k & 1;
        constexpr int k1 = k + 1, w1 = k1 & 3, lk1 = k1 >> 2, buf1 = k1 & 1;
        if (k >= n) return;
        __syncthreads();
        const double f = fbuf[buf * 64 + lane];
        const int pv = __builtin_amdgcn_readfirstlane(pvbuf[buf]);
        if (lane == pv) { used = true; myk = k; mypinv = pinvbuf[buf]; }
        if (k1 < NS && k1 < n && w == w1) {                     // (wave-uniform) the next column's owner
            R[lk1] = fma(-f, read_lane(R[lk1], pv), R[lk1]);
            gj_search(R[lk1], used, lane, fbuf + buf1 * 64, pvbuf + buf1, pinvbuf + buf1);
            gj_update<lk, NC, lk1>(R, f, pv);
        } else {
            gj_update<lk, NC, -1>(R, f, pv);
        }
    }, std::make_integer_sequence<int, NS>{});
    // row `lane` solved variable myk:  x(myk, :) = (right part of the row) / pivot
    const bool piv = used && !dead;
#pragma unroll
    for (int lc = R0; lc < NC; ++lc) {
        const int c = 4 * lc + w;
        if (c <= VC && !dead) Au[(piv ? myk : lane) * LA + c] = piv ? R[lc] * mypinv : 0.0;   // (rows >= n: never pivots)
    }
    __syncthreads();
}

// One application of a chunk map to a state, in LDS:  on entry Xs = X, Ys = Y (LDS); Phi, G, m,
// Xbar, Ybar of the map are read from global.  On exit Xs = X+, Ys = Y+.
//   A = I - X G ;  [K | v] = A^-1 [X | Y - X m] ;  X+ = Xbar + Phi K Phi^T ;  Y+ = Ybar + Phi v
template <int NS = 64>
__device__ __forceinline__ void cb_apply(double *Xs, double *Au, double *Bs, double *Ys, double *vs,
                                         const double *__restrict__ Pg, const double *__restrict__ Gg,
                                         const double *__restrict__ mg, const double *__restrict__ Xbar,
                                         const double *__restrict__ Ybar, const double (&xreg)[16],
                                         const double yreg, int tid, const int n = NS) {
    const int nt = (n + 15) >> 4;             // (maps of width n <= NS, zero-padded: cb_mm / cb_gauss_jordan)
    // Xbar/Ybar: global pointers, or nullptr to take them from registers (xreg[q] = element
    // e = tid + 256 q of the [j][i] layout, yreg = element tid)
    constexpr int LD = CbDims<NS>::LD, LA = CbDims<NS>::LA, VC = CbDims<NS>::VC;
    const int tx = tid & 15, ty = tid >> 4;
    cb_load<NS>(Bs, Gg, tid);
    if (tid < 64) vs[tid] = mg[tid];
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, Xs, LD, Bs, LD, tx, ty, nt);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int i = cb_row(ty, cc), j = cb_col(tx, a);
                if (NS == 64 || (i < NS && j < NS)) {
                    Au[i * LA + j] = ((i == j) ? 1.0 : 0.0) - acc[a][cc];
                    Au[i * LA + NS + j] = Xs[i * LD + j];
                }
            }
    }
    if (tid < NS) {                         // rhs = Y - X m
        double sacc = Ys[tid];
        for (int k = 0; k < n; ++k) sacc = fma(-Xs[tid * LD + k], vs[k], sacc);
        Au[tid * LA + VC] = sacc;
    }
    __syncthreads();
    cb_gauss_jordan<NS>(Au, Bs, tid, n);
    // Bs <- Phi ; Xs <- Z = Phi K  (K symmetrised)
    cb_load<NS>(Bs, Pg, tid);
    if (tid < NS) vs[tid] = Au[tid * LA + VC];
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_mm(acc,
              [&](int row, int k) { return Bs[row * LD + k]; },
              [&](int k, int col) { return 0.5 * (Au[k * LA + NS + col] + Au[col * LA + NS + k]); },
              tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Xs, LD, acc, tx, ty);
    }
    double ynew = 0.0;
    if (tid < NS) {                         // Y+ = Ybar + Phi v
        ynew = Ybar ? Ybar[tid] : yreg;
        for (int k = 0; k < n; ++k) ynew = fma(Bs[tid * LD + k], vs[k], ynew);
    }
    __syncthreads();
    {                                       // X+ = Xbar + Z Phi^T  -> Au left half as staging
        double acc[4][4] = {};
        cb_matmul<false, true>(acc, Xs, LD, Bs, LD, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Au, LA, acc, tx, ty);
    }
    __syncthreads();
    if (NS == 64 || (tid & 63) < NS) {
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {          // add Xbar (coalesced) and symmetrise
            const int e = tid + 256 * q, j = e >> 6, i = e & 63;
            Xs[i * LD + j] = (Xbar ? Xbar[e] : xreg[q]) + 0.5 * (Au[i * LA + j] + Au[j * LA + i]);
        }
    }
    if (tid < 64) Ys[tid] = ynew;
    __syncthreads();
}

// Sequential LFT combine over the chunks of each problem (one workgroup per problem).
// S_state/F_state slot c holds (Xbar_end, Ybar_end) of chunk c on entry and the TRUE start
// state of chunk c on exit (slot 0 <- 0).
template <int NS>                       // LDS matrix size: 64 (widths beyond 48: compile-time bounds), 48 or 32
__global__ void __launch_bounds__(256)
k_combine(const int nch, const int n_, const double *__restrict__ Phi_, const double *__restrict__ G_,
          const double *__restrict__ m_, double *__restrict__ S_state,
          double *__restrict__ F_state) {
    constexpr bool FULL = NS == 64;
    constexpr int LD = CbDims<NS>::LD, LA = CbDims<NS>::LA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Xs = lds;                       // [NS][LD]
    double *Au = Xs + NS * LD;              // [NS][LA]
    double *Bs = Au + NS * LA;              // [NS][LD]
    double *Ys = Bs + NS * LD;              // [64]
    double *vs = Ys + 64;                   // [64]
    const int pr = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = FULL ? 64 : n_;
    const bool in = NS == 64 || (tid & 63) < NS;    // this thread's elements e = tid + 256 q lie in rows < NS
    for (int e = tid; e < NS * LD; e += 256) Xs[e] = 0.0;
    if (tid < 64) Ys[tid] = 0.0;
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const size_t slot = (size_t)pr * nch + c;
        double *Sg = S_state + slot * 4096;
        double *Fg = F_state + slot * 64;
        // keep this chunk's nominal end state, publish its TRUE start state (beyond NS the slots hold zeros, and keep them)
        double xbar[16], ybar = 0.0;
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
            const int e = tid + 256 * q, j = e >> 6, i = e & 63;
            if (in) { xbar[q] = Sg[e]; Sg[e] = Xs[i * LD + j]; }
        }
        if (tid < 64) { ybar = Fg[tid]; Fg[tid] = Ys[tid]; }
        if (c == nch - 1) break;
        if (c == 0) {
            // from the zero state a chunk's map returns its own nominal end state (K = 0, v = 0): no solve
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NS / 4; ++q) {
                const int e = tid + 256 * q, j = e >> 6, i = e & 63;
                if (in) Xs[i * LD + j] = xbar[q];
            }
            if (tid < 64) Ys[tid] = ybar;
            __syncthreads();
            continue;
        }
        cb_apply<NS>(Xs, Au, Bs, Ys, vs, Phi_ + slot * 4096, G_ + slot * 4096, m_ + slot * 64,
                     nullptr, nullptr, xbar, ybar, tid, n);
    }
}

// ---- log-depth tree combine (Blelloch scan over the chunk maps) -----------------------------
// Map of a chunk: M = (Phi, G, Xbar, Ybar, m).  Up-sweep: map[right] <- map[right] o map[left]
//   D = (I - Xbar1 G2)^-1 ;  Phi12 = Phi2 D Phi1 ;  Xbar12 = Xbar2 + Phi2 D Xbar1 Phi2^T
//   G12 = G1 + Phi1^T G2 D Phi1 ;  v = D (Ybar1 - Xbar1 m2) ;  Ybar12 = Ybar2 + Phi2 v
//   m12 = m1 + Phi1^T (m2 - G2 v)                      (1 = left, applied first; 2 = right)
// Down-sweep on states: s[left] <- s[right] ; s[right] <- map[left](s[right]).
// Slots are (problem, index) with P = power-of-two indices per problem; identity maps pad.
struct TreeArgs {
    int P, d, n;                            // level: pairs (idx - d, idx), idx = (k+1) 2d - 1; n = state width
    int nch, pairs;                         // real chunks per problem; pairs launched per problem at this level
    double *Phi, *G, *S, *F, *m;            // maps   [B*nch][...]: the slots idx < nch, where the sweeps put them
    double *Xst, *Yst;                      // states [B*nch][4096] / [64]
    double *oPhi, *oG, *oS, *oF, *om, *oX, *oY;     // the slots nch <= idx < P (composites that reach into the padding
                                                    // and their states): [B*(P - nch)][...] in the caller's work buffer
};
struct TreeSlot { double *Phi, *G, *S, *F, *m, *X, *Y; };
__device__ __forceinline__ TreeSlot tree_slot(const TreeArgs &A, const int pr, const int idx) {
    TreeSlot t;
    if (idx < A.nch) {
        const size_t s = (size_t)pr * A.nch + idx;
        t.Phi = A.Phi + s * 4096; t.G = A.G + s * 4096; t.S = A.S + s * 4096; t.X = A.Xst + s * 4096;
        t.F = A.F + s * 64; t.m = A.m + s * 64; t.Y = A.Yst + s * 64;
    } else {
        const size_t s = (size_t)pr * (A.P - A.nch) + (idx - A.nch);
        t.Phi = A.oPhi + s * 4096; t.G = A.oG + s * 4096; t.S = A.oS + s * 4096; t.X = A.oX + s * 4096;
        t.F = A.oF + s * 64; t.m = A.om + s * 64; t.Y = A.oY + s * 64;
    }
    return t;
}

template <int NS>
__global__ void __launch_bounds__(256) k_tree_compose(const TreeArgs A) {
    constexpr bool FULL = NS == 64;
    constexpr int LD = CbDims<NS>::LD, LA = CbDims<NS>::LA, VC = CbDims<NS>::VC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *M0 = lds;                       // [NS][LD]
    double *Au = M0 + NS * LD;              // [NS][LA]  = AL | AR | one vector
    double *M1 = Au + NS * LA;              // [NS][LD]
    double *v1 = M1 + NS * LD;              // [64] x 4 small vectors
    const int pairs = A.pairs;              // (the pairs whose left range holds a real chunk: the others are not launched)
    const int pr = blockIdx.x / pairs, k = blockIdx.x - pr * pairs;
    const int ir = (k + 1) * 2 * A.d - 1, il = ir - A.d;
    const TreeSlot SL = tree_slot(A, pr, il), SR = tree_slot(A, pr, ir);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int n = FULL ? 64 : A.n, nt = FULL ? 4 : ((n + 15) >> 4);
    if (il + 1 >= A.nch) {
        // the right range [il + 1, ir] is padding (an identity map, nowhere stored): the composite is the left map
        for (int e = tid; e < 4096; e += 256) {
            SR.Phi[e] = SL.Phi[e];
            SR.G[e] = SL.G[e];
            SR.S[e] = SL.S[e];
        }
        if (tid < 64) { SR.F[tid] = SL.F[tid]; SR.m[tid] = SL.m[tid]; }
        return;
    }
    const bool in = NS == 64 || (tid & 63) < NS;    // this thread's slot elements e = tid + 256 q lie in rows < NS
    const bool vin = tid < NS;                      // ... its vector element exists
    double *w = v1, *vv = v1 + 64, *g2v = v1 + 128, *tmpv = v1 + 192;
    // a. M0 = Xbar1, M1 = G2 ;  AL = I - Xbar1 G2 ; AR = I ; the vector column = Ybar1 - Xbar1 m2
    cb_load<NS>(M0, SL.S, tid);
    cb_load<NS>(M1, SR.G, tid);
    if (tid < 64) tmpv[tid] = SR.m[tid];
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, M0, LD, M1, LD, tx, ty, nt);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = cb_row(ty, c), j = cb_col(tx, a);
                if (NS == 64 || (i < NS && j < NS)) {
                    Au[i * LA + j] = ((i == j) ? 1.0 : 0.0) - acc[a][c];
                    Au[i * LA + NS + j] = (i == j) ? 1.0 : 0.0;
                }
            }
    }
    if (vin) {
        double sacc = SL.F[tid];
        for (int kk = 0; kk < n; ++kk) sacc = fma(-M0[tid * LD + kk], tmpv[kk], sacc);
        Au[tid * LA + VC] = sacc;
    }
    __syncthreads();
    cb_gauss_jordan<NS>(Au, v1, tid, n);    // AR = D, the vector column = v   (scratch: w | vv | g2v, not yet in use;
                                            // G2 stays where it is -- with M1 as the scratch it was loaded twice)
    if (tid < 64) vv[tid] = vin ? Au[tid * LA + VC] : 0.0;
    __syncthreads();
    if (tid < 64) {                         // g2v = G2 v ;  m12 pieces need it
        double sacc = 0.0;
        if (vin)
            for (int kk = 0; kk < n; ++kk) sacc = fma(M1[tid * LD + kk], vv[kk], sacc);
        g2v[tid] = sacc;
    }
    // c. AL = D Xbar1
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, Au + NS, LA, M0, LD, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Au, LA, acc, tx, ty);
    }
    __syncthreads();
    // d. M0 = Phi2 ;  AL <- Phi2 (D Xbar1) ;  Xbar12 = Xbar2 + AL Phi2^T ;  Ybar12 = Ybar2 + Phi2 v
    cb_load<NS>(M0, SR.Phi, tid);
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, M0, LD, Au, LA, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Au, LA, acc, tx, ty);
    }
    if (tid < 64) {
        double sacc = SR.F[tid];
        if (vin)
            for (int kk = 0; kk < n; ++kk) sacc = fma(M0[tid * LD + kk], vv[kk], sacc);
        w[tid] = sacc;                      // Ybar12 (stored at the end)
    }
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, true>(acc, Au, LA, M0, LD, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Au, LA, acc, tx, ty);
    }
    __syncthreads();
    if (in) {
        double *Sr = SR.S;
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
            const int e = tid + 256 * q, j = e >> 6, i = e & 63;
            Sr[e] += 0.5 * (Au[i * LA + j] + Au[j * LA + i]);
        }
    }
    __syncthreads();
    // e. AL = Phi1 ;  AR <- D Phi1 ;  Phi12 = Phi2 (D Phi1)
    cb_load<NS>(Au, SL.Phi, tid, LA);
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, Au + NS, LA, Au, LA, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(Au + NS, LA, acc, tx, ty);
    }
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, M0, LD, Au + NS, LA, tx, ty, nt);
        double *Pr = SR.Phi;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) Pr[(size_t)cb_col(tx, a) * 64 + cb_row(ty, c)] = acc[a][c];
    }
    __syncthreads();
    // f. M1 <- G2 (D Phi1) ;  G12 = G1 + Phi1^T M1 ;  m12 = m1 + Phi1^T (m2 - G2 v)
    {
        double acc[4][4] = {};
        cb_matmul<false, false>(acc, M1, LD, Au + NS, LA, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(M1, LD, acc, tx, ty);
    }
    __syncthreads();
    {
        double acc[4][4] = {};
        cb_matmul<true, false>(acc, Au, LA, M1, LD, tx, ty, nt);
        __syncthreads();
        cb_store_lds<NS>(M0, LD, acc, tx, ty);  // Phi2 no longer needed
    }
    if (tid < 64) {
        double sacc = SL.m[tid];
        if (vin)
            for (int kk = 0; kk < n; ++kk) sacc = fma(Au[kk * LA + tid], tmpv[kk] - g2v[kk], sacc);
        SR.m[tid] = sacc;
        SR.F[tid] = w[tid];
    }
    __syncthreads();
    if (in) {
        double *Gr = SR.G;
        const double *Gl = SL.G;
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
            const int e = tid + 256 * q, j = e >> 6, i = e & 63;
            Gr[e] = Gl[e] + 0.5 * (M0[i * LD + j] + M0[j * LD + i]);
        }
    }
}

// The top of the down-sweep: the whole range starts from the zero state, and a map applied to the zero state
// returns its own nominal end state (K = 0, v = 0) -- so  s[left half's last slot] <- 0,
// s[last slot] <- (Xbar, Ybar) of the left half's composite: a copy instead of the root's memset and one
// level of k_tree_apply (a level costs one workgroup's latency whatever it holds: 55 us).
__global__ void __launch_bounds__(256) k_tree_top(const TreeArgs A) {
    const TreeSlot l = tree_slot(A, blockIdx.x, A.P / 2 - 1), r = tree_slot(A, blockIdx.x, A.P - 1);
    for (int e = threadIdx.x; e < 4096; e += 256) { l.X[e] = 0.0; r.X[e] = l.S[e]; }
    if (threadIdx.x < 64) { l.Y[threadIdx.x] = 0.0; r.Y[threadIdx.x] = l.F[threadIdx.x]; }
}

template <int NS>
__global__ void __launch_bounds__(256) k_tree_apply(const TreeArgs A) {
    constexpr bool FULL = NS == 64;
    constexpr int LD = CbDims<NS>::LD, LA = CbDims<NS>::LA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Xs = lds;
    double *Au = Xs + NS * LD;
    double *Bs = Au + NS * LA;
    double *Ys = Bs + NS * LD;
    double *vs = Ys + 64;
    const int pairs = A.pairs;
    const int pr = blockIdx.x / pairs, k = blockIdx.x - pr * pairs;
    const int ir = (k + 1) * 2 * A.d - 1, il = ir - A.d;
    const TreeSlot SL = tree_slot(A, pr, il), SR = tree_slot(A, pr, ir);
    const int tid = threadIdx.x;
    double *Xr = SR.X, *Xl = SL.X;
    if (il + 1 >= A.nch) {
        // the right range is padding: nobody reads its states -- the left child inherits the parent's, no solve
        for (int e = tid; e < 4096; e += 256) Xl[e] = Xr[e];
        if (tid < 64) SL.Y[tid] = SR.Y[tid];
        return;
    }
    // s[left] <- s[right] (incoming state) ;  s[right] <- map[left](s[right])
    // (the whole slot is handed to the left child -- its padding too: state slots are not cleared between scans;
    // of the right child's slot the leading NS x NS block is rewritten, the rest holds zeros already)
    const bool in = NS == 64 || (tid & 63) < NS;
    _Pragma("unroll 16")
    for (int e = tid; e < 4096; e += 256) {
        const int j = e >> 6, i = e & 63;
        const double v = Xr[e];
        Xl[e] = v;
        if (NS == 64 || (in && j < NS)) Xs[i * LD + j] = v;
    }
    if (tid < 64) { const double v = SR.Y[tid]; SL.Y[tid] = v; Ys[tid] = v; }
    __syncthreads();
    const double noreg[16] = {};
    cb_apply<NS>(Xs, Au, Bs, Ys, vs, SL.Phi, SL.G, SL.m,
                 SL.S, SL.F, noreg, 0.0, tid, FULL ? 64 : A.n);
    if (in) {
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) { const int e = tid + 256 * q, j = e >> 6, i = e & 63; Xr[e] = Xs[i * LD + j]; }
    }
    if (tid < 64) SR.Y[tid] = Ys[tid];
}

// ------------------------------------------------------------------------------------
// Triangular sweeps on the STORED scaled factor (u~, w~ = r/d rows, d, reset spans de), chunked:
// every (problem, chunk) starts from the state in F_state and leaves its end state there.
// In scaled coordinates the recurrences have no per-row decay (A.6/A.7 with P folded into the
// block scaling); at a reset row the state is multiplied by E = exp(-c de) once:
//   lower  : F <- [E](F + w~_{n-1} z_{n-1}) ;  z_n = y_n - u~_n . F           (ascending)
//   upper  : G <- [E_{n+1}](G + u~_{n+1} z_{n+1}) ;  z_n = y_n - w~_n . G     (descending; the
//            decay of reset row n+1 is crossed when stepping from n+1 down to n)
//   matmul : F <- [E](F + w~_{n-1} y_{n-1}) ;  z_n = y_n + u~_n . F           (ascending)
// with optional input scaling y <- y / d (upper: apply_inverse) or y <- y sqrt(d) (matmul:
// dot_tril).  R = 1: the state is lane-resident, one DPP tree per row.
// ------------------------------------------------------------------------------------
struct LinArgs {
    int64_t N, chunk_len;
    int nch, W, mode, scale, store;     // store: write z (final pass) or only the end state
    const double *c;                    // [B][W]
    const double *Ut, *Wt, *d, *de, *Y; // rows [B][N][64] / [B][N]; Y [B][N][R]
    double *Z;
    double *F_state;                    // [B*nch][64 * R]  (R = 1: [64])
    int R;
};

// Rows in flight per lane.  The sweep itself is a DPP reduction and two FMAs per row (~400 cycles): what it waits
// for is memory.  u~ / w~ of a row are one double per lane each: a register ring LIN1_RING rows deep, every slot
// refilled with the row LIN1_RING ahead as soon as its row is used (static indices: the row loop is unrolled over the
// ring).  The per-row scalars y, de, d are wave-uniform: fetched 64 rows at a time, lane j its row, one block ahead,
// parked in LDS and read back as broadcasts a row before their use.  (Two batches of eight rows with five registers
// per row and lane -- 241 VGPRs -- left every eighth row waiting on HBM: 0.18 of a pass' 0.45 ms at N = 1e6.)
constexpr int LIN1_RING = 24;
__global__ void __launch_bounds__(64) k_lin1(const LinArgs A) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int pr = b / A.nch, ch = b - pr * A.nch;
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t rows = (A.N - c0 < A.chunk_len) ? (A.N - c0) : A.chunk_len;
    const size_t pb = (size_t)pr * A.N + c0;
    const double *__restrict__ Ug = A.Ut + pb * 64 + lane;
    const double *__restrict__ Wg = A.Wt + pb * 64 + lane;
    const double *__restrict__ dg = A.d + pb;
    const double *__restrict__ eg = A.de + pb;
    const double *__restrict__ Yg = A.Y + pb;
    double *__restrict__ Zg = A.Z + pb;
    double *__restrict__ Fg = A.F_state + (size_t)b * 64;
    const double cj = (lane < A.W) ? A.c[(size_t)pr * A.W + lane] : 0.0;
    const bool up = A.mode == GF_SOLVE_UPPER, mm = A.mode == GF_MATMUL_LOWER;
    double F = A.store ? Fg[lane] : 0.0;            // (the local pass starts from zero)
    constexpr int D = LIN1_RING;
    // step s of the sweep works on row n(s): ascending, or descending for the upper solve (clamped: reads past
    // the chunk's last step fetch its last row again and are never used)
    auto row_of = [&](int64_t s) { s = (s > rows - 1) ? rows - 1 : s; return up ? (rows - 1 - s) : s; };
    __shared__ double sc[2][3][64];                 // y, de, d of 64 steps, two blocks
    double ys, es, dv;                              // the block after those: this lane's step
    auto load_sc = [&](const int64_t blk) {
        const int64_t n = row_of(64 * blk + lane);
        ys = Yg[n]; es = eg[n]; dv = A.scale ? dg[n] : 1.0;
    };
    auto put_sc = [&](const int buf) { sc[buf][0][lane] = ys; sc[buf][1][lane] = es; sc[buf][2][lane] = dv; };
    load_sc(0);
    double qu[D], qw[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { const int64_t n = row_of(j); qu[j] = Ug[(size_t)n * 64]; qw[j] = Wg[(size_t)n * 64]; }
    put_sc(0);
    load_sc(1);
    wave_lds_fence();
    double y_n = sc[0][0][0], e_n = sc[0][1][0], d_n = sc[0][2][0];       // step 0's scalars
    double carry = 0.0, prev = 0.0, de_up = -1.0;   // pending F += prev * carry (w~_{n-1} z_{n-1}, or u~_{n+1} z_{n+1})
    for (int64_t s0 = 0; s0 < rows; s0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int64_t s = s0 + j;
            if (s < rows) {                         // (wave-uniform)
                const int64_t n = up ? (rows - 1 - s) : s;
                const double u = qu[j], w = qw[j];
                double yn = y_n;
                const double de = e_n, dd = d_n;
                // the next step's scalars (a block boundary first hands the parked block over)
                const int li = (int)(s & 63);
                if (li == 63) { put_sc((int)(((s >> 6) + 1) & 1)); load_sc((s >> 6) + 2); wave_lds_fence(); }
                {
                    const int64_t s1 = s + 1;
                    const int b1 = (int)((s1 >> 6) & 1), l1 = (int)(s1 & 63);
                    y_n = sc[b1][0][l1]; e_n = sc[b1][1][l1]; d_n = sc[b1][2][l1];
                }
                if (!up) {
                    if (A.scale) yn = mm ? yn * sqrt(dd) : yn / dd;
                    F = fma(prev, carry, F);
                    if (de >= 0.0) F *= fm_exp(-cj * de);
                    const double dot = wave_sum(u * F);
                    const double zn = mm ? (yn + dot) : (yn - dot);
                    if (A.store && lane == 0) Zg[n] = zn;
                    carry = mm ? yn : zn;
                    prev = w;
                } else {
                    // the state handed DOWN to this chunk already carries the decay of the boundary it crossed
                    if (A.scale) yn = yn / dd;
                    F = fma(prev, carry, F);
                    if (de_up >= 0.0) F *= fm_exp(-cj * de_up);
                    const double dot = wave_sum(w * F);
                    const double zn = yn - dot;
                    if (A.store && lane == 0) Zg[n] = zn;
                    carry = zn;
                    prev = u;
                    de_up = de;
                }
            }
            {   // the slot's next tenant: the row LIN1_RING steps ahead
                const int64_t n2 = row_of(s + D);
                qu[j] = Ug[(size_t)n2 * 64];
                qw[j] = Wg[(size_t)n2 * 64];
            }
        }
    }
    F = fma(prev, carry, F);                        // pending folded (lower: decay left to the next chunk)
    if (up && de_up >= 0.0) F *= fm_exp(-cj * de_up);       // upper: cross the chunk's first-row boundary
    Fg[lane] = F;
}

// R right-hand sides: lane r owns column r of Y/Z and F[:, r] (ROWS doubles in VGPRs); the
// generator rows are staged per row in LDS and read back as wave-uniform operands.
template <int ROWS>
__global__ void __launch_bounds__(64) k_linR(const LinArgs A) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;                       // (problem, chunk)
    const int rt = blockIdx.y;                      // RHS tile of 64 columns
    const int pr = b / A.nch, ch = b - pr * A.nch;
    const int R = A.R;
    const int r = rt * 64 + lane;
    const bool rok = r < R;
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t rows = (A.N - c0 < A.chunk_len) ? (A.N - c0) : A.chunk_len;
    const size_t pb = (size_t)pr * A.N + c0;
    const double *__restrict__ Ug = A.Ut + pb * 64 + lane;
    const double *__restrict__ Wg = A.Wt + pb * 64 + lane;
    const double *__restrict__ dg = A.d + pb;
    const double *__restrict__ eg = A.de + pb;
    const double *__restrict__ Yg = A.Y + pb * R;
    double *__restrict__ Zg = A.Z + pb * R;
    double *__restrict__ Fg = A.F_state + (size_t)b * 64 * R;      // [i][R]
    const double cj = (lane < A.W) ? A.c[(size_t)pr * A.W + lane] : 0.0;
    const bool up = A.mode == GF_SOLVE_UPPER, mm = A.mode == GF_MATMUL_LOWER;
    __shared__ double s_a[64], s_b[64], s_e[64];
    double F[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) F[i] = (rok && A.store) ? Fg[(size_t)i * R + r] : 0.0;
    double carry = 0.0;
    s_a[lane] = 0.0;                                // pending push row (w~_{n-1} or u~_{n+1})
    double de_cross = -1.0;
    // everything a row needs from memory is fetched three rows ahead (the loads used to sit at
    // their point of use: one full memory round trip per row, 5 us)
    struct RowIn { double y, pull, push, e, d; };
    auto fetch = [&](int64_t s) {
        if (s > rows - 1) s = rows - 1;
        const int64_t n = up ? (rows - 1 - s) : s;
        RowIn q;
        q.y = rok ? Yg[(size_t)n * R + r] : 0.0;
        q.pull = up ? Wg[(size_t)n * 64] : Ug[(size_t)n * 64];
        q.push = up ? Ug[(size_t)n * 64] : Wg[(size_t)n * 64];
        q.e = eg[n];
        q.d = A.scale ? dg[n] : 1.0;
        return q;
    };
    RowIn p0 = fetch(0), p1 = fetch(1), p2 = fetch(2);
    for (int64_t s = 0; s < rows; ++s) {
        const int64_t n = up ? (rows - 1 - s) : s;
        const RowIn cur = p0;
        p0 = p1; p1 = p2; p2 = fetch(s + 3);
        double yn = cur.y;
        if (A.scale) yn = mm ? yn * sqrt(cur.d) : yn / cur.d;
        const double de = up ? de_cross : cur.e;    // decay to apply before this row's dot
        wave_lds_fence();
        s_b[lane] = cur.pull;
        const bool dec = de >= 0.0;
        if (dec) s_e[lane] = fm_exp(-cj * de);
        wave_lds_fence();
        double dot = 0.0, dot2 = 0.0;
        if (dec) {
#pragma unroll
            for (int i = 0; i < ROWS; i += 2) {
                F[i] = fma(s_a[i], carry, F[i]) * s_e[i];
                dot = fma(s_b[i], F[i], dot);
                F[i + 1] = fma(s_a[i + 1], carry, F[i + 1]) * s_e[i + 1];
                dot2 = fma(s_b[i + 1], F[i + 1], dot2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ROWS; i += 2) {
                F[i] = fma(s_a[i], carry, F[i]);
                dot = fma(s_b[i], F[i], dot);
                F[i + 1] = fma(s_a[i + 1], carry, F[i + 1]);
                dot2 = fma(s_b[i + 1], F[i + 1], dot2);
            }
        }
        dot += dot2;
        const double zn = mm ? (yn + dot) : (yn - dot);
        if (A.store && rok) Zg[(size_t)n * R + r] = zn;
        carry = mm ? yn : zn;
        wave_lds_fence();
        s_a[lane] = cur.push;
        if (up) de_cross = cur.e;
    }
    wave_lds_fence();
    const bool dec = up && de_cross >= 0.0;
    if (dec) s_e[lane] = fm_exp(-cj * de_cross);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        double v = fma(s_a[i], carry, F[i]);
        if (dec) v *= s_e[i];
        if (rok) Fg[(size_t)i * R + r] = v;
    }
}

// k_linR in k_factor7's lane tiling.  k_linR gives every lane ONE right-hand side and all ROWS state
// rows, so each FMA takes a wave-uniform row operand through an LDS broadcast read: 2 ROWS reads per row,
// and with eight waves per CU the LDS array -- not the 2 ROWS FMAs -- set the pace (4 % of the HBM roofline
// on cfg5).  The per-row update is exactly k_factor7's sweep (T += a q^T ; dot += b^T T with T = the
// state, columns = right-hand sides, q = the carries), so it takes the same tiling: lane (g, c) holds the
// right-hand sides 2c, 2c + 1 of half the state rows (pairs 4k + 2g, 4k + 2g + 1), one ds_read_b128
// feeds 8 FMAs (ROWS / 2 reads per row instead of 2 ROWS), and the lane's OWN right-hand side 2c + g
// (its y, z and carry) is connected to its two state columns by the same two v_permlane32_swap pairs.
// NODOT: the local pass of GF_MATMUL_LOWER (dot_tril) -- its carries are the inputs themselves, so the
// pass only accumulates the state: half the FMAs, no reduction.
template <int ROWS, bool NODOT>
__global__ void __launch_bounds__(64, 2) k_linR7(const LinArgs A) {
    const int lane = threadIdx.x;
    const int g = lane >> 5, c = lane & 31;
    const int b = blockIdx.x;                       // (problem, chunk)
    const int rt = blockIdx.y;                      // RHS tile of 64 columns
    const int pr = b / A.nch, ch = b - pr * A.nch;
    const int R = A.R;
    const int r = rt * 64 + 2 * c + g;              // this lane's own right-hand side
    const bool rok = r < R;
    const int r0 = rt * 64 + 2 * c;                 // its two state columns: r0, r0 + 1
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t rows = (A.N - c0 < A.chunk_len) ? (A.N - c0) : A.chunk_len;
    const size_t pb = (size_t)pr * A.N + c0;
    const double *__restrict__ Ug = A.Ut + pb * 64 + lane;
    const double *__restrict__ Wg = A.Wt + pb * 64 + lane;
    const double *__restrict__ dg = A.d + pb;
    const double *__restrict__ eg = A.de + pb;
    const double *__restrict__ Yg = A.Y + pb * R;
    double *__restrict__ Zg = A.Z + pb * R;
    double *__restrict__ Fg = A.F_state + (size_t)b * 64 * R;      // [i][R]
    const double cj = (lane < A.W) ? A.c[(size_t)pr * A.W + lane] : 0.0;
    const bool up = A.mode == GF_SOLVE_UPPER, mm = A.mode == GF_MATMUL_LOWER;
    __shared__ __attribute__((aligned(16))) double s_a[64], s_b[64], s_e[64];
    const double2 *pa = (const double2 *)s_a + g, *pb2 = (const double2 *)s_b + g;
    const double2 *pe = (const double2 *)s_e + g;
    double T[ROWS / 2][2];
#pragma unroll
    for (int m2 = 0; m2 < ROWS / 2; ++m2) {
        const int i = 4 * (m2 >> 1) + 2 * g + (m2 & 1);
        T[m2][0] = (r0 < R && A.store) ? Fg[(size_t)i * R + r0] : 0.0;
        T[m2][1] = (r0 + 1 < R && A.store) ? Fg[(size_t)i * R + r0 + 1] : 0.0;
    }
    double q0 = 0.0, q1 = 0.0, carry = 0.0;
    s_a[lane] = 0.0;                                // pending push row (w~_{n-1} or u~_{n+1})
    double de_cross = -1.0;
    struct RowIn { double y, pull, push, e, d; };
    auto fetch = [&](int64_t s) {
        if (s > rows - 1) s = rows - 1;
        const int64_t n = up ? (rows - 1 - s) : s;
        RowIn q;
        q.y = rok ? Yg[(size_t)n * R + r] : 0.0;
        q.pull = up ? Wg[(size_t)n * 64] : Ug[(size_t)n * 64];
        q.push = up ? Ug[(size_t)n * 64] : Wg[(size_t)n * 64];
        q.e = eg[n];
        q.d = A.scale ? dg[n] : 1.0;
        return q;
    };
    RowIn p0 = fetch(0), p1 = fetch(1), p2 = fetch(2);
    double2 ub[S7_AHEAD + 1], wb[S7_AHEAD + 1];
    for (int64_t s = 0; s < rows; ++s) {
        const int64_t n = up ? (rows - 1 - s) : s;
        const RowIn cur = p0;
        p0 = p1; p1 = p2; p2 = fetch(s + 3);
        double yn = cur.y;
        if (A.scale) yn = mm ? yn * sqrt(cur.d) : yn / cur.d;
        const double de = up ? de_cross : cur.e;    // decay to apply before this row's dot
        wave_lds_fence();
        s_b[lane] = cur.pull;
        const bool dec = de >= 0.0;
        if (dec) s_e[lane] = fm_exp(-cj * de);
        wave_lds_fence();
        if (dec) {                                  // fold the pending push, decay the state rows
            double d0, d1;
            sweep7_preload<ROWS>(ub, wb, pe, pa);
            sweep7_run<ROWS, true>(T, ub, wb, pe, pa, q0, q1, 1.0, 1.0, d0, d1);
            q0 = 0.0;
            q1 = 0.0;
        }
        if constexpr (NODOT) {
#pragma unroll
            for (int k = 0; k < ROWS / 4; ++k) {
                const double2 w = pa[2 * k];
                T[2 * k][0] = fma(w.x, q0, T[2 * k][0]);
                T[2 * k][1] = fma(w.x, q1, T[2 * k][1]);
                T[2 * k + 1][0] = fma(w.y, q0, T[2 * k + 1][0]);
                T[2 * k + 1][1] = fma(w.y, q1, T[2 * k + 1][1]);
            }
            carry = yn;
        } else {
            double acc0, acc1;
            __builtin_amdgcn_s_setprio(0);
            sweep7_preload<ROWS>(ub, wb, pb2, pa);
            sweep7_run<ROWS, false>(T, ub, wb, pb2, pa, q0, q1, 0.0, 0.0, acc0, acc1);
            __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);
            const double dot = own_column_sum(acc0, acc1);
            const double zn = mm ? (yn + dot) : (yn - dot);
            if (A.store && rok) Zg[(size_t)n * R + r] = zn;
            carry = mm ? yn : zn;
        }
        both_halves(carry, q0, q1);                 // the carries of this lane's two state columns
        wave_lds_fence();
        s_a[lane] = cur.push;
        if (up) de_cross = cur.e;
    }
    wave_lds_fence();
    const bool dec = up && de_cross >= 0.0;
    if (dec) s_e[lane] = fm_exp(-cj * de_cross);
    wave_lds_fence();
#pragma unroll
    for (int m2 = 0; m2 < ROWS / 2; ++m2) {
        const int i = 4 * (m2 >> 1) + 2 * g + (m2 & 1);
        double v0 = fma(s_a[i], q0, T[m2][0]), v1 = fma(s_a[i], q1, T[m2][1]);
        if (dec) { v0 *= s_e[i]; v1 *= s_e[i]; }
        if (r0 < R) Fg[(size_t)i * R + r0] = v0;
        if (r0 + 1 < R) Fg[(size_t)i * R + r0 + 1] = v1;
    }
}

// GF_MATMUL_LOWER (dot_tril, GP.sample) with many right-hand sides on the matrix pipe.  The sweep has no
// feedback in this mode (the carries are the inputs), so 16 rows at a time are plain products:
//     Z_b = Y_b + U_b F + strict_lower(U_b W_b^T) Y_b ,     F <- F + W_b^T Y_b
// (U_b, W_b: the block's 16 rows of u~, w~; F: W x R state; decay E o F first when the block starts on a
// reset row; a reset inside a block cuts it there, rows past the cut are masked).  k_linR7 does the same
// arithmetic row by row and runs at the pace of its LDS broadcasts (30 ds_read_b128 per row and wave: 14.9 %
// of the HBM roofline on cfg5).  Here one wave owns 64 right-hand sides of one chunk; everything is
// v_mfma_f64_16x16x4 (lane (i, k) = (lane & 15, lane >> 4) supplies A[i][4 s + k], B[4 s + k][i] of slice s
// and receives D[k + 4 r][i], r = 0..3), laid out so that no result ever changes lanes:
//   * the accumulators of  F += W_b^T Y_b  (tile mt, register r = state 16 mt + k + 4 r) ARE the B
//     operands of  U_b F  (slice s = 4 mt + r <-> state 4 s + k);
//   * the C layout of  G^T = W_b U_b^T  is the A layout of  strict_lower(G) Y_b  (entry (i, k + 4 r));
//   * the B operand rows of Y_b (4 s + k) are the C rows of Z_b (k + 4 r): one load serves both.
// 16 + 36 T MFMAs per block and wave (T = 4 tiles of 16 right-hand sides: the rows are read once), 16 T for
// the local pass; one wave per SIMD (the state alone is 128 accumulator registers).
// The block's u~ / w~ rows (2 x 8 KB, contiguous in memory) come in as flat 16-byte loads issued one block
// ahead, go through an LDS tile (row stride 66: the A-operand reads are conflict-free) and are read slice by
// slice at their point of use.  Taking the operands from global in their MFMA layout instead cost more than
// the MFMAs: a lane-per-row pattern is 64 separate sectors per load instruction (the texture addresser
// serialises them), and a masked load in a branch of its own gets its own s_waitcnt (30 round trips per block).
template <int MT, bool NODOT, int NWV>
__global__ void __launch_bounds__(64 * NWV, (NWV == 1 && !NODOT) ? 1 : 2) k_mmR_mfma(const LinArgs A) {
    constexpr int KS = 4 * MT, T = 2, LDT = 66, NQ = 8 / NWV;
    const int lane = threadIdx.x & 63, i = lane & 15, k = lane >> 4;
    const int wv = (NWV > 1) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int b = blockIdx.x, rt = blockIdx.y;
    const int pr = b / A.nch, ch = b - pr * A.nch;
    const int R = A.R;
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int rows = (int)((A.N - c0 < A.chunk_len) ? (A.N - c0) : A.chunk_len);
    const size_t pb = (size_t)pr * A.N + c0;
    const double *__restrict__ Ug = A.Ut + pb * 64;
    const double *__restrict__ Wg = A.Wt + pb * 64;
    const double *__restrict__ dg = A.d + pb;
    const double *__restrict__ eg = A.de + pb;
    const double *__restrict__ Yg = A.Y + pb * R;
    double *__restrict__ Zg = A.Z + pb * R;
    double *__restrict__ Fg = A.F_state + (size_t)b * 64 * R;      // [state][R]
    __shared__ double s_e[NWV][64];                 // decays of the states at a reset row
    const double c_lane = (lane < A.W) ? A.c[(size_t)pr * A.W + lane] : 0.0;
    // the block's rows, two buffers: a wave that runs ahead stashes block n + 1 while its partner still
    // reads block n (one barrier per block)
    __shared__ __attribute__((aligned(16))) double s_u[NODOT ? 2 : 2 * 16 * LDT], s_w[2 * 16 * LDT];
    int rhs[T], rhc[T];                             // (rhc: clamped, loads are unconditional; masks come after)
    bool rok[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        rhs[t] = (rt * NWV + wv) * 16 * T + 16 * t + i; rok[t] = rhs[t] < R; rhc[t] = rok[t] ? rhs[t] : R - 1;
    }
    d4 F[T][MT];                                    // F[t][mt][r] = state 16 mt + k + 4 r, rhs[t]
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)     // (the local pass starts from zero)
                F[t][mt][r] = NODOT ? 0.0 : Fg[(size_t)(16 * mt + k + 4 * r) * R + rhc[t]];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) F[t][mt][r] = rok[t] ? F[t][mt][r] : 0.0;
    // one block of inputs, as loaded (rows clamped to the chunk; no branch around any load); the waves of
    // the workgroup share the copy of the u~ / w~ rows (row 2 NWV q + 2 wv + (lane >> 5) of the tile)
    double pux[NODOT ? 1 : NQ], puy[NODOT ? 1 : NQ], pwx[NQ], pwy[NQ];
    double e_raw, d_raw, yraw[T][4];
#define GF_MM_FETCH(N0)                                                                             \
    do {                                                                                            \
        const int f_n0 = (N0);                                                                      \
        const int f_lim = (rows - f_n0 < 16) ? rows - f_n0 : 16;                                    \
        e_raw = eg[f_n0 + ((lane < f_lim) ? lane : f_lim - 1)];                                     \
        d_raw = dg[f_n0 + (((lane & 15) < f_lim) ? (lane & 15) : f_lim - 1)];                       \
        _Pragma("unroll")                                                                           \
        for (int s4 = 0; s4 < 4; ++s4) {                                                            \
            const int f_row = 4 * s4 + k;                                                           \
            const size_t f_n = (size_t)(f_n0 + ((f_row < f_lim) ? f_row : f_lim - 1));              \
            _Pragma("unroll")                                                                       \
            for (int t = 0; t < T; ++t) yraw[t][s4] = Yg[f_n * R + rhc[t]];                         \
        }                                                                                           \
        _Pragma("unroll")                                                                           \
        for (int q = 0; q < NQ; ++q) {  /* flat copy: 16-byte pieces of the 16 x 64 tile */         \
            const int f_row = 2 * NWV * q + 2 * wv + (lane >> 5);                                   \
            const size_t f_off = (size_t)(f_n0 + ((f_row < f_lim) ? f_row : f_lim - 1)) * 64 + 2 * (lane & 31); \
            const double2 f_w = *reinterpret_cast<const double2 *>(Wg + f_off);                     \
            pwx[q] = f_w.x; pwy[q] = f_w.y;                                                         \
            if constexpr (!NODOT) {                                                                 \
                const double2 f_u = *reinterpret_cast<const double2 *>(Ug + f_off);                 \
                pux[q] = f_u.x; puy[q] = f_u.y;                                                     \
            }                                                                                       \
        }                                                                                           \
    } while (0)
    GF_MM_FETCH(0);
    int buf = 0;
    for (int n0 = 0; n0 < rows; buf ^= 1) {
        const int lim = (rows - n0 < 16) ? rows - n0 : 16;
        const double *su = s_u + (NODOT ? 0 : buf * 16 * LDT), *sw = s_w + buf * 16 * LDT;
        if (NWV == 1) wave_lds_fence();             // (the block before the previous one has been read)
        // the block: rows n0 .. n0 + cnt - 1, cut at the first reset row after n0
        const double e_l = (lane < lim) ? e_raw : -1.0;
        const unsigned long long inner = __ballot(e_l >= 0.0 && lane >= 1);
        const int cnt = inner ? (int)__ffsll((long long)inner) - 1 : lim;
        const double de0 = read_lane(e_l, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {              // (rows past the cut enter the tile as zeros: no operand masks)
            const int row = 2 * NWV * q + 2 * wv + (lane >> 5);
            const int at = buf * 16 * LDT + row * LDT + 2 * (lane & 31);
            const bool in = row < cnt;
            *reinterpret_cast<double2 *>(&s_w[at]) = double2{in ? pwx[q] : 0.0, in ? pwy[q] : 0.0};
            if constexpr (!NODOT) *reinterpret_cast<double2 *>(&s_u[at]) = double2{in ? pux[q] : 0.0, in ? puy[q] : 0.0};
        }
        double yv[T][4];                            // rows n0 + 4 s + k (B operand of slice s; C rows of Z_b)
        bool okK[4];
        const double sq = A.scale ? sqrt(d_raw) : 1.0;      // lane l: sqrt(d) of row n0 + (l & 15)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            okK[s4] = 4 * s4 + k < cnt;
            const double sc = A.scale ? __shfl(sq, 4 * s4 + k) : 1.0;
#pragma unroll
            for (int t = 0; t < T; ++t) yv[t][s4] = (okK[s4] && rok[t]) ? yraw[t][s4] * sc : 0.0;
        }
        if (NWV == 1) wave_lds_fence(); else wg_lds_barrier();
        // the next block's loads fly under this block's MFMAs (past the chunk's end: the last row again,
        // never used -- an unconditional fetch keeps the staging registers out of scratch)
        GF_MM_FETCH((n0 + cnt < rows) ? n0 + cnt : rows - 1);
        if (de0 >= 0.0) {
            // reset row: F <- E o F before the block's products.  Lane l forms the decay of state l, the
            // lanes pick theirs (state 16 mt + k + 4 r) up from LDS: one exponential and 8 T multiplies per
            // lane.  (As diag(E) F on the matrix pipe -- the form this kernel had while F lived in accumulator
            // registers -- the decay cost 16 T MFMAs per reset: half the local pass when every 16-row block
            // starts on one, as with the solar-like kernels at one-minute cadence.)
            s_e[wv][lane] = fm_exp(-c_lane * de0);
            wave_lds_fence();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double e = s_e[wv][16 * mt + k + 4 * r];
#pragma unroll
                    for (int t = 0; t < T; ++t) F[t][mt][r] *= e;
                }
        }
        if constexpr (!NODOT) {
            d4 G = {0.0, 0.0, 0.0, 0.0};            // G^T = W_b U_b^T: lane (i, k) gets u~_i . w~_{k + 4 r}
            d4 Z[T];
#pragma unroll
            for (int t = 0; t < T; ++t) Z[t] = d4{yv[t][0], yv[t][1], yv[t][2], yv[t][3]};
#pragma unroll
            for (int s = 0; s < KS; ++s) {          // u~, w~[n0 + i][4 s + k]
                const double ua = su[i * LDT + 4 * s + k], wa = sw[i * LDT + 4 * s + k];
                G = GF_MFMA64(wa, ua, G);
#pragma unroll
                for (int t = 0; t < T; ++t) Z[t] = GF_MFMA64(ua, F[t][s >> 2][s & 3], Z[t]);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // (operand reads stay near their use)
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (k + 4 * r >= i) G[r] = 0.0;        // strictly lower
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) Z[t] = GF_MFMA64(G[s4], yv[t][s4], Z[t]);
                if (A.store && rok[t]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k + 4 * r < cnt) Zg[(size_t)(n0 + k + 4 * r) * R + rhs[t]] = Z[t][r];
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {        // w~[n0 + 4 s + k][16 mt + i]
                const double wt = sw[(4 * s4 + k) * LDT + 16 * mt + i];
#pragma unroll
                for (int t = 0; t < T; ++t) F[t][mt] = GF_MFMA64(wt, yv[t][s4], F[t][mt]);
            }
        n0 += cnt;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rok[t]) Fg[(size_t)(16 * mt + k + 4 * r) * R + rhs[t]] = F[t][mt][r];
}
#undef GF_MM_FETCH

constexpr int LC_TRANSPOSED = 0x100;     // (flag on the combine kernels' mode: transitions handed over transposed)

// out[m] = in[m]^T for a batch of 64 x 64 matrices (the transitions' transposes for the backward solve: once per factor)
__global__ void __launch_bounds__(256) k_transpose64(const double *__restrict__ in, double *__restrict__ out) {
    __shared__ double tile[64][65];
    const size_t m = blockIdx.x;
    for (int e = threadIdx.x; e < 4096; e += 256) tile[e >> 6][e & 63] = in[m * 4096 + e];
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) out[m * 4096 + e] = tile[e & 63][e >> 6];
}

// Linear combine of the chunk states of a sweep: on entry F_state slot c holds chunk c's end
// state from a zero start (local pass); on exit it holds the TRUE start state of chunk c.
//   lower  : F_{c+1} = Fbar_c + Phi_c F_c            (ascending; Phi = true closed-loop transition)
//   upper  : G_{c-1} = Gbar_c + Phi_c^T G_c          (descending)
//   matmul : F_{c+1} = Fbar_c + D_c o F_c            (D = product of the chunk's reset decays)
// One workgroup of 64 x RT threads per (problem, RHS tile); Phi is stored [j][i].
//
// Two-level form (PHASE 1 / 2; PHASE 0 is the plain scan over all chunks): the chunks are cut into
// segments of seg_len whose composed transitions Psi_s = Phi_{e-1} ... Phi_b are part of the factor
// (k_segment_transition, once per factor; the backward solve uses Psi_s^T).  PHASE 1 scans every
// segment concurrently from a zero start and leaves its end state in Vseg; a PHASE 0 scan of
// (Psi, Vseg) turns those into the segments' true start states; PHASE 2 rescans every segment from
// there and writes the chunks' start states.  Sequential depth 2 seg_len + nch / seg_len chunks
// instead of nch.
template <int PHASE>
__global__ void __launch_bounds__(256)
k_lincombine(const int nch, const int seg_len, const int mode_, const int R,
             const double *__restrict__ Phi_, const double *__restrict__ Dch_,
             double *__restrict__ F_state, double *__restrict__ Vseg) {
    // mode_ | LC_TRANSPOSED: Phi_ holds the TRANSPOSED transitions (backward solve): its loads then run along the
    // lanes like the forward solve's (a row of Phi^T per lane is one 128-byte run per lane and load: 0.12 against
    // 0.07 ms for the scan of 1954 chunks)
    const int mode = mode_ & ~LC_TRANSPOSED;
    const bool pre_t = (mode_ & LC_TRANSPOSED) != 0;
    // One workgroup of four waves per (problem, right-hand side): the scan is sequential over the
    // chunks only.  Lane i owns state row i; wave w multiplies columns 16w..16w+15 of the chunk's
    // 64 x 64 transition (rows of Phi^T for the backward solve), which it holds in registers and
    // fetches THREE chunks ahead (a ring of four 16-double buffers: the 32 KB per chunk come from
    // L2 with ~2 us latency, which a single buffer exposed on every step: 2.3 us per chunk, now
    // ~0.4); the four partial sums meet in LDS, one barrier per chunk.
    const int r = blockIdx.y, i = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool up = mode == GF_SOLVE_UPPER, mm = mode == GF_MATMUL_LOWER;
    const int nseg = PHASE ? (nch + seg_len - 1) / seg_len : 1;
    const int pr = blockIdx.x / nseg, sg = blockIdx.x - pr * nseg;
    const int c_lo = PHASE ? sg * seg_len : 0;
    const int c_hi = PHASE ? ((c_lo + seg_len < nch) ? c_lo + seg_len : nch) : nch;
    const int len = c_hi - c_lo;
    __shared__ __attribute__((aligned(16))) double s_cur[64];
    __shared__ double s_part[2][4][64];
    double P[4][16];
    auto chunk_of = [&](int s) { return up ? (c_hi - 1 - s) : (c_lo + s); };
    auto load_phi = [&](double (&buf)[16], int s) {
        if (s >= len) return;
        const double *Pg = Phi_ + ((size_t)pr * nch + chunk_of(s)) * 4096;
        if (!up || pre_t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[j] = Pg[(size_t)(16 * w + j) * 64 + i];    // Phi(i, j) (or Phi^T's)
        } else {
            const double2 *row = reinterpret_cast<const double2 *>(Pg + (size_t)i * 64 + 16 * w);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double2 v = row[j]; buf[2 * j] = v.x; buf[2 * j + 1] = v.y; }  // Phi(j, i)
        }
    };
    double cur = 0.0;                               // state row i (carried by wave 0)
    double *Vg = PHASE ? Vseg + ((size_t)pr * nseg + sg) * 64 * R + (size_t)i * R + r : nullptr;
    if (PHASE == 2 && w == 0) cur = *Vg;
    if (PHASE == 0 && mm) {                         // diagonal transitions: one wave does it all
        if (w != 0) return;
        constexpr int GB = 16;                      // (loads of sixteen chunks ahead of their dependent FMAs)
        for (int s0 = 0; s0 < nch; s0 += GB) {
            double loc[GB], dd[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int s = (s0 + k < nch) ? s0 + k : nch - 1;
                const size_t slot = (size_t)pr * nch + s;
                loc[k] = F_state[slot * 64 * R + (size_t)i * R + r];
                dd[k] = Dch_[slot * 64 + i];
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                if (s0 + k < nch) {
                    const size_t slot = (size_t)pr * nch + s0 + k;
                    F_state[slot * 64 * R + (size_t)i * R + r] = cur;
                    cur = fma(dd[k], cur, loc[k]);
                }
            }
        }
        return;
    }
    load_phi(P[0], 0); load_phi(P[1], 1); load_phi(P[2], 2);
    if (threadIdx.x < 64) s_cur[i] = cur;
    __syncthreads();
    auto step = [&](double (&buf)[16], double (&next)[16], int s) {
        const size_t slot = (size_t)pr * nch + chunk_of(s);
        double *Fg = F_state + slot * 64 * R + (size_t)i * R + r;
        double loc = 0.0;
        if (w == 0) { loc = *Fg; if (PHASE != 1) *Fg = cur; }   // local end state in, true start state out
        load_phi(next, s + 3);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            a0 = fma(buf[j], s_cur[16 * w + j], a0);
            a1 = fma(buf[j + 1], s_cur[16 * w + j + 1], a1);
        }
        s_part[s & 1][w][i] = a0 + a1;
        __syncthreads();
        if (w == 0) {                               // wave 0 carries the state
            const double *sp = &s_part[s & 1][0][0];
            cur = loc + ((sp[i] + sp[64 + i]) + (sp[128 + i] + sp[192 + i]));
            s_cur[i] = cur;
        }
        __syncthreads();
    };
    int s = 0;
    for (; s + 4 <= len; s += 4) {
        step(P[0], P[3], s);
        step(P[1], P[0], s + 1);
        step(P[2], P[1], s + 2);
        step(P[3], P[2], s + 3);
    }
    if (s < len) step(P[0], P[3], s);
    if (s + 1 < len) step(P[1], P[0], s + 1);
    if (s + 2 < len) step(P[2], P[1], s + 2);
    if (PHASE == 1 && w == 0) *Vg = cur;            // the segment's end state from a zero start
}

// The segment scans (PHASE 1 / 2 of k_lincombine) for MANY right-hand sides: 64 at a time, the chunk step
// S <- Fbar_c + Phi_c S (Phi_c^T for the backward solve) as one 64 x 64 x 64 product on the matrix pipe.  One
// workgroup per (problem, segment, 64 right-hand sides): the state S [j][r] sits in LDS as the B operand, Phi_c is
// the A operand straight from global -- lane (i, k) of wave w supplies Phi(16 w + i, 4 ks + k) -- fetched one chunk
// ahead together with the chunk's local end state (in accumulator layout: the MFMAs start from it).  One workgroup
// per right-hand side re-read every Phi_c 64 times and took 0.38 ms per phase for 64 right-hand sides at 1954
// chunks (a quarter of predict(return_var=True)); this one streams them once.
template <int PHASE>
__global__ void __launch_bounds__(256)
k_lincombine_R(const int nch, const int seg_len, const int mode_, const int R,
               const double *__restrict__ Phi_, double *__restrict__ F_state, double *__restrict__ Vseg) {
    const int mode = mode_ & ~LC_TRANSPOSED;
    const bool pre_t = (mode_ & LC_TRANSPOSED) != 0;         // (Phi_ holds the transposed transitions)
    constexpr int LDB = 80;                         // (rows k, k + 1 of the B operand 32 banks apart)
    __shared__ __attribute__((aligned(16))) double Sb[64 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, k = lane >> 4;
    const bool up = mode == GF_SOLVE_UPPER;
    const int nseg = (nch + seg_len - 1) / seg_len;
    const int pr = blockIdx.x / nseg, sg = blockIdx.x - pr * nseg;
    const int r0 = 64 * blockIdx.y;
    const int c_lo = sg * seg_len, c_hi = (c_lo + seg_len < nch) ? c_lo + seg_len : nch, len = c_hi - c_lo;
    auto chunk_of = [&](int sidx) { return up ? (c_hi - 1 - sidx) : (c_lo + sidx); };
    // a 64 x 64 block of a state array [64][R] in accumulator layout: x[q][rr] = (row 16 w + k + 4 rr, column r0 + 16 q + i)
    auto ld_state = [&](const double *base, d4 (&x)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = r0 + 16 * q + i;
                x[q][rr] = (col < R) ? base[(size_t)(16 * w + k + 4 * rr) * R + col] : 0.0;
            }
    };
    auto st_state = [&](double *base, const d4 (&x)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = r0 + 16 * q + i;
                if (col < R) base[(size_t)(16 * w + k + 4 * rr) * R + col] = x[q][rr];
            }
    };
    auto ld_A = [&](double (&a)[16], const int sidx) {
        if (sidx >= len) return;
        const double *Pg = Phi_ + ((size_t)pr * nch + chunk_of(sidx)) * 4096;      // Phi(i, j) at [j][i]
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            a[ks] = (up && !pre_t) ? Pg[(size_t)(16 * w + i) * 64 + 4 * ks + k] : Pg[(size_t)(4 * ks + k) * 64 + 16 * w + i];
    };
    auto slot_of = [&](int sidx) { return F_state + ((size_t)pr * nch + chunk_of(sidx)) * 64 * R; };
    d4 cur[4];
    double *Vg = Vseg + ((size_t)pr * nseg + sg) * 64 * R;
    if (PHASE == 2) ld_state(Vg, cur);
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = d4{0.0, 0.0, 0.0, 0.0};
    }
    auto put = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Sb[(16 * w + k + 4 * rr) * LDB + 16 * q + i] = cur[q][rr];
    };
    put();
    double a0[16], a1[16];
    d4 l0[4], l1[4];
    ld_A(a0, 0);
    ld_state(slot_of(0), l0);
    __syncthreads();
    auto step = [&](double (&a)[16], d4 (&loc)[4], double (&an)[16], d4 (&locn)[4], const int sidx) {
        ld_A(an, sidx + 1);
        if (sidx + 1 < len) ld_state(slot_of(sidx + 1), locn);
        if (PHASE == 2) st_state(slot_of(sidx), cur);           // local end state in (above), true start state out
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = loc[q];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double av = a[ks];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = GF_MFMA64(av, Sb[(4 * ks + k) * LDB + 16 * q + i], cur[q]);
        }
        __syncthreads();                            // every wave has read the old state
        put();
        __syncthreads();
    };
    int sidx = 0;
    for (; sidx + 2 <= len; sidx += 2) {
        step(a0, l0, a1, l1, sidx);
        step(a1, l1, a0, l0, sidx + 1);
    }
    if (sidx < len) step(a0, l0, a1, l1, sidx);
    if (PHASE == 1) st_state(Vg, cur);              // the segment's end state from a zero start
}

// Composed transition of every segment of seg_len chunks:  Psi_s = Phi_{e-1} ... Phi_{b+1} Phi_b
// (stored [j][i] like Phi).  One workgroup per segment: the running product lives in LDS (B operand
// of the MFMA tiles), the next chunk's Phi is the A operand straight from global (lanes = rows of a
// column: coalesced).
__global__ void __launch_bounds__(256)
k_segment_transition(const int nch, const int seg_len, const double *__restrict__ Phi_,
                     double *__restrict__ Psi_) {
    const int nseg = (nch + seg_len - 1) / seg_len;
    const int pr = blockIdx.x / nseg, sg = blockIdx.x - pr * nseg;
    const int c_lo = sg * seg_len, c_hi = (c_lo + seg_len < nch) ? c_lo + seg_len : nch;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    __shared__ __attribute__((aligned(16))) double Ps[64 * 64];     // row-major
    {
        const double *Pg = Phi_ + ((size_t)pr * nch + c_lo) * 4096;
        for (int e = tid; e < 4096; e += 256) Ps[(e & 63) * 64 + (e >> 6)] = Pg[e];
    }
    __syncthreads();
    for (int c = c_lo + 1; c < c_hi; ++c) {
        const double *Pg = Phi_ + ((size_t)pr * nch + c) * 4096;
        double acc[4][4];
        cb_mm(acc,
              [&](int row, int k) { return Pg[k * 64 + row]; },
              [&](int k, int col) { return Ps[k * 64 + col]; }, tx, ty);
        __syncthreads();
        cb_store_lds(Ps, 64, acc, tx, ty);
        __syncthreads();
    }
    double *Og = Psi_ + ((size_t)pr * nseg + sg) * 4096;
    for (int e = tid; e < 4096; e += 256) Og[e] = Ps[(e & 63) * 64 + (e >> 6)];
}

// ---- dense batched solve for the chunk maps of WIDE kernels --------------------------------------
// The time-parallel combine of a W > 63 kernel composes dense W x W maps (W <= 192): GEMMs and one
// LU solve per tree level and pair.  The GEMMs go to rocBLAS; the solves do not: hipSOLVER's blocked
// getrf / getrs is ~300 tiny launches per call at this size (3 ms per tree level, all of it launch
// latency -- two thirds of `compute` with gadfly's 86-term solar kernel at N = 1e5).  This kernel does
// one batched solve in ONE launch: Gauss-Jordan with implicit partial pivoting (rows are never
// swapped) on [A | B], held in registers.
//   16 waves per system, every wave holds ALL rows (lane l: rows l, l + 64, l + 128) of its columns:
//   column j of A belongs to wave j mod 16 (slot j / 16), the right-hand sides are dealt the same
//   way; blockIdx.y cuts B into slices of 64 columns, each slice eliminates its own copy of A.
//   Step k: the pivot search is local to the wave that owns column k; it publishes the multipliers
//   f_i = a_ik / a_pk (f_p = 0), the pivot's lane and row set through LDS.  After ONE barrier every
//   wave updates its slots; the pivot row's entry of a column is a readlane of the wave's own
//   registers (an SGPR operand of the FMAs: no LDS traffic in the update).  The owner of column
//   k + 1 forms that column first and runs the next search before its update, so the search overlaps
//   the other waves' FMAs.
//   The k loop is cut into phases of 16 steps, unrolled at compile time: in phase q the slots below q
//   hold eliminated columns and are left alone, and the owner's slot index is a constant.  All
//   register updates are unconditional straight-line code: alternatives that merge (one per row set
//   of the pivot, say) cost a copy of the whole register array per step, so the row set only selects
//   which registers the readlanes read.
//   A, B and X move through an LDS tile of 32 rows so that global accesses are contiguous (lanes
//   along a row); lane = row gathers of a row-major matrix cost more than the elimination.
// Singular systems (maps of chunks after a failed pivot) produce garbage, never a fault.
constexpr int DS_WAVES = 16;                        // 16 NR right-hand sides per workgroup (NR: template parameter)
constexpr int DS_TROWS = 32, DS_LDT = 193;          // LDS tile: 32 rows, odd leading dimension

template <int RS>
struct DenseSolveShared {
    double tile[DS_TROWS * DS_LDT];
    double f[2][RS][64];                            // multipliers of step k, buffer k & 1
    double rowpinv[64 * RS];
    int pl[2], rs[2], rowk[64 * RS];
    int seen[64 * RS];                              // (log-determinant: cycle walk of the row -> pivot map)
};

// pivot search on column k (values C of this wave's rows) by its owner; publishes into buffer k & 1
template <int RS>
__device__ __forceinline__ void ds_search(DenseSolveShared<RS> &sh, const double (&C)[RS], const unsigned used,
                                          const int k, const int lane) {
    const int buf = k & 1;
    double bx = 0.0, bc = -2.0;
    int br = 0;
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        double a = fabs(C[r]);
        a = (a == a) ? a : 0.0;
        const double cnd = ((used >> r) & 1u) ? -1.0 : a;
        if (cnd > bc) { bc = cnd; bx = C[r]; br = r; }
    }
    const double vm = wave_max(bc);
    const unsigned long long hit = __ballot(bc == vm);
    const int pl = __builtin_amdgcn_readfirstlane(hit ? (int)__ffsll((long long)hit) - 1 : 0);
    const int rsel = __builtin_amdgcn_readlane(br, pl);
    const double pinv = fast_rcp(read_lane(bx, pl));
#pragma unroll
    for (int r = 0; r < RS; ++r)
        sh.f[buf][r][lane] = (lane == pl && r == rsel) ? 0.0 : C[r] * pinv;
    if (lane == 0) {
        sh.pl[buf] = pl; sh.rs[buf] = rsel;
        sh.rowk[64 * rsel + pl] = k; sh.rowpinv[64 * rsel + pl] = pinv;
    }
}

// entry (lane pl, row set RSEL) of slots [Q, NS) as wave-uniform values; the empty asm keeps the three
// instances apart (merged, the compiler selects among the register sets with per-step copies)
template <int RS, int NS, int RSEL, int Q>
__device__ __forceinline__ void ds_pivot_row(double (&pe)[NS], const double (&R)[RS][NS], const int pl) {
#pragma unroll
    for (int c = Q; c < NS; ++c) {
        int lo = __builtin_amdgcn_readlane(__double2loint(R[RSEL][c]), pl);
        int hi = __builtin_amdgcn_readlane(__double2hiint(R[RSEL][c]), pl);
        if (RSEL == 0) asm volatile("; row set 0" : "+s"(lo), "+s"(hi));
        else if (RSEL == 1) asm volatile("; row set 1" : "+s"(lo), "+s"(hi));
        else asm volatile("; row set 2" : "+s"(lo), "+s"(hi));
        pe[c] = __hiloint2double(hi, lo);
    }
}

// phase Q of the elimination: steps k = 16 Q ... 16 Q + 15 (< n), slots [Q, NA + NR)
template <int RS, int NA, int NR, int Q>
__device__ __forceinline__ void ds_phase(DenseSolveShared<RS> &sh, double (&R)[RS][NA + NR], unsigned &used,
                                         const int n, const int wave, const int lane) {
    constexpr int NS = NA + NR;
    const int kend = (n - DS_WAVES * Q < DS_WAVES) ? n - DS_WAVES * Q : DS_WAVES;
    for (int kk = 0; kk < kend; ++kk) {
        const int k = DS_WAVES * Q + kk, buf = k & 1;
        // the owner of the next column is the critical path of the step: it outranks the three waves
        // that share its SIMD until its search is published
        const bool owner = k + 1 < n && wave == ((k + 1) & (DS_WAVES - 1));
        if (owner) __builtin_amdgcn_s_setprio(3);
        wg_lds_barrier();
        double f[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) f[r] = sh.f[buf][r][lane];
        const int pl = __builtin_amdgcn_readfirstlane(sh.pl[buf]);
        const int rsel = __builtin_amdgcn_readfirstlane(sh.rs[buf]);
        if (lane == pl) used |= 1u << rsel;
        double pe[NS];                              // the pivot row's entries of this wave's slots
        if (RS == 1 || rsel == 0) ds_pivot_row<RS, NS, 0, Q>(pe, R, pl);
        else if (RS == 2 || rsel == 1) ds_pivot_row<RS, NS, (RS > 1 ? 1 : 0), Q>(pe, R, pl);
        else ds_pivot_row<RS, NS, (RS > 2 ? 2 : 0), Q>(pe, R, pl);
        if (owner) {
            // column k + 1 sits in slot Q, or in slot Q + 1 when this is the last step of the phase (both
            // formed, then selected: an index that depends on kk would make the register arrays dynamic)
            constexpr int Q1 = (Q + 1 < NA) ? Q + 1 : Q;
            const bool same = kk + 1 < DS_WAVES;
            double C[RS];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const double ca = fma(-f[r], pe[Q], R[r][Q]), cb = fma(-f[r], pe[Q1], R[r][Q1]);
                C[r] = same ? ca : cb;
            }
            ds_search<RS>(sh, C, used, k + 1, lane);
            __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int c = Q; c < NS; ++c)
#pragma unroll
            for (int r = 0; r < RS; ++r) R[r][c] = fma(-f[r], pe[c], R[r][c]);
    }
}

// NR right-hand-side slots per wave: 4 (64 per workgroup) for many systems; 1 when few systems leave the chip
// empty -- every workgroup eliminates its own copy of A, so narrower slices only shorten the step
template <int RS, int NA, int NR>
__global__ void __launch_bounds__(64 * DS_WAVES)
k_dense_solve(const int n, const int nrhs, const double *__restrict__ A_, double *__restrict__ B_,
              double *__restrict__ logdet_out) {
    constexpr int NS = NA + NR, NT = 64 * DS_WAVES, BC = DS_WAVES * NR;
    const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *Ab = A_ + (size_t)b * n * n;
    double *Bb = B_ + (size_t)b * n * nrhs + (size_t)sl * BC;
    const int ncol = (nrhs - sl * BC < BC) ? nrhs - sl * BC : BC;      // columns of this slice
    __shared__ __attribute__((aligned(16))) DenseSolveShared<RS> sh;
    double R[RS][NS];                               // [0, NA): columns wave + 16 slot of A; then B
    const int half = lane >> 5, l32 = lane & 31;
    static_for([&](auto tt) {                       // A: 32 rows at a time, as a flat contiguous copy
        constexpr int t = decltype(tt)::value, r = t >> 1;
        const int row0 = DS_TROWS * t;
        for (int e = tid; e < DS_TROWS * n; e += NT) {
            const int rr = e / n, cc = e - rr * n;
            sh.tile[rr * DS_LDT + cc] = (row0 + rr < n) ? Ab[(size_t)row0 * n + e] : 0.0;
        }
        __syncthreads();
        if (half == (t & 1)) {
#pragma unroll
            for (int la = 0; la < NA; ++la) {
                const int j = wave + DS_WAVES * la;
                R[r][la] = (j < n) ? sh.tile[l32 * DS_LDT + j] : 0.0;
            }
        }
        __syncthreads();
    }, std::make_integer_sequence<int, 2 * RS>{});
    static_for([&](auto tt) {                       // this slice of B
        constexpr int t = decltype(tt)::value, r = t >> 1;
        const int row0 = DS_TROWS * t;
        for (int e = tid; e < DS_TROWS * BC; e += NT) {
            const int rr = e / BC, cc = e - rr * BC;
            sh.tile[rr * DS_LDT + cc] = (row0 + rr < n && cc < ncol) ? Bb[(size_t)(row0 + rr) * nrhs + cc] : 0.0;
        }
        __syncthreads();
        if (half == (t & 1)) {
#pragma unroll
            for (int lr = 0; lr < NR; ++lr) R[r][NA + lr] = sh.tile[l32 * DS_LDT + wave + DS_WAVES * lr];
        }
        __syncthreads();
    }, std::make_integer_sequence<int, 2 * RS>{});
    unsigned used = 0;                              // bit r: row 64 r + lane already served as a pivot
#pragma unroll
    for (int r = 0; r < RS; ++r) if (64 * r + lane >= n) used |= 1u << r;
    for (int e = tid; e < 64 * RS; e += NT) { sh.rowk[e] = -1; sh.rowpinv[e] = 0.0; }
    __syncthreads();
    if (wave == 0) {
        double C[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) C[r] = R[r][0];
        ds_search<RS>(sh, C, used, 0, lane);
    }
    static_for([&](auto q) { ds_phase<RS, NA, NR, decltype(q)::value>(sh, R, used, n, wave, lane); },
               std::make_integer_sequence<int, NA>{});
    __syncthreads();
    // row 64 r + lane solved variable rowk[...]: X(rowk, :) = its B slots / pivot, through the tile
    int myk[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        myk[r] = sh.rowk[64 * r + lane];
        const double pinv = sh.rowpinv[64 * r + lane];
#pragma unroll
        for (int lr = 0; lr < NR; ++lr) R[r][NA + lr] *= pinv;
    }
    for (int row0 = 0; row0 < n; row0 += DS_TROWS) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int rel = myk[r] - row0;
            if (rel >= 0 && rel < DS_TROWS) {
#pragma unroll
                for (int lr = 0; lr < NR; ++lr) sh.tile[rel * DS_LDT + wave + DS_WAVES * lr] = R[r][NA + lr];
            }
        }
        __syncthreads();
        for (int e = tid; e < DS_TROWS * BC; e += NT) {
            const int rr = e / BC, cc = e - rr * BC;
            if (row0 + rr < n && cc < ncol) Bb[(size_t)(row0 + rr) * nrhs + cc] = sh.tile[rr * DS_LDT + cc];
        }
        __syncthreads();
    }
    // log det A = sum over the pivots of log |pivot|, when det A > 0 (sign of the pivots' product times the
    // parity of the row -> pivot-step permutation); NaN otherwise (det <= 0, a row never chosen, not finite)
    if (logdet_out != nullptr && sl == 0 && wave == 0) {
        double acc = 0.0;
        int neg = 0;
        bool bad = false;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int row = 64 * r + lane;
            if (row < n) {
                const double pinv = sh.rowpinv[row];
                const int kk = sh.rowk[row];
                bad = bad || kk < 0 || kk >= n || !(fabs(pinv) > 0.0) || !(fabs(pinv) < 1.79e308);
                acc -= log(fabs(pinv));
                neg += (pinv < 0.0) ? 1 : 0;
            }
            sh.seen[row] = 0;
        }
        acc = wave_sum(acc);
        neg = (int)__popcll(__ballot(neg & 1));     // parity of the number of negative pivots
        const bool anybad = __ballot(bad) != 0ull;
        wave_lds_fence();
        if (lane == 0) {
            int swaps = 0, steps = 0;
            bool ok = !anybad;
            for (int s0 = 0; ok && s0 < n; ++s0) {      // cycles of row -> rowk[row]; every walk is bounded by n
                if (sh.seen[s0]) continue;
                int j = s0, len = 0;
                do {
                    sh.seen[j] = 1;
                    j = sh.rowk[j];
                    ++len;
                    ++steps;
                } while (j != s0 && j >= 0 && j < n && !sh.seen[j] && steps <= n);
                ok = ok && j == s0;                 // (anything else: not a permutation)
                swaps += len - 1;
            }
            const bool positive = ok && (((neg + swaps) & 1) == 0) && (acc == acc);
            logdet_out[b] = positive ? acc : __longlong_as_double(0x7ff8000000000000LL);
        }
    }
}

// Combine of GF_MATMUL_LOWER (dot_tril): the chunk transitions are DIAGONAL (D_c), so the start states are
// a plain scan  F_{c+1} = loc_c + D_c o F_c  of 64 x R independent scalar sequences.  One workgroup of 16
// waves per (problem, state row, 16 right-hand sides): a wave's four 16-lane groups take one segment of the
// chunks each (64 segments per workgroup), lane & 15 = right-hand side (the states are stored [row][R]: 128-byte
// pieces of a row; with lane = state row every lane touched its own cache line).  The segments are scanned
// concurrently (zero start; eight chunks of loads in flight per lane: the loop is otherwise one dependent memory
// round trip per chunk), the 64 segment summaries are chained in LDS, and every group rescans its segment from
// its true start state, writing the start state of every chunk.  (With lane = right-hand side over all 64 and
// 16 segments there were 64 workgroups per problem: a quarter of the chip, 45 us at cfg5's size.)
constexpr int LCM_WAVES = 16, LCM_BATCH = 8, LCM_SEGS = 4 * LCM_WAVES;
__global__ void __launch_bounds__(64 * LCM_WAVES)
k_lincombine_mm(const int nch, const int R, const int rows, const double *__restrict__ Dch_,
                double *__restrict__ F_state) {
    const int pr = blockIdx.x, i = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rr = lane & 15, sg = 4 * w + (lane >> 4);        // right-hand side within the 16, segment
    const int rq = 16 * blockIdx.z + rr;
    const int r = (rq < R) ? rq : (R - 1);          // idle lanes shadow the last right-hand side (no stores)
    const bool rok = rq < R;
    const int seg = (nch + LCM_SEGS - 1) / LCM_SEGS;
    const int s0 = (sg * seg < nch) ? sg * seg : nch, s1 = (s0 + seg < nch) ? (s0 + seg) : nch;
    __shared__ double s_end[LCM_SEGS][16], s_dec[LCM_SEGS][16];
    auto F_at = [&](int s) { return F_state + ((size_t)pr * nch + s) * rows * R + (size_t)i * R + r; };
    auto D_at = [&](int s) { return Dch_[((size_t)pr * nch + s) * rows + i]; };
    // phase 1: segment end state from a zero start, and the segment's total decay
    double cur = 0.0, dec = 1.0;
    for (int s = s0; s < s1; s += LCM_BATCH) {
        double loc[LCM_BATCH], dd[LCM_BATCH];
#pragma unroll
        for (int j = 0; j < LCM_BATCH; ++j) {
            const bool ok = s + j < s1;
            loc[j] = ok ? *F_at(s + j) : 0.0;
            dd[j] = ok ? D_at(s + j) : 1.0;
        }
#pragma unroll
        for (int j = 0; j < LCM_BATCH; ++j) { cur = fma(dd[j], cur, loc[j]); dec *= dd[j]; }
    }
    s_end[sg][rr] = cur;
    s_dec[sg][rr] = dec;
    __syncthreads();
    // phase 2: this segment's true start state (chain of the summaries before it)
    double start = 0.0;
    for (int g2 = 0; g2 < sg; ++g2) start = fma(s_dec[g2][rr], start, s_end[g2][rr]);
    // phase 3: rescan, leaving the true start state of every chunk in its slot
    cur = start;
    for (int s = s0; s < s1; s += LCM_BATCH) {
        double loc[LCM_BATCH], dd[LCM_BATCH];
#pragma unroll
        for (int j = 0; j < LCM_BATCH; ++j) {
            const bool ok = s + j < s1;
            loc[j] = ok ? *F_at(s + j) : 0.0;
            dd[j] = ok ? D_at(s + j) : 1.0;
        }
#pragma unroll
        for (int j = 0; j < LCM_BATCH; ++j) {
            if (rok && s + j < s1) *F_at(s + j) = cur;
            cur = fma(dd[j], cur, loc[j]);
        }
    }
}

// D_c[j] = exp(-c_j * (sum of the reset spans inside chunk c))   (diagonal chunk transition of
// matmul_lower; the first row of every chunk is a reset row)
__global__ void __launch_bounds__(64)
k_chunk_decay(const int64_t N, const int64_t chunk_len, const int nch, const int W,
              const double *__restrict__ c_, const double *__restrict__ de_, double *__restrict__ D_out) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const int pr = b / nch, ch = b - pr * nch;
    const int64_t c0 = (int64_t)ch * chunk_len;
    const int64_t rows = (N - c0 < chunk_len) ? (N - c0) : chunk_len;
    const double *eg = de_ + (size_t)pr * N + c0;
    double acc = 0.0;
    for (int64_t n = lane; n < rows; n += 64) { const double v = eg[n]; if (v > 0.0) acc += v; }
    acc = wave_sum(acc);
    const double cj = (lane < W) ? c_[(size_t)pr * W + lane] : 0.0;
    D_out[(size_t)b * 64 + lane] = fm_exp(-cj * acc);
}

// ------------------------------------------------------------------------------------
// log-likelihood reductions: fixed-shape two-stage tree (deterministic)
//   stage 1: G blocks per problem -> partial (sum log d, sum z^2/d);  stage 2: one wave
// ------------------------------------------------------------------------------------
constexpr int RED_BLOCK = 256;
constexpr int RED_MAXG = 512;

__host__ __device__ inline int red_groups(int64_t N) {
    int64_t g = (N + 4095) / 4096;
    if (g < 1) g = 1;
    if (g > RED_MAXG) g = RED_MAXG;
    return (int)g;
}

constexpr int RED_NACC = 3;         // sum log d, sum z^2/d, min d

__global__ void __launch_bounds__(RED_BLOCK) k_reduce1(int64_t N, const double *d, const double *z,
                                                       double *work) {
    const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    const double *dp = d + (size_t)b * N;
    const double *zp = z ? z + (size_t)b * N : nullptr;
    double s1 = 0.0, s2 = 0.0, mn = INFINITY;
    for (int64_t n = (int64_t)g * RED_BLOCK + threadIdx.x; n < N; n += (int64_t)G * RED_BLOCK) {
        const double dn = dp[n];
        s1 += log(dn);
        mn = fmin(mn, dn);
        if (zp) { const double zn = zp[n]; s2 += zn * zn / dn; }
    }
    wave_sum2(s1, s2);
    mn = -wave_max(-mn);
    __shared__ double sh[RED_NACC][RED_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; sh[2][wave] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0, t3 = INFINITY;
        for (int w = 0; w < RED_BLOCK / 64; ++w) { t1 += sh[0][w]; t2 += sh[1][w]; t3 = fmin(t3, sh[2][w]); }
        double *o = work + ((size_t)b * G + g) * RED_NACC;
        o[0] = t1; o[1] = t2; o[2] = t3;
    }
}

// acc[b] = {sum log d, sum z^2/d, min d}; init != 0 overwrites, else accumulates (tile streaming,
// fixed order)
__global__ void __launch_bounds__(64) k_reduce2(int G, const double *work, double *acc, int init) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, mn = INFINITY;
    for (int g = lane; g < G; g += 64) {
        const double *w = work + ((size_t)b * G + g) * RED_NACC;
        s1 += w[0];
        s2 += w[1];
        mn = fmin(mn, w[2]);
    }
    wave_sum2(s1, s2);
    mn = -wave_max(-mn);
    if (lane == 0) {
        double *a = acc + (size_t)b * RED_NACC;
        if (!init) { s1 += a[0]; s2 += a[1]; mn = fmin(mn, a[2]); }
        a[0] = s1; a[1] = s2; a[2] = mn;
    }
}

__global__ void k_finish(int B, int64_t N, const double *acc, const int32_t *info,
                         double *out, double *logdet) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool bad = info && info[b] != 0;
    const double s1 = acc[(size_t)b * RED_NACC], s2 = acc[(size_t)b * RED_NACC + 1];
    if (logdet) logdet[b] = bad ? -INFINITY : s1;
    if (out) out[b] = bad ? -INFINITY : (-0.5 * (s1 + (double)N * 1.8378770664093453) - 0.5 * s2);
}

// ------------------------------------------------------------------------------------
// K2/K3/K4 with ONE right-hand side: F lane-resident (column j = c*64 + lane), one wave
// per problem, one DPP tree per row.
// ------------------------------------------------------------------------------------
struct SolveArgs {
    int64_t N;
    int W, ld, R, mode;
    const double *U, *Wm, *P, *scale, *Y;
    double *Z;
    // chunk mode of k_solve_vec (gf_solve_chunk): the grid is (problem, chunk); chunk c sweeps its rows
    // from the state in F_state slot (problem * nch + c) -- [ld] doubles, pending push folded, the decay
    // of the boundary left to the receiving chunk -- and leaves its end state there; store = 0: no Z
    int64_t chunk_len;
    int nch, store;
    double *F_state;
};

// MODE and SCALED are compile-time (the three sweeps x with / without the per-row scale): no
// branches on the chain.  CT = 3 covers the full solar kernel (W = 172, ld = 176).
template <int CT, int MODE, bool SCALED>
__global__ void __launch_bounds__(64) k_solve_vec(const SolveArgs A) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x / A.nch, ch = blockIdx.x - b * A.nch;     // (problem, chunk); nch = 1: whole series
    const int64_t N = A.N;
    const int ld = A.ld;
    const size_t pb = (size_t)b * N;
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t L = (N - c0 < A.chunk_len) ? (N - c0) : A.chunk_len;         // rows of this chunk
    constexpr bool up = (MODE == GF_SOLVE_UPPER);
    constexpr bool mm = (MODE == GF_MATMUL_LOWER);
    // "push" rows multiply the carried value into F; "pull" rows are dotted with F
    const double *__restrict__ push = (up ? A.U : A.Wm) + pb * ld;
    const double *__restrict__ pull = (up ? A.Wm : A.U) + pb * ld;
    const double *__restrict__ Pg = A.P + pb * ld;
    const double *Y = A.Y + pb;         // (Z may alias Y: rows are read ahead of, never behind, the writes)
    const double *__restrict__ sc = SCALED ? A.scale + pb : nullptr;
    double *Z = A.Z + pb;
    // The sweep is one dependent chain per row (two FMAs, a 64-lane reduction: a few hundred cycles),
    // so a row's operands -- three generator rows, y, the scale -- must already be in registers when
    // the chain reaches it.  They are fetched DEPTH rows ahead into a register ring, by plain
    // unconditional vector loads that nothing touches until the row is processed: pad lanes read a
    // clamped column and are switched off by a 0/1 factor on P, the per-row scalars go through the
    // same (in-order) vector-load queue via an opaque zero lane offset instead of scalar loads the
    // wave would have to wait for on the spot.  Without the ring every row waited for its own HBM
    // round trip: 1.6 us per row.
    constexpr int DEPTH = 8;
    const int vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);      // 0 in every lane, not known to be uniform
    int col[CT];
    double okf[CT], F[CT];
    double *__restrict__ Fs = A.F_state ? A.F_state + (size_t)blockIdx.x * ld : nullptr;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int j = c * 64 + lane;
        col[c] = j < ld ? j : ld - 1;
        okf[c] = j < ld ? 1.0 : 0.0;
        F[c] = Fs ? Fs[col[c]] * okf[c] : 0.0;
    }
    auto row_of = [&](const int64_t s) { return up ? (c0 + L - 1 - s) : (c0 + s); };
    auto scaled = [&](const double yv, const double sv) {
        if constexpr (SCALED) return mm ? yv * sqrt(sv) : yv / sv;
        else return yv;
    };

    // the first row of the sweep carries nothing itself (carry = 0): the state handed over already holds
    // the pending push of the row before it (F = 0 at the very start).  The decay of a chunk boundary is
    // applied by the chunk that OWNS the boundary row -- its first row: ascending sweeps apply it on entry
    // (the first processed row), the descending sweep on exit, so that the homogeneous part of a chunk is
    // exactly its closed-loop transition Phi_c (Phi_c^T going down), as in k_lin1.
    double carry = 0.0;
    double rp[DEPTH][CT], ra[DEPTH][CT], rb[DEPTH][CT], ry[DEPTH], rs[DEPTH];
    auto clampN = [&](const int64_t n) { return n < 0 ? (int64_t)0 : (n > N - 1 ? N - 1 : n); };
    auto fetch = [&](const int slot, int64_t s) {
        s = s < L ? s : L - 1;                      // past the end: re-read the last row, never used
        const int64_t n = row_of(s);
        const int64_t prev = clampN(up ? (n + 1) : (n - 1));    // (rows outside the series: multiplied by 0)
        const int64_t prow = clampN(up ? (n + 1) : n);
        ry[slot] = Y[n + vz];
        rs[slot] = SCALED ? sc[n + vz] : 1.0;
        const bool entry = up && Fs != nullptr && s == 0;      // descending chunk entry: no decay here
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            rp[slot][c] = entry ? 1.0 : Pg[(size_t)prow * ld + col[c]];
            ra[slot][c] = push[(size_t)prev * ld + col[c]];
            rb[slot][c] = pull[(size_t)n * ld + col[c]];
        }
    };
    auto process = [&](const int64_t s, const double (&p)[CT], const double (&a)[CT],
                       const double (&q)[CT], const double yv, const double sv) {
        const int64_t n = row_of(s);
        const double yn = scaled(yv, sv);
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            F[c] = (p[c] * okf[c]) * fma(a[c], carry, F[c]);
            dot = fma(q[c], F[c], dot);
        }
        dot = wave_sum(dot);
        const double zn = mm ? (yn + dot) : (yn - dot);
        if (A.store && lane == 0) Z[n] = zn;
        carry = mm ? yn : zn;
    };
    int64_t s0 = 0;
    if (L >= DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) fetch(k, k);
        for (; s0 + DEPTH <= L; s0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                process(s0 + k, rp[k], ra[k], rb[k], ry[k], rs[k]);
                fetch(k, s0 + k + DEPTH);
            }
        }
    }
    for (; s0 < L; ++s0) {                          // fewer than DEPTH rows left
        fetch(0, s0);
        process(s0, rp[0], ra[0], rb[0], ry[0], rs[0]);
    }
    if (Fs) {                                       // end state: pending push folded, no decay
        const int64_t last = row_of(L - 1);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const double a = push[(size_t)last * ld + col[c]];
            double v = fma(a, carry, F[c]);
            if (up) v *= Pg[(size_t)last * ld + col[c]];       // cross the chunk's first-row boundary
            if (c * 64 + lane < ld) Fs[c * 64 + lane] = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// K2/K3/K4 with R right-hand sides: lane r of a wave owns column r of Y/Z and the W-vector
// F[:, r] in VGPRs; the generator rows are wave-uniform operands (prefetched by vector loads,
// broadcast through LDS).  Widths above 64 split j over NWV waves (one LDS exchange/row).
// ------------------------------------------------------------------------------------
template <int JB, int NWV>
__global__ void __launch_bounds__(64 * NWV) k_solve_rhs(const SolveArgs A) {
    static_assert(JB % 2 == 0 && JB <= 64, "column slice: even, at most one wave wide");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (problem, chunk): nch = 1 is the whole series; chunk mode (F_state != nullptr) as in k_solve_vec --
    // start state in, end state out (pending push folded; ascending sweeps apply a boundary's decay on
    // entry, the descending one on exit), Z only when A.store
    const int b = blockIdx.y / A.nch, ch = blockIdx.y - b * A.nch;
    const int r = blockIdx.x * 64 + lane;
    const int R = A.R;
    const bool rok = r < R;
    const int rc = rok ? r : (R - 1);   // idle lanes re-read the last right-hand side, never store
    const int64_t N = A.N;
    const int ld = A.ld;
    const size_t pb = (size_t)b * N;
    const int64_t c0 = (int64_t)ch * A.chunk_len;
    const int64_t L = (N - c0 < A.chunk_len) ? (N - c0) : A.chunk_len;
    const bool up = (A.mode == GF_SOLVE_UPPER);
    const bool mm = (A.mode == GF_MATMUL_LOWER);
    const double *__restrict__ push = (up ? A.U : A.Wm) + pb * ld;
    const double *__restrict__ pull = (up ? A.Wm : A.U) + pb * ld;
    const double *__restrict__ Pg = A.P + pb * ld;
    const double *Y = A.Y + pb * R;     // (Z may alias Y: rows are read ahead of, never behind, the writes)
    const double *__restrict__ sc = A.scale ? A.scale + pb : nullptr;
    double *Z = A.Z + pb * R;
    double *__restrict__ Fs = A.F_state ? A.F_state + (size_t)blockIdx.y * ld * R : nullptr;     // [ld][R]
    const int j0 = wave * JB;
    int jn = ld - j0;                    // valid columns in this wave's slice
    if (jn > JB) jn = JB;
    if (jn < 0) jn = 0;
    // The generator rows of this wave's slice are wave-uniform operands of the per-lane recurrences.
    // Lane l fetches column j0 + l of the three rows DEPTH rows ahead (plain vector loads into a
    // register ring), the row being processed is staged in LDS and read back as broadcasts.  (As
    // scalar loads issued when the row needed them, they cost a memory round trip per row: 5.6 us
    // per row at W = 172.)
    constexpr int DEPTH = 8;
    const bool lok = lane < jn;
    const int jl = j0 + (lok ? lane : (jn > 0 ? jn - 1 : 0));
    const int jc = jl < ld ? jl : ld - 1;
    const int vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);

    __shared__ __attribute__((aligned(16))) double s_row[NWV][3][64];
    __shared__ double s_dot[2][NWV][64];
    double *sp = s_row[wave][0], *sa = s_row[wave][1], *sb = s_row[wave][2];

    double F[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) F[j] = (Fs && j < jn) ? Fs[(size_t)(j0 + j) * R + rc] : 0.0;
    auto row_of = [&](const int64_t s) { return up ? (c0 + L - 1 - s) : (c0 + s); };
    auto scaled = [&](const double yv, const double sv) {
        return sc ? (mm ? yv * sqrt(sv) : yv / sv) : yv;
    };
    // the first row of the sweep carries nothing itself: the state handed over already holds the pending
    // push of the row before it (zero at the very start)
    double carry = 0.0;
    double rp[DEPTH], ra[DEPTH], rb[DEPTH], ry[DEPTH], rs[DEPTH];
    auto clampN = [&](const int64_t n) { return n < 0 ? (int64_t)0 : (n > N - 1 ? N - 1 : n); };
    auto fetch = [&](const int slot, int64_t s) {
        s = s < L ? s : L - 1;                      // past the end: re-read the last row, never used
        const int64_t n = row_of(s);
        const int64_t prev = clampN(up ? (n + 1) : (n - 1));    // (rows outside the series: multiplied by 0)
        const int64_t prow = clampN(up ? (n + 1) : n);
        ry[slot] = Y[(size_t)n * R + rc];
        rs[slot] = sc ? sc[n + vz] : 1.0;
        const bool entry = up && Fs != nullptr && s == 0;      // descending chunk entry: no decay here
        rp[slot] = entry ? 1.0 : Pg[(size_t)prow * ld + jc];
        ra[slot] = push[(size_t)prev * ld + jc];
        rb[slot] = pull[(size_t)n * ld + jc];
    };
    auto process = [&](const int64_t s, const double pv, const double av, const double bv,
                       const double yv, const double sv) {
        const int64_t n = row_of(s);
        const double yn = scaled(yv, sv);
        wave_lds_fence();                           // the previous row's broadcasts are done
        sp[lane] = lok ? pv : 0.0;                  // columns past the slice: all zero, F stays 0
        sa[lane] = lok ? av : 0.0;
        sb[lane] = lok ? bv : 0.0;
        wave_lds_fence();
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < JB; j += 2) {
            const double2 p2 = *(const double2 *)(sp + j), a2 = *(const double2 *)(sa + j);
            const double2 b2 = *(const double2 *)(sb + j);
            F[j] = p2.x * fma(a2.x, carry, F[j]);
            F[j + 1] = p2.y * fma(a2.y, carry, F[j + 1]);
            dot = fma(b2.x, F[j], dot);
            dot = fma(b2.y, F[j + 1], dot);
        }
        if constexpr (NWV > 1) {
            const int buf = (int)(s & 1);
            s_dot[buf][wave][lane] = dot;
            wg_lds_barrier();
            dot = s_dot[buf][0][lane];
#pragma unroll
            for (int w2 = 1; w2 < NWV; ++w2) dot += s_dot[buf][w2][lane];
        }
        const double zn = mm ? (yn + dot) : (yn - dot);
        if (A.store && rok && wave == 0) Z[(size_t)n * R + r] = zn;
        carry = mm ? yn : zn;
    };
    int64_t s0 = 0;
    if (L >= DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) fetch(k, k);
        for (; s0 + DEPTH <= L; s0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                process(s0 + k, rp[k], ra[k], rb[k], ry[k], rs[k]);
                fetch(k, s0 + k + DEPTH);
            }
        }
    }
    for (; s0 < L; ++s0) {                          // fewer than DEPTH rows left
        fetch(0, s0);
        process(s0, rp[0], ra[0], rb[0], ry[0], rs[0]);
    }
    if (Fs) {                                       // end state: pending push folded, no decay
        const int64_t last = row_of(L - 1);
        wave_lds_fence();
        sa[lane] = lok ? push[(size_t)last * ld + jc] : 0.0;
        sp[lane] = (lok && up) ? Pg[(size_t)last * ld + jc] : 1.0;     // cross the chunk's first-row boundary
        wave_lds_fence();
        if (rok) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (j < jn) Fs[(size_t)(j0 + j) * R + r] = sp[j] * fma(sa[j], carry, F[j]);
        }
    }
}

// ------------------------------------------------------------------------------------
// K5: conditional mean at new times.  One wave per (problem, direction);
// F lane-resident; consecutive observed rows advance with the stored propagator rows P2,
// only the hop from the last observed row to the query time needs an exp.
//   dir 0 (lower): ascending;  dir 1 (upper): descending.  Partial results go to work[dir].
// ------------------------------------------------------------------------------------
struct GmmArgs {
    int64_t M, N;
    int W, ld;
    const double *c;
    const double *t1; int64_t t1_bs;
    const double *U1, *V1;
    const double *t2; int64_t t2_bs;
    const double *U2, *V2, *P2, *alpha;
    const int64_t *qidx;            // [B][M]: number of observed rows with t2 <= t1[m]
    double *work;
};

// Rows are streamed in unrolled blocks of GB (all loads of a block are issued before its
// FMAs: one dependent FMA per row instead of one memory round trip per row); a query is
// emitted when the sweep passes its row:  lower: after row qidx-1;  upper: after row qidx.
template <int CT>
__global__ void __launch_bounds__(64) k_gmm(const GmmArgs A) {
    constexpr int GB = 8;
    const int lane = threadIdx.x, b = blockIdx.x, dir = blockIdx.y;
    const int64_t M = A.M, N = A.N;
    const int ld = A.ld;
    const double *t1 = A.t1 + (size_t)b * A.t1_bs;
    const double *t2 = A.t2 + (size_t)b * A.t2_bs;
    const double *__restrict__ Q1 = (dir == 0 ? A.U1 : A.V1) + (size_t)b * M * ld;   // query-side rows
    const double *__restrict__ G2 = (dir == 0 ? A.V2 : A.U2) + (size_t)b * N * ld;   // data-side rows
    const double *__restrict__ P2 = A.P2 + (size_t)b * N * ld;
    const double *__restrict__ al = A.alpha + (size_t)b * N;
    const int64_t *__restrict__ qi = A.qidx + (size_t)b * M;
    double *out = A.work + ((size_t)b * 2 + dir) * M;
    bool colok[CT];
    int col[CT];
    double F[CT], cj[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int j = c * 64 + lane;
        colok[c] = j < A.W;
        col[c] = colok[c] ? j : 0;      // pad lanes re-read column 0 and are masked in registers
        F[c] = 0.0;
        cj[c] = colok[c] ? A.c[(size_t)b * A.W + j] : 0.0;
    }
    auto emit = [&](int64_t m, double tdata) {
        const double dt = (dir == 0) ? (tdata - t1[m]) : (t1[m] - tdata);   // <= 0
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (colok[c]) acc = fma(Q1[(size_t)m * ld + col[c]] * exp(cj[c] * dt), F[c], acc);
        acc = wave_sum(acc);
        if (lane == 0) out[m] = acc;
    };
    if (dir == 0) {
        int64_t m = 0;
        while (m < M && qi[m] == 0) { if (lane == 0) out[m] = 0.0; ++m; }
        int64_t next = (m < M) ? qi[m] : (N + 1);          // emit after row next-1
        int64_t n = 0;
        for (; n + GB <= N; n += GB) {
            double pr[GB][CT], gr[GB][CT], an[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                an[k] = al[n + k];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const size_t o = (size_t)(n + k) * ld + col[c];
                    pr[k][c] = P2[o];                       // exp(c (t2[n-1] - t2[n])); row 0: 1
                    gr[k][c] = G2[o];
                }
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
#pragma unroll
                for (int c = 0; c < CT; ++c) F[c] = colok[c] ? fma(pr[k][c], F[c], gr[k][c] * an[k]) : 0.0;
                while (next == n + k + 1) {                 // wave-uniform, rare
                    emit(m, t2[n + k]);
                    ++m;
                    next = (m < M) ? qi[m] : (N + 1);
                }
            }
        }
        for (; n < N; ++n) {
            const double a1 = al[n];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const size_t o = (size_t)n * ld + col[c];
                F[c] = colok[c] ? fma(P2[o], F[c], G2[o] * a1) : 0.0;
            }
            while (next == n + 1) {
                emit(m, t2[n]);
                ++m;
                next = (m < M) ? qi[m] : (N + 1);
            }
        }
    } else {
        int64_t m = M - 1;
        while (m >= 0 && qi[m] == N) { if (lane == 0) out[m] = 0.0; --m; }
        int64_t next = (m >= 0) ? qi[m] : -1;               // emit after row next (descending)
        int64_t n = N - 1;
        for (; n - GB + 1 >= 0; n -= GB) {
            double pr[GB][CT], gr[GB][CT], an[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int64_t r = n - k;
                an[k] = al[r];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const size_t o = (size_t)r * ld + col[c];
                    // exp(c (t2[r] - t2[r+1])) is propagator row r+1 (the last row decays nothing)
                    pr[k][c] = (r + 1 < N) ? P2[o + ld] : 1.0;
                    gr[k][c] = G2[o];
                }
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
#pragma unroll
                for (int c = 0; c < CT; ++c) F[c] = colok[c] ? fma(pr[k][c], F[c], gr[k][c] * an[k]) : 0.0;
                while (next == n - k) {
                    emit(m, t2[n - k]);
                    --m;
                    next = (m >= 0) ? qi[m] : -1;
                }
            }
        }
        for (; n >= 0; --n) {
            const double a1 = al[n];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const size_t o = (size_t)n * ld + col[c];
                const double pj = (n + 1 < N) ? P2[o + ld] : 1.0;
                F[c] = colok[c] ? fma(pj, F[c], G2[o] * a1) : 0.0;
            }
            while (next == n) {
                emit(m, t2[n]);
                --m;
                next = (m >= 0) ? qi[m] : -1;
            }
        }
    }
}

// ---- chunk-parallel form (long series): the recurrence F <- p o F + g alpha is linear with a
// DIAGONAL transition, so chunks of rows are swept concurrently:
//   k_gmm_chunk<CT, 0>  local pass from F = 0: end state F_loc and the chunk's decay D = prod p
//   k_gmm_scan          F_start of every chunk (sequential over chunks, 64-wide FMAs)
//   k_gmm_chunk<CT, 1>  chunks that contain queries are swept again from F_start and emit
// Row ranges: dir 0 sweeps rows a..b ascending with p_n = P2[n]; dir 1 sweeps b..a descending
// with p_n = P2[n+1] (1 for the last row of the series).  A query is emitted after row
// qidx-1 (dir 0) / row qidx (dir 1), as in k_gmm.
struct GmmChunk {
    int64_t chunk_len;
    int nch;
    double *Floc, *Dloc, *Fstart;   // [B][2][nch][CT*64]
};

__device__ __forceinline__ int64_t gmm_lower_bound(const int64_t *q, int64_t M, int64_t v) {
    int64_t lo = 0, hi = M;         // first m with q[m] >= v
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (q[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

template <int CT, int PHASE>
__global__ void __launch_bounds__(64) k_gmm_chunk(const GmmArgs A, const GmmChunk C) {
    const int lane = threadIdx.x, b = blockIdx.x, dir = blockIdx.y, ch = blockIdx.z;
    const int64_t M = A.M, N = A.N;
    const int ld = A.ld;
    const int64_t a = (int64_t)ch * C.chunk_len;
    const int64_t e = (a + C.chunk_len < N) ? (a + C.chunk_len) : N;      // rows [a, e)
    const double *t1 = A.t1 + (size_t)b * A.t1_bs;
    const double *t2 = A.t2 + (size_t)b * A.t2_bs;
    const double *__restrict__ Q1 = (dir == 0 ? A.U1 : A.V1) + (size_t)b * M * ld;
    const double *__restrict__ G2 = (dir == 0 ? A.V2 : A.U2) + (size_t)b * N * ld;
    const double *__restrict__ P2 = A.P2 + (size_t)b * N * ld;
    const double *__restrict__ al = A.alpha + (size_t)b * N;
    const int64_t *__restrict__ qi = A.qidx + (size_t)b * M;
    double *out = A.work + ((size_t)b * 2 + dir) * M;
    const size_t slot = (((size_t)b * 2 + dir) * C.nch + ch) * (CT * 64);
    bool colok[CT];
    int col[CT];
    double F[CT], D[CT], cj[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int j = c * 64 + lane;
        colok[c] = j < A.W;
        col[c] = colok[c] ? j : 0;
        F[c] = (PHASE == 1) ? C.Fstart[slot + c * 64 + lane] : 0.0;
        D[c] = 1.0;
        cj[c] = colok[c] ? A.c[(size_t)b * A.W + j] : 0.0;
    }
    // queries of this chunk: [m_lo, m_hi)
    int64_t m_lo = 0, m_hi = 0;
    if (PHASE == 1) {
        if (dir == 0) { m_lo = gmm_lower_bound(qi, M, a + 1); m_hi = gmm_lower_bound(qi, M, e + 1); }
        else          { m_lo = gmm_lower_bound(qi, M, a);     m_hi = gmm_lower_bound(qi, M, e); }
        // queries before the first / after the last observed row get no contribution
        if (dir == 0 && ch == 0) for (int64_t m = lane; m < m_lo; m += 64) out[m] = 0.0;
        if (dir == 1 && ch == C.nch - 1) for (int64_t m = m_hi + lane; m < M; m += 64) out[m] = 0.0;
        if (m_lo == m_hi) return;
    }
    auto emit = [&](int64_t m, double tdata) {
        const double dt = (dir == 0) ? (tdata - t1[m]) : (t1[m] - tdata);   // <= 0
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (colok[c]) acc = fma(Q1[(size_t)m * ld + col[c]] * exp(cj[c] * dt), F[c], acc);
        acc = wave_sum(acc);
        if (lane == 0) out[m] = acc;
    };
    constexpr int GB = 8;
    if (dir == 0) {
        int64_t m = m_lo;
        int64_t next = (PHASE == 1) ? qi[m] : (N + 2);                      // emit after row next-1
        const int64_t last = (PHASE == 1) ? qi[m_hi - 1] : e;               // rows [a, last)
        for (int64_t n = a; n < last; n += GB) {
            double pr[GB][CT], gr[GB][CT], an[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int64_t r = (n + k < last) ? (n + k) : (last - 1);
                an[k] = al[r];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const size_t o = (size_t)r * ld + col[c];
                    pr[k][c] = P2[o];
                    gr[k][c] = G2[o];
                }
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                if (n + k < last) {                                          // wave-uniform
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        F[c] = colok[c] ? fma(pr[k][c], F[c], gr[k][c] * an[k]) : 0.0;
                        if (PHASE == 0) D[c] *= pr[k][c];
                    }
                    if (PHASE == 1) {
                        while (m < m_hi && next == n + k + 1) {
                            emit(m, t2[n + k]);
                            ++m;
                            next = (m < m_hi) ? qi[m] : (N + 2);
                        }
                    }
                }
            }
        }
    } else {
        int64_t m = m_hi - 1;
        int64_t next = (PHASE == 1) ? qi[m] : -2;                           // emit after row next
        const int64_t first = (PHASE == 1) ? qi[m_lo] : a;                  // rows (e-1) .. first
        for (int64_t n = e - 1; n >= first; n -= GB) {
            double pr[GB][CT], gr[GB][CT], an[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int64_t r = (n - k >= first) ? (n - k) : first;
                an[k] = al[r];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const size_t o = (size_t)r * ld + col[c];
                    pr[k][c] = (r + 1 < N) ? P2[o + ld] : 1.0;
                    gr[k][c] = G2[o];
                }
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                if (n - k >= first) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        F[c] = colok[c] ? fma(pr[k][c], F[c], gr[k][c] * an[k]) : 0.0;
                        if (PHASE == 0) D[c] *= pr[k][c];
                    }
                    if (PHASE == 1) {
                        while (m >= m_lo && next == n - k) {
                            emit(m, t2[n - k]);
                            --m;
                            next = (m >= m_lo) ? qi[m] : -2;
                        }
                    }
                }
            }
        }
    }
    if (PHASE == 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            C.Floc[slot + c * 64 + lane] = F[c];
            C.Dloc[slot + c * 64 + lane] = colok[c] ? D[c] : 0.0;
        }
    }
}

// F_start of every chunk.  dir 0: F_start[0] = 0, F_start[c+1] = D_c F_start[c] + F_loc[c];
// dir 1 runs from the last chunk down.
__global__ void __launch_bounds__(256) k_gmm_scan(const int CTW, const GmmChunk C) {
    const int b = blockIdx.x, dir = blockIdx.y;
    const size_t base = ((size_t)b * 2 + dir) * C.nch * CTW;
    // sixteen chunks' (D, F_loc) are fetched before their sixteen dependent FMAs: one memory round trip per
    // chunk made the scan 0.28 ms for the 1024 chunks of a 1e6-row series (a tenth of predict at new times)
    constexpr int GB = 16;
    for (int j = threadIdx.x; j < CTW; j += 256) {
        double F = 0.0;
        for (int s0 = 0; s0 < C.nch; s0 += GB) {
            double d[GB], f[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                int sidx = s0 + k;
                sidx = (sidx < C.nch) ? sidx : C.nch - 1;
                const int c = (dir == 0) ? sidx : C.nch - 1 - sidx;
                const size_t o = base + (size_t)c * CTW + j;
                d[k] = C.Dloc[o];
                f[k] = C.Floc[o];
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int sidx = s0 + k;
                if (sidx < C.nch) {
                    const int c = (dir == 0) ? sidx : C.nch - 1 - sidx;
                    C.Fstart[base + (size_t)c * CTW + j] = F;
                    F = fma(d[k], F, f[k]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_cross: cross-covariance block K(t_n, t*_r) of the celerite kernel, written in the
// [B][N][R] layout the multi-RHS sweeps take (conditional variance / covariance:
// celerite2's ConditionalDistribution builds the same dense block on the host).
//   K = sum_r a_r e^{-c_r tau} + sum_c (a_c cos d_c tau + b_c sin d_c tau) e^{-c_c tau},
//   tau = |t_n - t*_r|.
// A term is Re[(a - i b) w(tau)] with w(tau) = e^{(-c + i d) tau}, and w(tau_b + delta) = w(tau_b) w(delta): one
// workgroup takes 256 rows, whose times span [tmin, tmax].  A query at or beyond tmax has tau = (t* - tmax) +
// (tmax - t_n), one at or before tmin tau = (tmin - t*) + (t_n - tmin) -- both parts non-negative, both factors
// decaying.  So per workgroup and term: (a - i b) w(t* - tmax) or w(tmin - t*) for every query (a table in LDS), and
// per row w(tmax - t_n), w(t_n - tmin); an entry is then TWO FMAs per term instead of a sine, a cosine and an
// exponential (1.9e9 of each per 64 queries at N = 1e6, J = 30: 5.3 ms, half of predict(return_var=True)).  Queries
// inside the workgroup's span are evaluated entry by entry, as before (one workgroup per query).  Lane = row, RT
// accumulators per lane; the row's RT entries leave as one contiguous run.
// The exposure-integrated kernel differs from its coefficient form for tau < delta; the caller
// patches those (at most a few per query) entries.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void cross_w(const double cc, const double dd, const double x, double &re, double &im) {
    const double e = fm_exp(-cc * x);               // x >= 0
    if (dd == 0.0) { re = e; im = 0.0; return; }    // (real term)
    double si, cs;
    const double ph = dd * x;
    if (fabs(ph) < FM_SINCOS_RANGE) fm_sincos(ph, &si, &cs); else sincos(ph, &si, &cs);
    re = e * cs;
    im = e * si;
}

template <int RT>
__global__ void __launch_bounds__(256)
k_cross(const int64_t N, const int R, const int Jr, const int Jc,
        const double *__restrict__ ar_, const double *__restrict__ cr_,
        const double *__restrict__ ac_, const double *__restrict__ bc_,
        const double *__restrict__ cc_, const double *__restrict__ dc_,
        const double *__restrict__ t_, const int64_t t_bs,
        const double *__restrict__ ts_, const int64_t ts_bs, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int J = Jr + Jc, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *co = lds;                               // [J][4]: a, b, c, d (real terms: b = d = 0)
    double2 *AE = reinterpret_cast<double2 *>(co + 4 * J);      // [RT][J]: (a - i b) w(tau_b), zero for queries inside
    double *tq = reinterpret_cast<double *>(AE + (size_t)RT * J);  // [RT]
    __shared__ double s_mm[2][4];
    __shared__ unsigned long long s_side[3];        // bit r: query at / beyond tmax, at / before tmin, inside
    for (int i = tid; i < J; i += 256) {
        double *q = co + 4 * i;
        if (i < Jr) { q[0] = ar_[(size_t)b * Jr + i]; q[1] = 0.0; q[2] = cr_[(size_t)b * Jr + i]; q[3] = 0.0; }
        else {
            const size_t o = (size_t)b * Jc + (i - Jr);
            q[0] = ac_[o]; q[1] = bc_[o]; q[2] = cc_[o]; q[3] = dc_[o];
        }
    }
    const int64_t n = (int64_t)blockIdx.x * 256 + tid;
    const bool row = n < N;
    const double tn = t_[(size_t)b * t_bs + (row ? n : N - 1)];
    {
        const double hi = wave_max(tn), lo = -wave_max(-tn);
        if (lane == 0) { s_mm[0][wave] = hi; s_mm[1][wave] = lo; }
    }
    __syncthreads();
    const double tmax = fmax(fmax(s_mm[0][0], s_mm[0][1]), fmax(s_mm[0][2], s_mm[0][3]));
    const double tmin = fmin(fmin(s_mm[1][0], s_mm[1][1]), fmin(s_mm[1][2], s_mm[1][3]));
    const double dL = tmax - tn, dR = tn - tmin;   // >= 0
    double *og = out + ((size_t)b * N + (row ? n : 0)) * R;
    for (int r0 = 0; r0 < R; r0 += RT) {
        const int rt = (R - r0 < RT) ? R - r0 : RT;
        __syncthreads();
        if (tid < 64) {
            const double q = (tid < rt) ? ts_[(size_t)b * ts_bs + r0 + tid] : 0.0;
            if (tid < RT) tq[tid] = q;
            const bool L = tid < rt && q >= tmax, Rr = tid < rt && !L && q <= tmin, in = tid < rt && !L && !Rr;
            const unsigned long long mL = __ballot(L), mR = __ballot(Rr), mI = __ballot(in);
            if (tid == 0) { s_side[0] = mL; s_side[1] = mR; s_side[2] = mI; }
        }
        __syncthreads();
        const unsigned long long mL = s_side[0], mR = s_side[1], mI = s_side[2];
        for (int e = tid; e < rt * J; e += 256) {
            const int r = e / J, i = e - r * J;
            const double *q = co + 4 * i;
            double re = 0.0, im = 0.0;
            if (!((mI >> r) & 1ull)) {
                const double taub = ((mL >> r) & 1ull) ? tq[r] - tmax : tmin - tq[r];
                double wr, wi;
                cross_w(q[2], q[3], taub, wr, wi);
                re = fma(q[0], wr, q[1] * wi);      // (a - i b)(wr + i wi)
                im = fma(q[0], wi, -q[1] * wr);
            }
            AE[(size_t)r * J + i] = double2{re, im};
        }
        __syncthreads();
        double acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = 0.0;
        // (the masks as scalars: the tests below become scalar branches, not exec-mask switches)
        const unsigned mLlo = __builtin_amdgcn_readfirstlane((unsigned)mL), mLhi = __builtin_amdgcn_readfirstlane((unsigned)(mL >> 32));
        const unsigned mRlo = __builtin_amdgcn_readfirstlane((unsigned)mR), mRhi = __builtin_amdgcn_readfirstlane((unsigned)(mR >> 32));
        for (int i = 0; i < J; ++i) {
            const double *q = co + 4 * i;
            double lr = 0.0, li = 0.0, rr = 0.0, ri = 0.0;
            if (mL) cross_w(q[2], q[3], dL, lr, li);            // (workgroup-uniform)
            if (mR) cross_w(q[2], q[3], dR, rr, ri);
            const double2 *ae = AE + i;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const bool isL = ((r < 32 ? mLlo : mLhi) >> (r & 31)) & 1u, isR = ((r < 32 ? mRlo : mRhi) >> (r & 31)) & 1u;
                if (isL) { const double2 v = ae[(size_t)r * J]; acc[r] = fma(v.x, lr, fma(-v.y, li, acc[r])); }
                else if (isR) { const double2 v = ae[(size_t)r * J]; acc[r] = fma(v.x, rr, fma(-v.y, ri, acc[r])); }
            }
        }
        if (mI) {                                   // queries inside this workgroup's span: entry by entry
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if ((mI >> r) & 1ull) {
                    const double tau = fabs(tn - tq[r]);
                    double k = 0.0;
                    for (int i = 0; i < J; ++i) {
                        const double *q = co + 4 * i;
                        double wr, wi;
                        cross_w(q[2], q[3], tau, wr, wi);
                        k = fma(q[0], wr, fma(q[1], wi, k));
                    }
                    acc[r] = k;
                }
            }
        }
        if (row) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (r < rt) og[r0 + r] = acc[r];
        }
    }
}

// mu[b][m] = work[b][0][m] + work[b][1][m]
__global__ void k_add2(int64_t M, const double *work, double *mu) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (m < M) mu[b * M + m] = work[(b * 2) * M + m] + work[(b * 2 + 1) * M + m];
}

// ------------------------------------------------------------------------------------
// Power spectra of light curves / prior draws and their binning: the step
// after `sample` in the reference's own hot-path test (gadfly/tests/test_core.py:29-34).  The
// transform itself is hipFFT's (a plain library FFT); these two kernels are what the reference
// does around it in numpy / scipy.binned_statistic.
// ------------------------------------------------------------------------------------
// power[r][k] = |X[r][first + k]|^2 * norm      (PowerSpectrum._fft, psd.py:566-587)
// X interleaved complex [R][M]; re*re + im*im without contraction, as numpy forms it.
__global__ void __launch_bounds__(256)
k_psd_power(const int64_t M, const int64_t Mout, const int64_t first, const double norm,
            const double2 *__restrict__ spec, double *__restrict__ power) {
#pragma clang fp contract(off)      // plain operators under this pragma: HIP's __dmul_rn / __dadd_rn wrappers still contract
    const size_t r = blockIdx.y;
    const double2 *X = spec + r * (size_t)M + first;
    double *P = power + r * (size_t)Mout;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < Mout; k += (int64_t)gridDim.x * 256) {
        const double2 v = X[k];
        P[k] = (v.x * v.x + v.y * v.y) * norm;
    }
}

// sum over the 256 threads of a workgroup, result in every thread (fixed shape: deterministic)
__device__ __forceinline__ double wg_sum256(double v, double *red) {
    v = wave_sum(v);
    __syncthreads();                                // red may still be read from the last call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup per (bin, series): the bin's points are the contiguous range [start[b], start[b+1])
// of the ascending frequency axis x.  bin_power_spectrum's two statistics (psd.py:186-227):
//   stat = trapz(y, x) / (x_last - x_first)                          (one point: y itself)
//   err  = std(y) / sqrt(n) * mean(x) / (x_last - x_first) / constant (one point: y itself)
// and NaN for an empty bin (scipy.stats.binned_statistic's value for a failing statistic).
__global__ void __launch_bounds__(256)
k_psd_bin(const int64_t M, const int nb, const double *__restrict__ x,
          const double *__restrict__ power, const int64_t *__restrict__ start,
          const double constant, double *__restrict__ stat, double *__restrict__ err) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const size_t r = blockIdx.y;
    const int64_t s = start[b], e = start[b + 1], n = e - s;
    const double *y = power + r * (size_t)M;
    double *st = stat + r * (size_t)nb + b, *er = err + r * (size_t)nb + b;
    if (n <= 0) {
        if (threadIdx.x == 0) { *st = __builtin_nan(""); *er = __builtin_nan(""); }
        return;
    }
    const double span = x[e - 1] - x[s];
    if (n == 1 || !(span > 0.0)) {
        if (threadIdx.x == 0) { *st = y[s]; *er = y[s]; }
        return;
    }
    double tz = 0.0, sy = 0.0, sx = 0.0;
    for (int64_t i = s + threadIdx.x; i < e; i += 256) {
        const double yi = y[i], xi = x[i];
        sy += yi;
        sx += xi;
        if (i + 1 < e) tz += (x[i + 1] - xi) * (yi + y[i + 1]) * 0.5;
    }
    tz = wg_sum256(tz, red);
    sy = wg_sum256(sy, red);
    sx = wg_sum256(sx, red);
    const double mean = sy / (double)n;
    double ss = 0.0;
    for (int64_t i = s + threadIdx.x; i < e; i += 256) {
        const double dv = y[i] - mean;
        ss = fma(dv, dv, ss);
    }
    ss = wg_sum256(ss, red);
    if (threadIdx.x == 0) {
        *st = tz / span;
        *er = sqrt(ss / (double)n) / sqrt((double)n) * (sx / (double)n) / span / constant;
    }
}

// ------------------------------------------------------------------------------------
// Light-curve ingestion: fill the missing cadences of an otherwise evenly
// sampled series by linear interpolation -- interpolate_missing_data (gadfly/interp.py:6-60), what
// the reference runs before every FFT power spectrum (psd.py:495, :531).  Times ascending.
//   cadence index c_i = rint((t_i - t_0) / dt)  (or the given cadence numbers, minus the first);
//   after point i the indices c_i + 1 .. c_{i+1} - 1 are missing: grid time x = t_0 + index * dt,
//   flux np.interp(x, t, f) = slope * (x - t_j) + f_j on the interval that holds x,
// each product and sum rounded on its own, as numpy forms them (bit-identical results).
// Steps: counts (1 + gap) per point, an exclusive scan, then the merge by time (below).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t cadence_index(const double *t, const int64_t *cad, int64_t i,
                                                 double t0, double dt) {
#pragma clang fp contract(off)
    return cad ? (cad[i] - cad[0]) : (int64_t)rint((t[i] - t0) / dt);
}

constexpr int SCAN_PER_WG = 2048;       // elements per 256-thread workgroup (8 per thread)

// counts[i] = 1 + number of cadences missing after point i; block_sums[wg] = sum over the workgroup's
// SCAN_PER_WG elements; offsets[i] = exclusive scan of counts WITHIN the workgroup
__global__ void __launch_bounds__(256)
k_interp_count(const int64_t N, const double *__restrict__ t, const int64_t *__restrict__ cad,
               const double t0, const double dt, int64_t *__restrict__ offsets,
               int64_t *__restrict__ block_sums) {
    __shared__ int64_t part[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_WG + (int64_t)threadIdx.x * 8;
    int64_t loc[8], run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t i = base + k;
        int64_t cnt = 0;
        if (i < N) {
            cnt = 1;
            if (i + 1 < N) {
                const int64_t gap = cadence_index(t, cad, i + 1, t0, dt) - cadence_index(t, cad, i, t0, dt) - 1;
                if (gap > 0) cnt += gap;
            }
        }
        loc[k] = run;
        run += cnt;
    }
    part[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan of the 256 partials
        const int64_t v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const int64_t before = part[threadIdx.x] - run;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + k < N) offsets[base + k] = before + loc[k];
    if (threadIdx.x == 255) block_sums[blockIdx.x] = part[255];
}

// exclusive scan of the block sums in place (one workgroup, any count); total -> offsets[N]
__global__ void __launch_bounds__(256)
k_interp_scan_top(const int64_t nblocks, const int64_t N, int64_t *__restrict__ block_sums,
                  int64_t *__restrict__ offsets) {
    __shared__ int64_t part[256];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += 256) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t v = (i < nblocks) ? block_sums[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int64_t u = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[N] = carry;
}

__global__ void __launch_bounds__(256)
k_interp_offsets(const int64_t N, const int64_t *__restrict__ block_sums, int64_t *__restrict__ offsets) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) offsets[i] += block_sums[i / SCAN_PER_WG];
}

// The filled series is the reference's merge BY TIME of the input points and the missing cadences'
// grid times t_0 + m dt (interp.py:56-59).  With cadence numbers given, dt is a median and the grid
// drifts against the (e.g. barycentric) time stamps, so a missing cadence's grid time need not lie
// between its neighbours' time stamps: positions come from ranks, not from the gap a cadence sits in.
//   input point i   -> i + #{missing m : x_m <  t_i}
//   missing cadence -> (its rank among the missing) + #{i : t_i <= x_m}      (ties: input points first)
__device__ __forceinline__ double grid_time(const double t0, const int64_t m, const double dt) {
#pragma clang fp contract(off)
    return t0 + (double)m * dt;
}

// G[i] = offsets[i] - i = number of missing cadences before point i's gap
__global__ void __launch_bounds__(256)
k_interp_points(const int64_t N, const double *__restrict__ t, const double *__restrict__ f,
                const int64_t *__restrict__ cad, const double t0, const double dt,
                const int64_t *__restrict__ offsets, double *__restrict__ t_out, double *__restrict__ f_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double ti = t[i];
    // M = smallest cadence index whose grid time is >= t_i (monotone in m: settle it exactly)
    int64_t M = (int64_t)floor((ti - t0) / dt) - 1;
    while (grid_time(t0, M, dt) >= ti) --M;
    while (grid_time(t0, M, dt) < ti) ++M;
    // j = last input point with cadence index < M; missing below M = G[j] + those of j's gap below M
    int64_t lo = 0, hi = N;                         // first point with index >= M
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cadence_index(t, cad, mid, t0, dt) < M) lo = mid + 1; else hi = mid;
    }
    int64_t before = 0;
    if (lo > 0) {
        const int64_t j = lo - 1;
        const int64_t gap = offsets[j + 1] - offsets[j] - 1;
        int64_t part = M - cadence_index(t, cad, j, t0, dt) - 1;
        part = part < 0 ? 0 : (part > gap ? gap : part);
        before = (offsets[j] - j) + part;
    }
    t_out[i + before] = ti;
    f_out[i + before] = f[i];
}

__global__ void __launch_bounds__(256)
k_interp_missing(const int64_t N, const double *__restrict__ t, const double *__restrict__ f,
                 const int64_t *__restrict__ cad, const double t0, const double dt,
                 const int64_t *__restrict__ offsets, double *__restrict__ t_out, double *__restrict__ f_out) {
#pragma clang fp contract(off)      // numpy rounds the product and the sum separately (no FMA)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int64_t G = offsets[i] - i, gap = offsets[i + 1] - offsets[i] - 1;
    if (gap <= 0) return;
    const int64_t ci = cadence_index(t, cad, i, t0, dt);
    int64_t p = i + 1;                              // #{points with t <= x}: grows with k, starts near i
    for (int64_t k = 1; k <= gap; ++k) {
        const double x = grid_time(t0, ci + k, dt);
        if (p > 0 && t[p - 1] > x) {                // the grid has drifted back past point p-1: search below
            int64_t lo = 0, hi = p - 1;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (t[mid] <= x) lo = mid + 1; else hi = mid; }
            p = lo;
        }
        while (p < N && t[p] <= x) ++p;
        // np.interp: clamped outside the series, the sample itself on a knot, else the chord
        double v;
        if (p == 0) v = f[0];
        else if (p == N) v = f[N - 1];
        else {
            const int64_t j = p - 1;
            const double slope = (f[j + 1] - f[j]) / (t[j + 1] - t[j]);
            v = slope * (x - t[j]) + f[j];
        }
        const int64_t o = (G + k - 1) + p;
        t_out[o] = x;
        f_out[o] = v;
    }
}

// ------------------------------------------------------------------------------------
// k_factor2w: the block-scaled recurrence of k_factor2 for widths beyond one wave (64 < W <= 256),
// in k_factor's multi-wave mapping.  Scaled coordinates turn the per-row decay
// S <- P (S + ...) P (two multiplies per element and row) into a rescaling at reset rows only:
// per element and row one FMA for the pending rank-1 update and one for the mat-vec, instead of
// the four operations of k_factor.  Inputs are gf_build_scaled's rows (U~, V~, a, de); the state
// handed from tile to tile is k_factor's layout [RT rows][CT * 64 columns] with the pending
// update folded in, F~ in column form.  Log-likelihood path only (no factor rows are stored).
// ------------------------------------------------------------------------------------
struct Factor2wArgs {
    int64_t N, n_first;
    int W, ld;
    const double *c, *a, *Ut, *Vt, *de, *y;
    int64_t y_bs;
    double *d, *z, *S_state, *F_state;
    int32_t *info;
};

template <int RB, int CT, int NW>
__global__ void __launch_bounds__(64 * NW, 2) k_factor2w(const Factor2wArgs A) {     // 2 waves per SIMD
    static_assert(RB % 4 == 0, "rows per wave must be a multiple of 4");
    constexpr int WPC = CT * 64;
    constexpr int RT = RB * NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    if (A.info[b] != 0) return;         // an earlier tile of this problem already failed
    const int ld = A.ld;
    const size_t pb = (size_t)b * A.N;
    const double *__restrict__ ag = A.a + pb;
    const double *__restrict__ Ug = A.Ut + pb * ld;
    const double *__restrict__ Vg = A.Vt + pb * ld;
    const double *__restrict__ eg = A.de + pb;
    const double *__restrict__ yg = A.y + (size_t)b * A.y_bs;
    double *__restrict__ Sg = A.S_state + (size_t)b * RT * WPC;
    double *__restrict__ Fg = A.F_state + (size_t)b * WPC;

    __shared__ double s_part[2][NW][WPC];
    __shared__ double s_vec[NW][3][WPC];
    double *sv_u = s_vec[wave][0], *sv_w = s_vec[wave][1], *sv_e = s_vec[wave][2];

    double S[RB][CT], Fv[CT], q[CT];         // pending update: S += w q^T, F~ += q z (q = r / d)
    bool colok[CT];
    int colc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int j = c * 64 + lane;
        colok[c] = j < ld;
        colc[c] = colok[c] ? j : (ld - 1);
        Fv[c] = Fg[j];
        q[c] = 0.0;
        sv_w[j] = 0.0;
#pragma unroll
        for (int r = 0; r < RB; ++r) S[r][c] = Sg[(size_t)(wave * RB + r) * WPC + j];
    }
    double zp = 0.0;                    // z of the pending row
    int32_t fail = 0;

    // row operands one row ahead: plain unconditional loads (clamped pad columns, masked at use; the
    // per-row scalars through the vector-load queue): the caller pads every buffer by two rows
    const int vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);
    double un[CT], vn[CT], an, yn, en;
#pragma unroll
    for (int c = 0; c < CT; ++c) { un[c] = Ug[colc[c]]; vn[c] = Vg[colc[c]]; }
    an = ag[vz]; yn = yg[vz]; en = eg[vz];

    for (int64_t n = 0; n < A.N; ++n) {
        double u[CT], v[CT];
        const double a_n = an, y_n = yn;
        const double e_n = read_lane(en, 0);    // wave-uniform: say so
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            u[c] = colok[c] ? un[c] : 0.0;
            v[c] = colok[c] ? vn[c] : 0.0;
            sv_u[c * 64 + lane] = u[c];
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const size_t o = (size_t)(n + 1) * ld + colc[c];
            un[c] = Ug[o];
            vn[c] = Vg[o];
        }
        an = ag[n + 1 + vz]; yn = yg[n + 1 + vz]; en = eg[n + 1 + vz];

        if (e_n >= 0.0) {               // reset row: fold the pending update, decay by exp(-c de)
            double el[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int j = c * 64 + lane;     // (decay rates are read here: resets are rare, registers are not)
                el[c] = (j < A.W) ? exp(-A.c[(size_t)b * A.W + j] * e_n) : 1.0;
                sv_e[c * 64 + lane] = el[c];
            }
            wave_lds_fence();
#pragma unroll
            for (int r0 = 0; r0 < RB; r0 += 4) {    // batches of 4 rows, fenced: hipcc otherwise hoists
                double wi[4], ei[4];                // every LDS read of the loop (and spills)
#pragma unroll
                for (int k = 0; k < 4; ++k) { wi[k] = sv_w[wave * RB + r0 + k]; ei[k] = sv_e[wave * RB + r0 + k]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        S[r0 + k][c] = fma(wi[k], q[c], S[r0 + k][c]) * (ei[k] * el[c]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) { Fv[c] = el[c] * fma(q[c], zp, Fv[c]); q[c] = 0.0; }
            zp = 0.0;
        }
        wave_lds_fence();

        double acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < RB; r0 += 4) {
            double wi[4], ui[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { wi[k] = sv_w[wave * RB + r0 + k]; ui[k] = sv_u[wave * RB + r0 + k]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    S[r0 + k][c] = fma(wi[k], q[c], S[r0 + k][c]);
                    acc[c] = fma(ui[k], S[r0 + k][c], acc[c]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        double tmp[CT];
        if constexpr (NW > 1) {
            const int buf = (int)(n & 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) s_part[buf][wave][c * 64 + lane] = acc[c];
            wg_lds_barrier();
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                double t2 = s_part[buf][0][c * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) t2 += s_part[buf][w2][c * 64 + lane];
                tmp[c] = t2;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) tmp[c] = acc[c];
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            Fv[c] = fma(q[c], zp, Fv[c]);
            s1 = fma(u[c], tmp[c], s1);
            s2 = fma(u[c], Fv[c], s2);
        }
        wave_sum2(s1, s2);
        const double dn = a_n - s1;
        const double zn = y_n - s2;
        if (!(dn > 0.0)) {              // uniform across the whole workgroup
            const int64_t ng = A.n_first + n + 1;
            fail = (int32_t)(ng > 0x7fffffff ? 0x7fffffff : ng);
            break;
        }
        const double inv = fast_rcp(dn);
        zp = zn;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const double r = v[c] - tmp[c];     // pad lanes: 0 - 0
            q[c] = r * inv;
            sv_w[c * 64 + lane] = r;
        }
        if (wave == 0 && lane == 0) { A.d[pb + n] = dn; A.z[pb + n] = zn; }
    }
    if (fail) {
        if (wave == 0 && lane == 0) A.info[b] = fail;
        return;
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = wave * RB + r;
        const double wi = sv_w[i];
#pragma unroll
        for (int c = 0; c < CT; ++c) Sg[(size_t)i * WPC + c * 64 + lane] = fma(wi, q[c], S[r][c]);
    }
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) Fg[c * 64 + lane] = fma(q[c], zp, Fv[c]);
    }
}

// ------------------------------------------------------------------------------------
// dispatch helpers
// ------------------------------------------------------------------------------------
template <int RB, int CT, int NW>
int launch_factor(const FactorArgs &A, int B, int nch, hipStream_t st) {
    hipLaunchKernelGGL((k_factor<RB, CT, NW>), dim3(nch, B), dim3(64 * NW), 0, st, A);
    return check_launch("gf_factor");
}

int dispatch_factor(const FactorArgs &A, int B, int nch, hipStream_t st) {
    const int W = A.W;
    if (W <= 16)  return launch_factor<4, 1, 4>(A, B, nch, st);
    if (W <= 32)  return launch_factor<8, 1, 4>(A, B, nch, st);
    if (W <= 48)  return launch_factor<12, 1, 4>(A, B, nch, st);
    if (W <= 64)  return launch_factor<16, 1, 4>(A, B, nch, st);
    if (W <= 96)  return launch_factor<24, 2, 4>(A, B, nch, st);
    if (W <= 128) return launch_factor<32, 2, 4>(A, B, nch, st);
    if (W <= 192) return launch_factor<24, 3, 8>(A, B, nch, st);
    return launch_factor<32, 4, 8>(A, B, nch, st);
}

template <int RB, int CT, int NW>
int launch_factor2w(const Factor2wArgs &A, int B, hipStream_t st) {
    hipLaunchKernelGGL((k_factor2w<RB, CT, NW>), dim3(B), dim3(64 * NW), 0, st, A);
    return check_launch("gf_factor_scaled");
}

int dispatch_factor2w(const Factor2wArgs &A, int B, hipStream_t st) {      // same shapes as dispatch_factor
    const int W = A.W;
    if (W <= 96)  return launch_factor2w<24, 2, 4>(A, B, st);
    if (W <= 128) return launch_factor2w<32, 2, 4>(A, B, st);
    if (W <= 192) return launch_factor2w<24, 3, 8>(A, B, st);
    return launch_factor2w<32, 4, 8>(A, B, st);
}

// Propagator rows of a factor stored in block-scaled form (rows u~, w~ = r/d and the reset spans de of
// gf_chunk_sweep): in scaled coordinates the row-to-row propagator is 1, and exp(-c de) at a reset row.
// With these rows the general-width sweeps (gf_solve) run on the scaled factor unchanged.
__global__ void __launch_bounds__(256)
k_scaled_propagator(const int64_t N, const int W, const int ld, const double *__restrict__ c_,
                    const double *__restrict__ de_, double *__restrict__ P_out) {
    const int b = blockIdx.y;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = id / ld;
    if (n >= N) return;
    const int j = (int)(id - n * ld);
    const double de = de_[(size_t)b * N + n];
    const double cj = (j < W) ? c_[(size_t)b * W + j] : 0.0;
    P_out[((size_t)b * N + n) * ld + j] = (de > 0.0) ? exp(-cj * de) : 1.0;
}

// wide fused sweep: NW = sweep waves for W + 1 columns (32 per wave, one spare column for the forward
// solve), TR = rows per lane = W / 4 rounded up to a multiple of 4; one more wave generates the rows
struct WideShape { int nw, tr; };
inline WideShape wide_shape(int W) {
    WideShape s;
    s.nw = (W + 1 + 31) / 32;
    s.tr = (W + 15) / 16 * 4;
    return s;
}

template <int TR, int NW>
int launch_factorw(const FactorWArgs &A, const FactorWPtrs &P, int grid, hipStream_t st) {
    hipLaunchKernelGGL((k_factorw<TR, NW>), dim3(grid), dim3(64 * (NW + 1)), 0, st, A, P.ac, P.bc, P.cc, P.dc,
                       P.diag_add, P.cmax, P.t, P.diag, P.y, P.d, P.z, P.r_out, P.Ut_out, P.Wt_out, P.de_out,
                       P.S_state, P.info);
    return 0;
}

int dispatch_factorw(const FactorWArgs &A, const FactorWPtrs &P, int W, int grid, hipStream_t st) {
    const WideShape s = wide_shape(W);
#define GF_FW(TRv, NWv) if (s.tr == TRv && s.nw == NWv) return launch_factorw<TRv, NWv>(A, P, grid, st);
    GF_FW(16, 3) GF_FW(20, 3) GF_FW(24, 3)
    GF_FW(24, 4) GF_FW(28, 4) GF_FW(32, 4)
    GF_FW(32, 5) GF_FW(36, 5) GF_FW(40, 5)
    GF_FW(40, 6) GF_FW(44, 6)
#undef GF_FW
    return -1;
}

}  // namespace

// ======================================================================================
// C-ABI
// ======================================================================================
// (shared with gadfly_dense.hip: `which` = 0, 1 here, 2 there)
bool gf_internal_lds_opt_in(int which, hipStream_t st, const void *const *funcs, int nfuncs, size_t bytes) {
    static std::atomic<bool> done[4][64];
    if (which < 0 || which >= 4) return false;
    int dev = -1;
    if (st == nullptr || hipStreamGetDevice(st, &dev) != hipSuccess) {
        if (hipGetDevice(&dev) != hipSuccess) return false;
    }
    if (dev < 0 || dev >= 64) return false;
    if (done[which][dev].load(std::memory_order_acquire)) return true;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return false;
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return false;
    bool ok = true;
    for (int i = 0; i < nfuncs; ++i)
        ok = ok && hipFuncSetAttribute(funcs[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    if (cur != dev) (void)hipSetDevice(cur);
    if (ok) done[which][dev].store(true, std::memory_order_release);
    return ok;
}


extern "C" {

int gf_version(void) { return 100; }

const char *gf_last_error(void) { return g_err; }

int gf_leading_dim(int W) {
    if (W < 1 || W > GF_MAX_WIDTH) return -1;
    return (W + 15) / 16 * 16;
}

int gf_build_matrices(int B, int64_t N, int64_t n_first, int Jr, int Jc, int ld,
                      const double *ar, const double *cr, const double *ac,
                      const double *bc, const double *cc, const double *dc,
                      const double *diag_add,
                      const double *t, int64_t t_bs,
                      const double *diag, int64_t diag_bs,
                      double *a, double *U, double *V, double *P, void *stream) {
    const int W = Jr + 2 * Jc;
    if (B < 1 || N < 1) return set_err("gf_build_matrices: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_build_matrices: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_build_matrices: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (!t || !U || !V || (a && !diag_add)) return set_err("gf_build_matrices: null pointer%s", "");
    BuildArgs A;
    if (n_first < 0) return set_err("gf_build_matrices: negative n_first%s", "");
    A.N = N; A.n_first = n_first; A.Jr = Jr; A.Jc = Jc; A.ld = ld; A.units = Jr + Jc + (ld - W);
    A.ar = ar; A.cr = cr; A.ac = ac; A.bc = bc; A.cc = cc; A.dc = dc; A.diag_add = diag_add;
    A.t = t; A.t_bs = t_bs; A.diag = diag; A.diag_bs = diag_bs;
    A.a = a; A.U = U; A.V = V; A.P = P;
    const int64_t total = N * A.units;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return set_err("gf_build_matrices: problem too large%s", "");
    hipLaunchKernelGGL(k_build, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, A);
    return check_launch("gf_build_matrices");
}

int64_t gf_state_size(int W) {
    if (W < 1 || W > GF_MAX_WIDTH) return -1;
    if (W <= 16)  return 16 * 64;
    if (W <= 32)  return 32 * 64;
    if (W <= 48)  return 48 * 64;
    if (W <= 64)  return 64 * 64;
    if (W <= 96)  return 96 * 128;
    if (W <= 128) return 128 * 128;
    if (W <= 192) return 192 * 192;
    return 256 * 256;
}

int gf_state_cols(int W) {
    if (W < 1 || W > GF_MAX_WIDTH) return -1;
    return (W + 63) / 64 * 64;
}

int gf_factor(int B, int64_t N, int64_t n_first, int W, int ld,
              const double *a, const double *U, const double *V, const double *P,
              const double *y, int64_t y_bs,
              double *d, double *Wm, double *z,
              double *S_state, double *F_state, int32_t *info, void *stream) {
    if (B < 1 || N < 1) return set_err("gf_factor: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_factor: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_factor: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (!a || !U || !V || !P || !d || !info) return set_err("gf_factor: null pointer%s", "");
    if (z && !y) return set_err("gf_factor: z requested without y%s", "");
    FactorArgs A;
    if (F_state && !S_state) return set_err("gf_factor: F_state without S_state%s", "");
    A.N = N; A.chunk_len = N; A.n_first = n_first; A.W = W; A.ld = ld;
    A.a = a; A.U = U; A.V = V; A.P = P; A.y = y; A.y_bs = y_bs;
    A.d = d; A.Wm = Wm; A.z = z;
    A.S_in = S_state; A.F_in = F_state; A.S_out = S_state; A.F_out = F_state;
    A.info = info;
    return dispatch_factor(A, B, 1, (hipStream_t)stream);
}

int gf_scaled_supported(int W) { return (W >= 1 && W <= 64) ? 1 : 0; }

// widths the block-scaled build + multi-wave sweep (k_build2 + k_factor2w) take beyond one wave
int gf_scaled_wide_supported(int W) { return (W > 64 && W <= 192) ? 1 : 0; }   // (beyond: register spills, gf_factor is faster)

int gf_build_scaled(int B, int64_t N, int64_t n_first, int Jr, int Jc, int ld,
                    const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc,
                    const double *diag_add, const double *cmax, int block,
                    const double *t, int64_t t_bs,
                    const double *diag, int64_t diag_bs,
                    double *a, double *Ut, double *Vt, double *de, void *stream) {
    const int W = Jr + 2 * Jc;
    if (B < 1 || N < 1) return set_err("gf_build_scaled: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (!gf_scaled_supported(W) && !gf_scaled_wide_supported(W)) return set_err("gf_build_scaled: width %s%lld unsupported", "", W);
    if (ld < W || (ld & 15)) return set_err("gf_build_scaled: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (block < 1 || block > 64 || (block & (block - 1))) return set_err("gf_build_scaled: block=%s%lld must be a power of two in 1..64", "", block);
    if (n_first < 0 || (n_first % block) != 0) return set_err("gf_build_scaled: n_first=%s%lld must be a non-negative multiple of block=%lld", "", n_first, block);
    if (!t || !a || !Ut || !Vt || !de || !diag_add || !cmax) return set_err("gf_build_scaled: null pointer%s", "");
    Build2Args A;
    A.N = N; A.n_first = n_first; A.Jr = Jr; A.Jc = Jc; A.ld = ld; A.units = Jr + Jc + (ld - W);
    A.block = block; A.gap = (block > 1) ? SC_SPAN / (double)(block - 1) : 0.0;
    A.ar = ar; A.cr = cr; A.ac = ac; A.bc = bc; A.cc = cc; A.dc = dc; A.diag_add = diag_add; A.cmax = cmax;
    A.t = t; A.t_bs = t_bs; A.diag = diag; A.diag_bs = diag_bs;
    A.a = a; A.Ut = Ut; A.Vt = Vt; A.de = de;
    const int64_t blocks = (N + B2_ROWS - 1) / B2_ROWS;
    if (blocks > 0x7fffffffLL) return set_err("gf_build_scaled: problem too large%s", "");
    hipLaunchKernelGGL(k_build2, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, A);
    return check_launch("gf_build_scaled");
}

#define GF_F2_CASE(R) case R: hipLaunchKernelGGL((k_factor2<R>), dim3(B), dim3(64), 0, st, N, n_first, ld, W, c, a, Ut, Vt, de, y, y_bs, d, z, S_state, F_state, info); break;

int gf_factor_scaled(int B, int64_t N, int64_t n_first, int W, int ld, const double *c,
                     const double *a, const double *Ut, const double *Vt, const double *de,
                     const double *y, int64_t y_bs,
                     double *d, double *z, double *S_state, double *F_state,
                     int32_t *info, void *stream) {
    if (B < 1 || N < 1) return set_err("gf_factor_scaled: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (!gf_scaled_supported(W) && !gf_scaled_wide_supported(W)) return set_err("gf_factor_scaled: width %s%lld unsupported", "", W);
    if (ld < W || (ld & 15)) return set_err("gf_factor_scaled: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (!c || !a || !Ut || !Vt || !de || !y || !d || !z || !S_state || !F_state || !info)
        return set_err("gf_factor_scaled: null pointer%s", "");
    hipStream_t st = (hipStream_t)stream;
    if (W > 64) {                       // state: gf_state_size(W) / gf_state_cols(W) doubles per problem
        Factor2wArgs A;
        A.N = N; A.n_first = n_first; A.W = W; A.ld = ld;
        A.c = c; A.a = a; A.Ut = Ut; A.Vt = Vt; A.de = de; A.y = y; A.y_bs = y_bs;
        A.d = d; A.z = z; A.S_state = S_state; A.F_state = F_state; A.info = info;
        return dispatch_factor2w(A, B, st);
    }
    const int rows = (W + 3) / 4 * 4;
    switch (rows) {
        GF_F2_CASE(4) GF_F2_CASE(8) GF_F2_CASE(12) GF_F2_CASE(16) GF_F2_CASE(20) GF_F2_CASE(24)
        GF_F2_CASE(28) GF_F2_CASE(32) GF_F2_CASE(36) GF_F2_CASE(40) GF_F2_CASE(44) GF_F2_CASE(48)
        GF_F2_CASE(52) GF_F2_CASE(56) GF_F2_CASE(60) GF_F2_CASE(64)
        default: return set_err("gf_factor_scaled: internal dispatch error%s", "");
    }
    return check_launch("gf_factor_scaled");
}

// Fused sweeps exist for W <= 63 (any mix of terms) and, for kernels of complex terms only, up to
// W = 176 (k_factorw).  gf_fused_state_size: doubles of S_state per (problem, chunk) slot.
int gf_fused_supported(int Jr, int Jc) {
    const int W = Jr + 2 * Jc;
    if (Jr < 0 || Jc < 0 || W < 1) return 0;
    if (W <= 63) return 1;
    return (Jr == 0 && W <= 176) ? 1 : 0;
}

int64_t gf_fused_state_size(int Jr, int Jc) {
    if (!gf_fused_supported(Jr, Jc)) return -1;
    const int W = Jr + 2 * Jc;
    if (W <= 63) return 64 * 64;
    const WideShape ws = wide_shape(W);
    return (int64_t)(32 * ws.nw) * (4 * ws.tr);                                // [columns][rows], padded
}

// which fused sweep runs (argument `variant` of gf_loglike_fused / gf_chunk_sweep / gf_chunk_transition)
//   GF_SWEEP_AUTO   the lane-tiled k_factor7 / k_phi7 for kernels of complex terms only (Jr = 0,
//                   Jc <= 31: every gadfly kernel with Q >= 1/2), k_factor3 / k_phi otherwise
//   GF_SWEEP_COLUMN k_factor3 / k_phi (one column per lane), any term mix
//   GF_SWEEP_TILED  k_factor7 / k_phi7; an error if the term structure does not allow it
static bool sweep_tiled(int variant, int Jr, int Jc) {
    return variant != GF_SWEEP_COLUMN && Jr == 0 && Jc <= 31;
}

#define GF_F3_ARGS dim3(B * chunk_count), dim3(64), 0, st, N, n_first, chunk_len, nch, chunk_first, chunk_count, Jr, Jc, (block | (gen_period << 8) | zero_start), gap, ar, cr, ac, bc, cc, dc, diag_add, cmax, t, t_bs, diag, diag_bs, y, y_bs, d, z, r_out, Ut_out, Wt_out, de_out, S_state, F_state, info
#define GF_F3_CASE(R) case R: if (tiled && rowstore) hipLaunchKernelGGL((k_factor7<R, true>), GF_F3_ARGS); else if (tiled) hipLaunchKernelGGL((k_factor7<R, false>), GF_F3_ARGS); else hipLaunchKernelGGL((k_factor3<R>), GF_F3_ARGS); break;

static int check_sweep_options(const char *who, int gen_period, int variant, int Jr, int Jc) {
    if (gen_period < 1 || gen_period > 64 || (gen_period & (gen_period - 1)))
        return set_err("%s: gen_period=%lld must be a power of two in 1..64", who, gen_period);
    if (variant < GF_SWEEP_AUTO || variant > GF_SWEEP_TILED)
        return set_err("%s: unknown sweep variant %lld", who, variant);
    if (variant == GF_SWEEP_TILED && !(Jr == 0 && Jc <= 31))
        return set_err("%s: GF_SWEEP_TILED needs complex terms only (Jr = 0) and Jc <= 31 (Jc=%lld)", who, Jc);
    return 0;
}

static int fused_launch(const char *who, int B, int64_t N, int64_t n_first, int64_t chunk_len, int nch,
                        int chunk_first, int chunk_count,
                        int Jr, int Jc, int block, int gen_period, int variant,
                        const double *ar, const double *cr, const double *ac,
                        const double *bc, const double *cc, const double *dc,
                        const double *diag_add, const double *cmax,
                        const double *t, int64_t t_bs, const double *diag, int64_t diag_bs,
                        const double *y, int64_t y_bs,
                        double *d, double *z, double *r_out, double *Ut_out, double *Wt_out,
                        double *de_out, double *S_state, double *F_state,
                        int32_t *info, void *stream) {
    const int W = Jr + 2 * Jc;
    if (B < 1 || N < 1) return set_err("%s: empty problem (N=%lld)", who, N);
    // (u~ rows / reset spans and the w~ rows are stored independently: a nominal pass needs no w~ rows)
    if ((Ut_out != nullptr) != (de_out != nullptr) || (Wt_out && !Ut_out))
        return set_err("%s: Ut_out and de_out go together, Wt_out needs them", who);
    if (!gf_fused_supported(Jr, Jc))
        return set_err("%s: width %lld unsupported (1..63 any terms; 64..176 complex terms only)", who, W);
    if (block < 1 || block > 64 || (block & (block - 1))) return set_err("%s: block=%lld must be a power of two in 1..64", who, block);
    if (n_first < 0 || (n_first % block) != 0) return set_err("%s: n_first=%lld must be a non-negative multiple of block=%lld", who, n_first, block);
    if (nch < 1 || chunk_len < 1 || (nch > 1 && (chunk_len % block) != 0) || (int64_t)nch * chunk_len < N
        || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("%s: bad chunking (chunk_len=%lld, nch=%lld)", who, chunk_len, nch);
    if (chunk_first < 0 || chunk_count < 0 || chunk_first + chunk_count > nch)
        return set_err("%s: bad chunk range (first=%lld, count=%lld)", who, chunk_first, chunk_count);
    if (!t || !y || !d || !z || !S_state || (!F_state && W <= 63) || !info || !diag_add || !cmax)
        return set_err("%s: null pointer", who);
    const int zero_start = (variant & GF_SWEEP_ZERO_START) ? (1 << 30) : 0;
    variant &= ~GF_SWEEP_ZERO_START;
    if (check_sweep_options(who, gen_period, W > 63 ? GF_SWEEP_AUTO : variant, Jr, Jc)) return -1;
    if (chunk_count == 0) return 0;
    const bool tiled = sweep_tiled(variant, Jr, Jc);
    const double gap = (block > 1) ? SC_SPAN / (double)(block - 1) : 0.0;
    hipStream_t st = (hipStream_t)stream;
    if (W > 63) {                       // wide kernels: one workgroup per (problem, chunk), k_factorw
        FactorWArgs A;
        A.N = N; A.n_first = n_first; A.chunk_len = chunk_len; A.nch = nch; A.ch0 = chunk_first; A.nsel = chunk_count; A.Jc = Jc;
        A.block_sub = block | (gen_period << 8) | zero_start; A.gap = gap;
        A.t_bs = t_bs; A.diag_bs = diag_bs; A.y_bs = y_bs;
        FactorWPtrs P;
        P.ac = ac; P.bc = bc; P.cc = cc; P.dc = dc; P.diag_add = diag_add; P.cmax = cmax;
        P.t = t; P.diag = diag; P.y = y;
        P.d = d; P.z = z; P.r_out = r_out; P.Ut_out = Ut_out; P.Wt_out = Wt_out; P.de_out = de_out;
        P.S_state = S_state; P.info = info;
        if (dispatch_factorw(A, P, W, B * chunk_count, st)) return set_err("%s: internal dispatch error (wide)", who);
        return check_launch(who);
    }
    const int rows = (W + 3) / 4 * 4;
    const bool rowstore = r_out || Ut_out || Wt_out || de_out;
    switch (rows) {
        GF_F3_CASE(4) GF_F3_CASE(8) GF_F3_CASE(12) GF_F3_CASE(16) GF_F3_CASE(20) GF_F3_CASE(24)
        GF_F3_CASE(28) GF_F3_CASE(32) GF_F3_CASE(36) GF_F3_CASE(40) GF_F3_CASE(44) GF_F3_CASE(48)
        GF_F3_CASE(52) GF_F3_CASE(56) GF_F3_CASE(60) GF_F3_CASE(64)
        default: return set_err("%s: internal dispatch error", who);
    }
    return check_launch(who);
}

int gf_loglike_fused(int B, int64_t N, int64_t n_first, int Jr, int Jc, int block,
                     int gen_period, int variant,
                     const double *ar, const double *cr, const double *ac,
                     const double *bc, const double *cc, const double *dc,
                     const double *diag_add, const double *cmax,
                     const double *t, int64_t t_bs, const double *diag, int64_t diag_bs,
                     const double *y, int64_t y_bs,
                     double *d, double *z, double *S_state, double *F_state,
                     int32_t *info, void *stream) {
    return fused_launch("gf_loglike_fused", B, N, n_first, N, 1, 0, 1, Jr, Jc, block, gen_period, variant, ar, cr, ac, bc, cc, dc,
                        diag_add, cmax, t, t_bs, diag, diag_bs, y, y_bs, d, z, nullptr,
                        nullptr, nullptr, nullptr, S_state, F_state, info, stream);
}

int gf_chunk_sweep(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count,
                   int Jr, int Jc, int block, int gen_period, int variant,
                   const double *ar, const double *cr, const double *ac,
                   const double *bc, const double *cc, const double *dc,
                   const double *diag_add, const double *cmax,
                   const double *t, int64_t t_bs, const double *diag, int64_t diag_bs,
                   const double *y, int64_t y_bs,
                   double *d, double *z, double *r_out, double *Ut_out, double *Wt_out,
                   double *de_out, double *S_state, double *F_state,
                   int32_t *info, void *stream) {
    return fused_launch("gf_chunk_sweep", B, N, 0, chunk_len, nch, chunk_first, chunk_count, Jr, Jc, block, gen_period, variant, ar, cr, ac, bc, cc, dc,
                        diag_add, cmax, t, t_bs, diag, diag_bs, y, y_bs, d, z, r_out,
                        Ut_out, Wt_out, de_out, S_state, F_state, info, stream);
}

#define GF_PHI_ARGS dim3(B * chunk_count), dim3(64), 0, st, N, chunk_len, nch, chunk_first, chunk_count, W, c, de, dbar, zbar, rbar, Ut, Phi_out, G_out, m_out
#define GF_PHI_CASE(R) case R: if (tiled) hipLaunchKernelGGL((k_phi7<R>), GF_PHI_ARGS); else hipLaunchKernelGGL((k_phi<R>), GF_PHI_ARGS); break;

int gf_chunk_transition(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count,
                        int Jr, int Jc, int variant, const double *c, const double *de, const double *dbar,
                        const double *zbar, const double *rbar, const double *Ut,
                        double *Phi_out, double *G_out, double *m_out, void *stream) {
    const int W = Jr + 2 * Jc;
    if (B < 1 || N < 1) return set_err("gf_chunk_transition: empty problem (N=%s%lld)", "", N);
    if (W < 1 || W > 63) return set_err("gf_chunk_transition: width %s%lld unsupported (1..63)", "", W);
    if (nch < 1 || chunk_len < 1 || (nch > 1 && (chunk_len % 64) != 0) || (int64_t)nch * chunk_len < N
        || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("gf_chunk_transition: bad chunking (chunk_len=%s%lld, nch=%lld)", "", chunk_len, nch);
    if (chunk_first < 0 || chunk_count < 0 || chunk_first + chunk_count > nch)
        return set_err("gf_chunk_transition: bad chunk range (first=%s%lld, count=%lld)", "", chunk_first, chunk_count);
    if (!c || !de || !dbar || !zbar || !rbar || !Ut || !Phi_out || !G_out || !m_out)
        return set_err("gf_chunk_transition: null pointer%s", "");
    if (check_sweep_options("gf_chunk_transition", 1, variant, Jr, Jc)) return -1;
    if (chunk_count == 0) return 0;
    const bool tiled = sweep_tiled(variant, Jr, Jc);
    hipStream_t st = (hipStream_t)stream;
    const int rows = (W + 3) / 4 * 4;
    switch (rows) {
        GF_PHI_CASE(4) GF_PHI_CASE(8) GF_PHI_CASE(12) GF_PHI_CASE(16) GF_PHI_CASE(20) GF_PHI_CASE(24)
        GF_PHI_CASE(28) GF_PHI_CASE(32) GF_PHI_CASE(36) GF_PHI_CASE(40) GF_PHI_CASE(44) GF_PHI_CASE(48)
        GF_PHI_CASE(52) GF_PHI_CASE(56) GF_PHI_CASE(60) GF_PHI_CASE(64)
        default: return set_err("gf_chunk_transition: internal dispatch error%s", "");
    }
    return check_launch("gf_chunk_transition");
}

// the > 64 KB dynamic-LDS opt-in is a per-device function attribute: remember it per device -- the
// device the STREAM belongs to (not the calling thread's current device) -- and apply it with that
// device current.  Returns false if the attribute could not be set.
static bool lds_opt_in(int which, hipStream_t st, const void *const *funcs, int nfuncs, size_t bytes) {
    return gf_internal_lds_opt_in(which, st, funcs, nfuncs, bytes);
}

static int cb_ns(int W) { return W > 48 ? 64 : (W > 32 ? 48 : 32); }
static size_t cb_lds_bytes(int W) {      // 48 x 48 matrices: two workgroups per CU; 32 x 32: four
    const int NS = cb_ns(W);
    return sizeof(double) * ((size_t)NS * (2 * (NS + 1) + 2 * NS + 2) + 256);
}

int gf_chunk_combine(int B, int nch, int W, const double *Phi, const double *G, const double *m,
                     double *S_state, double *F_state, void *stream) {
    if (B < 1 || nch < 1) return set_err("gf_chunk_combine: empty problem%s", "");
    if (W < 1 || W > 64) return set_err("gf_chunk_combine: width %s%lld unsupported (1..64)", "", W);
    if (!Phi || !G || !m || !S_state || !F_state) return set_err("gf_chunk_combine: null pointer%s", "");
    const size_t lds = cb_lds_bytes(W);
    const void *fn[3] = {(const void *)k_combine<64>, (const void *)k_combine<48>, (const void *)k_combine<32>};
    if (!lds_opt_in(0, (hipStream_t)stream, fn, 3, cb_lds_bytes(64))) return set_err("gf_chunk_combine: cannot opt in to %s%lld bytes of LDS", "", (long long)lds);
    hipStream_t st = (hipStream_t)stream;
    switch (cb_ns(W)) {
    case 64: hipLaunchKernelGGL(k_combine<64>, dim3(B), dim3(256), lds, st, nch, W, Phi, G, m, S_state, F_state); break;
    case 48: hipLaunchKernelGGL(k_combine<48>, dim3(B), dim3(256), lds, st, nch, W, Phi, G, m, S_state, F_state); break;
    default: hipLaunchKernelGGL(k_combine<32>, dim3(B), dim3(256), lds, st, nch, W, Phi, G, m, S_state, F_state); break;
    }
    return check_launch("gf_chunk_combine");
}

// Tree (log-depth) version of gf_chunk_combine, on the same arrays [B * nch] (Phi, G, m and S/F = the nominal end
// states; Xst/Yst [B * nch] receive the TRUE start states).  The scan works on P slots per problem, P the power of
// two with P / 2 < nch <= P: the slots beyond nch stand for identity maps and are never read (a pair whose right
// range is padding copies its left map, one that is padding altogether is not launched); what the scan writes at
// such an index -- composites that reach into the padding, their states -- lives in `work`
// (gf_chunk_combine_tree_work doubles).  The map arrays are overwritten.
int64_t gf_chunk_combine_tree_work(int B, int P, int nch) {
    if (B < 1 || P < 2 || nch < 1 || nch > P) return -1;
    const int64_t n = (int64_t)B * (P - nch);
    return n * (4 * 4096 + 3 * 64) + 8;
}

int gf_chunk_combine_tree(int B, int P, int nch, int W, double *Phi, double *G, double *m, double *S, double *F,
                          double *Xst, double *Yst, double *work, void *stream) {
    if (B < 1 || P < 2 || (P & (P - 1))) return set_err("gf_chunk_combine_tree: P=%s%lld must be a power of two >= 2", "", P);
    if (nch <= P / 2 || nch > P) return set_err("gf_chunk_combine_tree: nch=%s%lld must lie in (P / 2, P]", "", nch);
    if (W < 1 || W > 64) return set_err("gf_chunk_combine_tree: width %s%lld unsupported (1..64)", "", W);
    if (!Phi || !G || !m || !S || !F || !Xst || !Yst || !work) return set_err("gf_chunk_combine_tree: null pointer%s", "");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = cb_lds_bytes(W);
    const void *fn[6] = {(const void *)k_tree_compose<64>, (const void *)k_tree_apply<64>,
                         (const void *)k_tree_compose<48>, (const void *)k_tree_apply<48>,
                         (const void *)k_tree_compose<32>, (const void *)k_tree_apply<32>};
    if (!lds_opt_in(1, st, fn, 6, cb_lds_bytes(64))) return set_err("gf_chunk_combine_tree: cannot opt in to %s%lld bytes of LDS", "", (long long)lds);
    TreeArgs A;
    A.P = P; A.n = W; A.nch = nch; A.Phi = Phi; A.G = G; A.S = S; A.F = F; A.m = m; A.Xst = Xst; A.Yst = Yst;
    {
        const size_t no = (size_t)B * (P - nch);
        A.oPhi = work; A.oG = A.oPhi + no * 4096; A.oS = A.oG + no * 4096; A.oX = A.oS + no * 4096;
        A.oF = A.oX + no * 4096; A.om = A.oF + no * 64; A.oY = A.om + no * 64;
    }
    const int ns = cb_ns(W);
    // pairs of a level whose left range [2 d k, 2 d k + d - 1] holds a real chunk; the rest is padding: not launched
    auto real_pairs = [&](int d) { return (nch + 2 * d - 1) / (2 * d); };
    // up-sweep.  Its top level would compose the whole range into the last slot -- a map the down-sweep never
    // applies (it uses LEFT children only): left out, one level's latency less
    for (int d = 1; d < P / 2; d *= 2) {
        A.d = d; A.pairs = real_pairs(d);
        const dim3 grid(B * A.pairs);
        if (ns == 64) hipLaunchKernelGGL(k_tree_compose<64>, grid, dim3(256), lds, st, A);
        else if (ns == 48) hipLaunchKernelGGL(k_tree_compose<48>, grid, dim3(256), lds, st, A);
        else hipLaunchKernelGGL(k_tree_compose<32>, grid, dim3(256), lds, st, A);
    }
    A.d = P / 2; A.pairs = 1;
    hipLaunchKernelGGL(k_tree_top, dim3(B), dim3(256), 0, st, A);                    // root state = zero, first level
    for (int d = P / 4; d >= 1; d /= 2) {   // down-sweep
        A.d = d; A.pairs = real_pairs(d);
        const dim3 grid(B * A.pairs);
        if (ns == 64) hipLaunchKernelGGL(k_tree_apply<64>, grid, dim3(256), lds, st, A);
        else if (ns == 48) hipLaunchKernelGGL(k_tree_apply<48>, grid, dim3(256), lds, st, A);
        else hipLaunchKernelGGL(k_tree_apply<32>, grid, dim3(256), lds, st, A);
    }
    return check_launch("gf_chunk_combine_tree");
}

int gf_chunk_transition_wide(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count,
                             int Jc, const double *c, const double *de, const double *dbar, const double *rbar,
                             const double *Ut, double *h_out, double *Phi_out, void *stream) {
    const int W = 2 * Jc;
    if (chunk_first < 0 || chunk_count < 0 || chunk_first + chunk_count > nch)
        return set_err("gf_chunk_transition_wide: bad chunk range (first=%s%lld, count=%lld)", "", chunk_first, chunk_count);
    if (chunk_count == 0) return 0;
    if (B < 1 || N < 1) return set_err("gf_chunk_transition_wide: empty problem (N=%s%lld)", "", N);
    if (W <= 63 || !gf_fused_supported(0, Jc)) return set_err("gf_chunk_transition_wide: width %s%lld unsupported (64..176)", "", W);
    if (nch < 1 || chunk_len < 1 || (nch > 1 && (chunk_len % 64) != 0) || (int64_t)nch * chunk_len < N
        || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("gf_chunk_transition_wide: bad chunking (chunk_len=%s%lld, nch=%lld)", "", chunk_len, nch);
    if (!c || !de || !dbar || !rbar || !Ut || !h_out || !Phi_out) return set_err("gf_chunk_transition_wide: null pointer%s", "");
    const WideShape ws = wide_shape(W);
    const int CP = 32 * ws.nw, tr8 = ws.tr / 2;     // rows per lane with 8 row groups
    const int64_t grid = (int64_t)B * chunk_count * ((W + 15) / 16);
    if (grid > 0x7fffffffLL) return set_err("gf_chunk_transition_wide: problem too large%s", "");
    hipStream_t st = (hipStream_t)stream;
    // one kilobyte per row vector while the row, its pivot pair and its reset-span pair fit (CP <= 124), else two
    const int ni = (CP * 8 + 32 <= 1024) ? 1 : 2;
#define GF_PW(TRv, Dv, NIv) if (tr8 == TRv && ni == NIv) { hipLaunchKernelGGL((k_phiw<TRv, Dv, NIv>), dim3((unsigned)grid), dim3(64), 0, st, N, chunk_len, nch, chunk_first, chunk_count, W, CP, c, de, dbar, rbar, Ut, h_out, Phi_out); return check_launch("gf_chunk_transition_wide"); }
    GF_PW(8, 4, 1) GF_PW(10, 4, 1) GF_PW(12, 4, 1)
    GF_PW(12, 4, 2) GF_PW(14, 4, 2) GF_PW(16, 4, 2) GF_PW(18, 4, 2) GF_PW(20, 4, 2) GF_PW(22, 4, 2)
#undef GF_PW
    return set_err("gf_chunk_transition_wide: internal dispatch error%s", "");
    return check_launch("gf_chunk_transition_wide");
}

// leading dimension of the rows gf_chunk_sweep stores (Ut_out, Wt_out, r_out): 64 for W <= 63, the padded
// column count of the wide sweep beyond
int gf_fused_row_stride(int Jr, int Jc) {
    if (!gf_fused_supported(Jr, Jc)) return -1;
    const int W = Jr + 2 * Jc;
    return (W <= 63) ? 64 : 32 * wide_shape(W).nw;
}

int gf_scaled_propagator(int B, int64_t N, int W, int ld, const double *c, const double *de,
                         double *P_out, void *stream) {
    if (B < 1 || N < 1) return set_err("gf_scaled_propagator: empty problem (N=%s%lld)", "", N);
    if (W < 1 || ld < W) return set_err("gf_scaled_propagator: bad width / ld (W=%s%lld, ld=%lld)", "", W, ld);
    if (!c || !de || !P_out) return set_err("gf_scaled_propagator: null pointer%s", "");
    const int64_t blocks = (N * ld + 255) / 256;
    if (blocks > 0x7fffffffLL) return set_err("gf_scaled_propagator: problem too large%s", "");
    hipLaunchKernelGGL(k_scaled_propagator, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream,
                       N, W, ld, c, de, P_out);
    return check_launch("gf_scaled_propagator");
}

#define GF_LINR_CASE(Rw) case Rw: if (mode == GF_MATMUL_LOWER && !store) hipLaunchKernelGGL((k_linR7<Rw, true>), dim3(B * nch, (R + 63) / 64), dim3(64), 0, st, A); else hipLaunchKernelGGL((k_linR7<Rw, false>), dim3(B * nch, (R + 63) / 64), dim3(64), 0, st, A); break;

int gf_chunk_linear(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int R,
                    int scale, int store, const double *c,
                    const double *Ut, const double *Wt, const double *d, const double *de,
                    const double *Y, double *Z, double *F_state, void *stream) {
    if (mode < 0 || mode > 2) return set_err("gf_chunk_linear: bad mode %s%lld", "", mode);
    if (B < 1 || N < 1 || R < 1) return set_err("gf_chunk_linear: empty problem (N=%s%lld, R=%lld)", "", N, R);
    if (W < 1 || W > 63) return set_err("gf_chunk_linear: width %s%lld unsupported (1..63)", "", W);
    if (nch < 1 || chunk_len < 1 || (int64_t)nch * chunk_len < N || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("gf_chunk_linear: bad chunking (chunk_len=%s%lld, nch=%lld)", "", chunk_len, nch);
    if (!c || !Ut || !Wt || !d || !de || !Y || !Z || !F_state) return set_err("gf_chunk_linear: null pointer%s", "");
    LinArgs A;
    A.N = N; A.chunk_len = chunk_len; A.nch = nch; A.W = W; A.mode = mode; A.scale = scale; A.store = store;
    A.c = c; A.Ut = Ut; A.Wt = Wt; A.d = d; A.de = de; A.Y = Y; A.Z = Z; A.F_state = F_state; A.R = R;
    hipStream_t st = (hipStream_t)stream;
    if (R == 1) {
        hipLaunchKernelGGL(k_lin1, dim3(B * nch), dim3(64), 0, st, A);
    } else if (mode == GF_MATMUL_LOWER && R >= 16) {
        // no feedback in this mode: blocks of 16 rows as matrix products (k_mmR_mfma)
        // two waves per workgroup share the block's rows (32 right-hand sides each); one when R <= 32
        const int nwv = (R > 32) ? 2 : 1;
        const dim3 grid(B * nch, (R + 32 * nwv - 1) / (32 * nwv));
#define GF_MM_LAUNCH(MTv, ND, NWVv) hipLaunchKernelGGL((k_mmR_mfma<MTv, ND, NWVv>), grid, dim3(64 * NWVv), 0, st, A)
#define GF_MM_CASE(MTv) case MTv: if (store) { if (nwv == 2) GF_MM_LAUNCH(MTv, false, 2); else GF_MM_LAUNCH(MTv, false, 1); } \
                                  else { if (nwv == 2) GF_MM_LAUNCH(MTv, true, 2); else GF_MM_LAUNCH(MTv, true, 1); } break;
        switch ((W + 15) / 16) {
            GF_MM_CASE(1) GF_MM_CASE(2) GF_MM_CASE(3) GF_MM_CASE(4)
            default: return set_err("gf_chunk_linear: internal dispatch error%s", "");
        }
#undef GF_MM_LAUNCH
#undef GF_MM_CASE
    } else {
        const int rows = (W + 3) / 4 * 4;
        switch (rows) {
            GF_LINR_CASE(4) GF_LINR_CASE(8) GF_LINR_CASE(12) GF_LINR_CASE(16) GF_LINR_CASE(20) GF_LINR_CASE(24)
            GF_LINR_CASE(28) GF_LINR_CASE(32) GF_LINR_CASE(36) GF_LINR_CASE(40) GF_LINR_CASE(44) GF_LINR_CASE(48)
            GF_LINR_CASE(52) GF_LINR_CASE(56) GF_LINR_CASE(60) GF_LINR_CASE(64)
            default: return set_err("gf_chunk_linear: internal dispatch error%s", "");
        }
    }
    return check_launch("gf_chunk_linear");
}

int gf_chunk_linear_combine(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int R,
                            const double *c, const double *de, const double *Phi,
                            double *D_work, double *F_state, void *stream) {
    if (mode < 0 || mode > 2) return set_err("gf_chunk_linear_combine: bad mode %s%lld", "", mode);
    if (B < 1 || nch < 1 || R < 1) return set_err("gf_chunk_linear_combine: empty problem%s", "");
    if (!F_state || (mode == GF_MATMUL_LOWER ? (!D_work || !c || !de) : !Phi))
        return set_err("gf_chunk_linear_combine: null pointer%s", "");
    hipStream_t st = (hipStream_t)stream;
    if (mode == GF_MATMUL_LOWER) {
        hipLaunchKernelGGL(k_chunk_decay, dim3(B * nch), dim3(64), 0, st, N, chunk_len, nch, W, c, de, D_work);
        hipLaunchKernelGGL(k_lincombine_mm, dim3(B, 64, (R + 15) / 16), dim3(64 * LCM_WAVES), 0, st, nch, R, 64, D_work, F_state);
        return check_launch("gf_chunk_linear_combine");
    }
    hipLaunchKernelGGL(k_lincombine<0>, dim3(B, R), dim3(256), 0, st, nch, nch, mode, R, Phi, D_work, F_state,
                       (double *)nullptr);
    return check_launch("gf_chunk_linear_combine");
}

int gf_chunk_segment_transitions(int B, int nch, int seg_len, const double *Phi, double *Psi_out,
                                 double *PhiT_out, double *PsiT_out, void *stream) {
    if (B < 1 || nch < 1 || seg_len < 1) return set_err("gf_chunk_segment_transitions: empty problem%s", "");
    if (!Phi || !Psi_out || ((PhiT_out != nullptr) != (PsiT_out != nullptr)))
        return set_err("gf_chunk_segment_transitions: null pointer (PhiT_out and PsiT_out go together)%s", "");
    const int nseg = (nch + seg_len - 1) / seg_len;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_segment_transition, dim3(B * nseg), dim3(256), 0, st, nch, seg_len, Phi, Psi_out);
    if (PhiT_out) {
        hipLaunchKernelGGL(k_transpose64, dim3(B * nch), dim3(256), 0, st, Phi, PhiT_out);
        hipLaunchKernelGGL(k_transpose64, dim3(B * nseg), dim3(256), 0, st, (const double *)Psi_out, PsiT_out);
    }
    return check_launch("gf_chunk_segment_transitions");
}

int gf_chunk_linear_combine_seg(int mode, int B, int nch, int seg_len, int R, const double *Phi,
                                const double *Psi, const double *PhiT, const double *PsiT,
                                double *F_state, double *V_work, void *stream) {
    if (mode != GF_SOLVE_LOWER && mode != GF_SOLVE_UPPER)
        return set_err("gf_chunk_linear_combine_seg: bad mode %s%lld (the solves only)", "", mode);
    if ((PhiT != nullptr) != (PsiT != nullptr)) return set_err("gf_chunk_linear_combine_seg: PhiT and PsiT go together%s", "");
    if (mode == GF_SOLVE_UPPER && PhiT) { Phi = PhiT; Psi = PsiT; mode |= LC_TRANSPOSED; }
    if (B < 1 || nch < 1 || R < 1 || seg_len < 1) return set_err("gf_chunk_linear_combine_seg: empty problem%s", "");
    if (!Phi || !Psi || !F_state || !V_work) return set_err("gf_chunk_linear_combine_seg: null pointer%s", "");
    const int nseg = (nch + seg_len - 1) / seg_len;
    hipStream_t st = (hipStream_t)stream;
    // the segment scans: per right-hand side, or 64 right-hand sides at a time on the matrix pipe
    const bool many = R >= 16;
    const dim3 gridR(B * nseg, (R + 63) / 64);
    if (many) hipLaunchKernelGGL(k_lincombine_R<1>, gridR, dim3(256), 0, st, nch, seg_len, mode, R, Phi, F_state, V_work);
    else hipLaunchKernelGGL(k_lincombine<1>, dim3(B * nseg, R), dim3(256), 0, st, nch, seg_len, mode, R, Phi,
                            (const double *)nullptr, F_state, V_work);
    hipLaunchKernelGGL(k_lincombine<0>, dim3(B, R), dim3(256), 0, st, nseg, nseg, mode, R, Psi,
                       (const double *)nullptr, V_work, (double *)nullptr);
    if (many) hipLaunchKernelGGL(k_lincombine_R<2>, gridR, dim3(256), 0, st, nch, seg_len, mode, R, Phi, F_state, V_work);
    else hipLaunchKernelGGL(k_lincombine<2>, dim3(B * nseg, R), dim3(256), 0, st, nch, seg_len, mode, R, Phi,
                            (const double *)nullptr, F_state, V_work);
    return check_launch("gf_chunk_linear_combine_seg");
}

int gf_dense_solve_logdet(int batch, int n, int nrhs, const double *A, double *B, double *logdet_out, void *stream) {
    if (batch < 1 || n < 1 || nrhs < 1) return set_err("gf_dense_solve: empty problem (n=%s%lld, nrhs=%lld)", "", n, nrhs);
    if (n > 192) return set_err("gf_dense_solve: n=%s%lld unsupported (max %lld)", "", n, 192);
    if (!A || !B) return set_err("gf_dense_solve: null pointer%s", "");
    hipStream_t st = (hipStream_t)stream;
    // few systems: 16 right-hand sides per workgroup (more, shorter-stepping workgroups); else 64
    const int slices4 = (nrhs + DS_WAVES * 4 - 1) / (DS_WAVES * 4);
    const bool narrow = (long long)batch * slices4 <= 64;
    const int per = DS_WAVES * (narrow ? 1 : 4);
    const dim3 grid(batch, (nrhs + per - 1) / per);
#define GF_DS(RSv, NAv) do { if (narrow) hipLaunchKernelGGL((k_dense_solve<RSv, NAv, 1>), grid, dim3(64 * DS_WAVES), 0, st, n, nrhs, A, B, logdet_out); \
                             else hipLaunchKernelGGL((k_dense_solve<RSv, NAv, 4>), grid, dim3(64 * DS_WAVES), 0, st, n, nrhs, A, B, logdet_out); } while (0)
    if (n <= 64) GF_DS(1, 4);
    else if (n <= 96) GF_DS(2, 6);
    else if (n <= 128) GF_DS(2, 8);
    else if (n <= 176) GF_DS(3, 11);
    else GF_DS(3, 12);
#undef GF_DS
    return check_launch("gf_dense_solve");
}

int gf_dense_solve(int batch, int n, int nrhs, const double *A, double *B, void *stream) {
    return gf_dense_solve_logdet(batch, n, nrhs, A, B, nullptr, stream);
}

int64_t gf_reduce_work(int64_t N) { return RED_NACC * (int64_t)red_groups(N); }

int gf_reduce_tile(int B, int64_t N, const double *d, const double *z,
                   double *work, double *acc, int init, void *stream) {
    if (B < 1 || N < 1) return set_err("gf_reduce_tile: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (!d || !work || !acc) return set_err("gf_reduce_tile: null pointer%s", "");
    const int G = red_groups(N);
    hipLaunchKernelGGL(k_reduce1, dim3(G, B), dim3(RED_BLOCK), 0, (hipStream_t)stream, N, d, z, work);
    hipLaunchKernelGGL(k_reduce2, dim3(B), dim3(64), 0, (hipStream_t)stream, G, work, acc, init);
    return check_launch("gf_reduce_tile");
}

int gf_loglike_finish(int B, int64_t N, const double *acc, const int32_t *info,
                      double *out, double *logdet, void *stream) {
    if (B < 1 || N < 1) return set_err("gf_loglike_finish: empty problem (B=%s%lld, N=%lld)", "", B, N);
    if (!acc || (!out && !logdet)) return set_err("gf_loglike_finish: null pointer%s", "");
    hipLaunchKernelGGL(k_finish, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, N, acc, info, out, logdet);
    return check_launch("gf_loglike_finish");
}

// one-right-hand-side sweeps: `grid` single-wave workgroups (problems x chunks)
static void launch_solve_vec(const SolveArgs &A, int grid, hipStream_t st) {
    const int mode = A.mode;
#define GF_SV(CT, M, S) hipLaunchKernelGGL((k_solve_vec<CT, M, S>), dim3(grid), dim3(64), 0, st, A)
#define GF_SV_MODE(CT) do { const bool sd = A.scale != nullptr; \
        if (mode == GF_SOLVE_LOWER) { if (sd) GF_SV(CT, GF_SOLVE_LOWER, true); else GF_SV(CT, GF_SOLVE_LOWER, false); } \
        else if (mode == GF_SOLVE_UPPER) { if (sd) GF_SV(CT, GF_SOLVE_UPPER, true); else GF_SV(CT, GF_SOLVE_UPPER, false); } \
        else { if (sd) GF_SV(CT, GF_MATMUL_LOWER, true); else GF_SV(CT, GF_MATMUL_LOWER, false); } } while (0)
    if (A.ld <= 64)       GF_SV_MODE(1);
    else if (A.ld <= 128) GF_SV_MODE(2);
    else if (A.ld <= 192) GF_SV_MODE(3);
    else                  GF_SV_MODE(4);
#undef GF_SV_MODE
#undef GF_SV
}

// several right-hand sides: one workgroup per (tile of 64 right-hand sides, problem x chunk)
static void launch_solve_rhs(const SolveArgs &A, int slots, hipStream_t st) {
    const int ld = A.ld;
    const dim3 grid((A.R + 63) / 64, slots);
    if (ld <= 16)       hipLaunchKernelGGL((k_solve_rhs<16, 1>), grid, dim3(64), 0, st, A);
    else if (ld <= 32)  hipLaunchKernelGGL((k_solve_rhs<32, 1>), grid, dim3(64), 0, st, A);
    else if (ld <= 48)  hipLaunchKernelGGL((k_solve_rhs<48, 1>), grid, dim3(64), 0, st, A);
    else if (ld <= 64)  hipLaunchKernelGGL((k_solve_rhs<64, 1>), grid, dim3(64), 0, st, A);
    else if (ld <= 96)  hipLaunchKernelGGL((k_solve_rhs<48, 2>), grid, dim3(128), 0, st, A);
    else if (ld <= 128) hipLaunchKernelGGL((k_solve_rhs<64, 2>), grid, dim3(128), 0, st, A);
    else if (ld <= 192) hipLaunchKernelGGL((k_solve_rhs<48, 4>), grid, dim3(256), 0, st, A);
    else                hipLaunchKernelGGL((k_solve_rhs<64, 4>), grid, dim3(256), 0, st, A);
}

// Chunk-parallel form of the one-right-hand-side sweeps of gf_solve (any width; what the wide stored
// factor uses): local pass (F_state zeroed by the caller, store = 0) -> combine of the chunk states on the
// caller's side (GF_MATMUL_LOWER: diagonal decays, gf_chunk_diag_scan; the solves: the chunks' closed-loop
// transitions) -> final pass from the true start states (store = 1).  F_state: [B * nch][ld].
int gf_solve_chunk(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int ld,
                   const double *U, const double *Wm, const double *P, const double *scale,
                   const double *Y, double *Z, double *F_state, int store, void *stream) {
    if (mode < 0 || mode > 2) return set_err("gf_solve_chunk: bad mode %s%lld", "", mode);
    if (B < 1 || N < 1) return set_err("gf_solve_chunk: empty problem (N=%s%lld)", "", N);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_solve_chunk: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_solve_chunk: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (nch < 1 || chunk_len < 1 || (int64_t)nch * chunk_len < N || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("gf_solve_chunk: bad chunking (chunk_len=%s%lld, nch=%lld)", "", chunk_len, nch);
    if (!U || !Wm || !P || !Y || !Z || !F_state) return set_err("gf_solve_chunk: null pointer%s", "");
    if ((int64_t)B * nch > 0x7fffffffLL) return set_err("gf_solve_chunk: problem too large%s", "");
    SolveArgs A;
    A.N = N; A.W = W; A.ld = ld; A.R = 1; A.mode = mode;
    A.U = U; A.Wm = Wm; A.P = P; A.scale = scale; A.Y = Y; A.Z = Z;
    A.chunk_len = chunk_len; A.nch = nch; A.store = store; A.F_state = F_state;
    launch_solve_vec(A, B * nch, (hipStream_t)stream);
    return check_launch("gf_solve_chunk");
}

// gf_solve_chunk with R right-hand sides (k_solve_rhs in chunk mode): Y, Z [B][N][R]; F_state
// [B * nch][ld][R].  The combine between the two passes is the caller's (W x W chunk transitions applied to
// W x R states: batched GEMMs).
int gf_solve_chunk_rhs(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int ld, int R,
                       const double *U, const double *Wm, const double *P, const double *scale,
                       const double *Y, double *Z, double *F_state, int store, void *stream) {
    if (mode < 0 || mode > 2) return set_err("gf_solve_chunk_rhs: bad mode %s%lld", "", mode);
    if (B < 1 || N < 1 || R < 1) return set_err("gf_solve_chunk_rhs: empty problem (N=%s%lld, R=%lld)", "", N, R);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_solve_chunk_rhs: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_solve_chunk_rhs: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (nch < 1 || chunk_len < 1 || (int64_t)nch * chunk_len < N || (int64_t)(nch - 1) * chunk_len >= N)
        return set_err("gf_solve_chunk_rhs: bad chunking (chunk_len=%s%lld, nch=%lld)", "", chunk_len, nch);
    if (!U || !Wm || !P || !Y || !Z || !F_state) return set_err("gf_solve_chunk_rhs: null pointer%s", "");
    if ((int64_t)B * nch > 65535) return set_err("gf_solve_chunk_rhs: too many chunks (B * nch = %s%lld > 65535)", "", (int64_t)B * nch);
    SolveArgs A;
    A.N = N; A.W = W; A.ld = ld; A.R = R; A.mode = mode;
    A.U = U; A.Wm = Wm; A.P = P; A.scale = scale; A.Y = Y; A.Z = Z;
    A.chunk_len = chunk_len; A.nch = nch; A.store = store; A.F_state = F_state;
    launch_solve_rhs(A, B * nch, (hipStream_t)stream);
    return check_launch("gf_solve_chunk_rhs");
}

// Scan of chunk states with DIAGONAL transitions: F_state slot c [rows][R] <- true start state of chunk c
// from the local end states, F_{c+1} = loc_c + D_c o F_c (D [B * nch][rows]); gf_chunk_linear_combine's
// GF_MATMUL_LOWER branch for any number of state rows.
int gf_chunk_diag_scan(int B, int nch, int rows, int R, const double *D, double *F_state, void *stream) {
    if (B < 1 || nch < 1 || rows < 1 || R < 1 || R > 64)
        return set_err("gf_chunk_diag_scan: bad shape (rows=%s%lld, R=%lld)", "", rows, R);
    if (!D || !F_state) return set_err("gf_chunk_diag_scan: null pointer%s", "");
    hipLaunchKernelGGL(k_lincombine_mm, dim3(B, rows, (R + 15) / 16), dim3(64 * LCM_WAVES), 0, (hipStream_t)stream,
                       nch, R, rows, D, F_state);
    return check_launch("gf_chunk_diag_scan");
}

int gf_solve(int mode, int B, int64_t N, int W, int ld, int R,
             const double *U, const double *Wm, const double *P, const double *scale,
             const double *Y, double *Z, void *stream) {
    if (mode < 0 || mode > 2) return set_err("gf_solve: bad mode %s%lld", "", mode);
    if (B < 1 || N < 1 || R < 1) return set_err("gf_solve: empty problem (N=%s%lld, R=%lld)", "", N, R);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_solve: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_solve: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (!U || !Wm || !P || !Y || !Z) return set_err("gf_solve: null pointer%s", "");
    if (mode == GF_MATMUL_LOWER && Y == Z) return set_err("gf_solve: GF_MATMUL_LOWER cannot run in place%s", "");
    SolveArgs A;
    A.N = N; A.W = W; A.ld = ld; A.R = R; A.mode = mode;
    A.U = U; A.Wm = Wm; A.P = P; A.scale = scale; A.Y = Y; A.Z = Z;
    A.chunk_len = N; A.nch = 1; A.store = 1; A.F_state = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (R == 1) {
        launch_solve_vec(A, B, st);
    } else {
        launch_solve_rhs(A, B, st);
    }
    return check_launch("gf_solve");
}

// K(t_n, t*_r) for n < N, r < R into out [B][N][R] (row-major; R <= 4096 query times per call).
int gf_cross_covariance(int B, int64_t N, int R, int Jr, int Jc,
                        const double *ar, const double *cr, const double *ac,
                        const double *bc, const double *cc, const double *dc,
                        const double *t, int64_t t_bs, const double *ts, int64_t ts_bs,
                        double *out, void *stream) {
    if (B < 1 || N < 1 || R < 1 || R > 4096) return set_err("gf_cross_covariance: bad shape (N=%s%lld, R=%lld)", "", N, R);
    if (Jr < 0 || Jc < 0 || Jr + Jc < 1 || Jr + 2 * Jc > GF_MAX_WIDTH) return set_err("gf_cross_covariance: bad term counts%s", "");
    if (!t || !ts || !out || (Jr && (!ar || !cr)) || (Jc && (!ac || !bc || !cc || !dc)))
        return set_err("gf_cross_covariance: null pointer%s", "");
    // queries per pass (accumulators per lane) so that the table of (term, query) factors fits 48 KB of LDS
    const int J = Jr + Jc;
    const int RT = (J <= 44) ? 64 : (J <= 92) ? 32 : 16;
    const size_t lds = sizeof(double) * ((size_t)4 * J + (size_t)2 * RT * J + RT);
    const int64_t blocks = (N + 255) / 256;
    if (blocks > 0x7fffffffLL) return set_err("gf_cross_covariance: problem too large%s", "");
    const dim3 grid((unsigned)blocks, B);
    hipStream_t st = (hipStream_t)stream;
#define GF_CX(RTv) hipLaunchKernelGGL((k_cross<RTv>), grid, dim3(256), lds, st, N, R, Jr, Jc, ar, cr, ac, bc, cc, dc, t, t_bs, ts, ts_bs, out)
    if (RT == 64) GF_CX(64); else if (RT == 32) GF_CX(32); else GF_CX(16);
#undef GF_CX
    return check_launch("gf_cross_covariance");
}

int64_t gf_interp_work(int64_t N) {
    return (N < 1) ? 0 : (N + SCAN_PER_WG - 1) / SCAN_PER_WG;
}

int gf_interp_plan(int64_t N, const double *t, const int64_t *cadences, double dt,
                   int64_t *offsets, int64_t *work, void *stream) {
    if (N < 1) return set_err("gf_interp_plan: empty series (N=%s%lld)", "", N);
    if (!t || !offsets || !work) return set_err("gf_interp_plan: null pointer%s", "");
    if (!(dt > 0.0)) return set_err("gf_interp_plan: the cadence must be positive%s", "");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = gf_interp_work(N);
    double t0;
    if (hipMemcpyAsync(&t0, t, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return set_err("gf_interp_plan: cannot read the first time%s", "");
    hipLaunchKernelGGL(k_interp_count, dim3((unsigned)nb), dim3(256), 0, st, N, t, cadences, t0, dt, offsets, work);
    hipLaunchKernelGGL(k_interp_scan_top, dim3(1), dim3(256), 0, st, nb, N, work, offsets);
    hipLaunchKernelGGL(k_interp_offsets, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, work, offsets);
    return check_launch("gf_interp_plan");
}

int gf_interp_fill(int64_t N, const double *t, const double *f, const int64_t *cadences, double dt,
                   const int64_t *offsets, double *t_out, double *f_out, void *stream) {
    if (N < 1) return set_err("gf_interp_fill: empty series (N=%s%lld)", "", N);
    if (!t || !f || !offsets || !t_out || !f_out) return set_err("gf_interp_fill: null pointer%s", "");
    hipStream_t st = (hipStream_t)stream;
    double t0;
    if (hipMemcpyAsync(&t0, t, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return set_err("gf_interp_fill: cannot read the first time%s", "");
    hipLaunchKernelGGL(k_interp_points, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, t, f, cadences,
                       t0, dt, offsets, t_out, f_out);
    hipLaunchKernelGGL(k_interp_missing, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, t, f, cadences,
                       t0, dt, offsets, t_out, f_out);
    return check_launch("gf_interp_fill");
}

int gf_psd_power(int R, int64_t M, int64_t first, double norm, const double *spec,
                 double *power, void *stream) {
    if (R < 1 || M < 1 || first < 0 || first >= M) return set_err("gf_psd_power: bad shape (M=%s%lld, first=%lld)", "", M, first);
    if (!spec || !power) return set_err("gf_psd_power: null pointer%s", "");
    if (R > 65535) return set_err("gf_psd_power: more than 65535 series%s", "");
    const int64_t Mout = M - first;
    int64_t blocks = (Mout + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_psd_power, dim3((unsigned)blocks, R), dim3(256), 0, (hipStream_t)stream, M, Mout,
                       first, norm, (const double2 *)spec, power);
    return check_launch("gf_psd_power");
}

int gf_psd_bin(int R, int64_t M, int nb, const double *x, const double *power,
               const int64_t *start, double constant, double *stat, double *err, void *stream) {
    if (R < 1 || M < 1 || nb < 1) return set_err("gf_psd_bin: bad shape (M=%s%lld, bins=%lld)", "", M, (int64_t)nb);
    if (!x || !power || !start || !stat || !err) return set_err("gf_psd_bin: null pointer%s", "");
    if (R > 65535) return set_err("gf_psd_bin: more than 65535 series%s", "");
    hipLaunchKernelGGL(k_psd_bin, dim3(nb, R), dim3(256), 0, (hipStream_t)stream, M, nb, x, power, start,
                       constant, stat, err);
    return check_launch("gf_psd_bin");
}

// chunking of the conditional-mean sweeps: ~2048 waves over (problem, direction, chunk), chunks of
// at least 256 rows; one chunk = the sequential k_gmm
static void gmm_chunking(int B, int64_t N, int *nch, int64_t *chunk_len) {
    int64_t n = N / 256;
    const int64_t cap = (1024 / B) > 1 ? (1024 / B) : 1;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    int64_t len = (N + n - 1) / n;
    *nch = (int)((N + len - 1) / len);
    *chunk_len = len;
}

// doubles of `work` that gf_general_matmul needs
int64_t gf_general_matmul_work(int B, int64_t M, int64_t N, int W) {
    int nch;
    int64_t chunk_len;
    if (B < 1 || N < 1 || M < 1 || W < 1) return 0;
    gmm_chunking(B, N, &nch, &chunk_len);
    const int CT = (W <= 64) ? 1 : (W <= 128) ? 2 : 4;
    return (int64_t)B * 2 * M + (nch > 1 ? 3 * (int64_t)B * 2 * nch * CT * 64 : 0);
}

int gf_general_matmul(int B, int64_t M, int64_t N, int W, int ld,
                      const double *c,
                      const double *t1, int64_t t1_bs, const double *U1, const double *V1,
                      const double *t2, int64_t t2_bs, const double *U2, const double *V2,
                      const double *P2, const double *alpha, const int64_t *qidx,
                      double *work, double *mu, void *stream) {
    if (B < 1 || N < 1 || M < 1) return set_err("gf_general_matmul: empty problem (M=%s%lld, N=%lld)", "", M, N);
    if (W < 1 || W > GF_MAX_WIDTH) return set_err("gf_general_matmul: width %s%lld unsupported (max %lld)", "", W, GF_MAX_WIDTH);
    if (ld < W || (ld & 15)) return set_err("gf_general_matmul: ld=%s%lld must be a multiple of 16 and >= W=%lld", "", ld, W);
    if (!c || !t1 || !U1 || !V1 || !t2 || !U2 || !V2 || !P2 || !alpha || !qidx || !work || !mu)
        return set_err("gf_general_matmul: null pointer%s", "");
    GmmArgs A;
    A.M = M; A.N = N; A.W = W; A.ld = ld; A.c = c;
    A.t1 = t1; A.t1_bs = t1_bs; A.U1 = U1; A.V1 = V1;
    A.t2 = t2; A.t2_bs = t2_bs; A.U2 = U2; A.V2 = V2; A.P2 = P2; A.alpha = alpha;
    A.qidx = qidx; A.work = work;
    hipStream_t st = (hipStream_t)stream;
    const int CT = (W <= 64) ? 1 : (W <= 128) ? 2 : 4;
    int nch;
    int64_t chunk_len;
    gmm_chunking(B, N, &nch, &chunk_len);
    if (nch > 1) {                      // long series: chunk-parallel (decayed prefix sums)
        GmmChunk C;
        C.chunk_len = chunk_len; C.nch = nch;
        const size_t per = (size_t)B * 2 * nch * CT * 64;
        C.Floc = work + (size_t)B * 2 * M;
        C.Dloc = C.Floc + per;
        C.Fstart = C.Dloc + per;
        const dim3 grid(B, 2, nch);
        if (CT == 1)      hipLaunchKernelGGL((k_gmm_chunk<1, 0>), grid, dim3(64), 0, st, A, C);
        else if (CT == 2) hipLaunchKernelGGL((k_gmm_chunk<2, 0>), grid, dim3(64), 0, st, A, C);
        else              hipLaunchKernelGGL((k_gmm_chunk<4, 0>), grid, dim3(64), 0, st, A, C);
        hipLaunchKernelGGL(k_gmm_scan, dim3(B, 2), dim3(256), 0, st, CT * 64, C);
        if (CT == 1)      hipLaunchKernelGGL((k_gmm_chunk<1, 1>), grid, dim3(64), 0, st, A, C);
        else if (CT == 2) hipLaunchKernelGGL((k_gmm_chunk<2, 1>), grid, dim3(64), 0, st, A, C);
        else              hipLaunchKernelGGL((k_gmm_chunk<4, 1>), grid, dim3(64), 0, st, A, C);
    }
    else if (W <= 64)  hipLaunchKernelGGL(k_gmm<1>, dim3(B, 2), dim3(64), 0, st, A);
    else if (W <= 128) hipLaunchKernelGGL(k_gmm<2>, dim3(B, 2), dim3(64), 0, st, A);
    else               hipLaunchKernelGGL(k_gmm<4>, dim3(B, 2), dim3(64), 0, st, A);
    hipLaunchKernelGGL(k_add2, dim3((unsigned)((M + 255) / 256), B), dim3(256), 0, st, M, work, mu);
    return check_launch("gf_general_matmul");
}

}  // extern "C"
