// Wave-level primitives (64 lanes, DPP / readlane: no LDS traffic) shared by the library's translation units.
// Internal linkage: every translation unit gets its own inlined copies.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double read_lane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// max over the 64 lanes (same DPP tree), broadcast
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_get<0xB1, 0xf>(v));
    v = fmax(v, dpp_get<0x4E, 0xf>(v));
    v = fmax(v, dpp_get<0x141, 0xf>(v));
    v = fmax(v, dpp_get<0x140, 0xf>(v));
    double r1 = read_lane(v, 15), r2 = read_lane(v, 31), r3 = read_lane(v, 47), r4 = read_lane(v, 63);
    return fmax(fmax(r1, r2), fmax(r3, r4));
}

// 1/x for normal positive x: hardware estimate + two Newton steps (error <= ~1 ulp); the
// IEEE division sequence costs ~14 VALU instructions on a path every row waits for.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// max of a 32-bit key over the 64 lanes, broadcast (DPP inside the rows of 16 lanes, scalar across them)
__device__ __forceinline__ int wave_max_key(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));    // row_mirror
    const int r1 = __builtin_amdgcn_readlane(v, 15), r2 = __builtin_amdgcn_readlane(v, 31),
              r3 = __builtin_amdgcn_readlane(v, 47), r4 = __builtin_amdgcn_readlane(v, 63);
    return max(max(r1, r2), max(r3, r4));
}

// Gauss(-Jordan) elimination with lanes = rows (gadfly_hip.hip, cb_gauss_jordan): the owner's part of one step: pivot row among the lanes with !used for the column entries cval;
// multipliers into fb[lane], pivot lane and 1 / pivot into pvb / pinvb
__device__ __forceinline__ void gj_search(const double cval, const bool used, const int lane, double *fb, int *pvb,
                                          double *pinvb) {
    const double rc = fast_rcp(cval);                               // (of the lane's own entry: off the chain)
    const int key = used ? -1 : (__double2hiint(cval) & 0x7fffffff);
    const int vm = wave_max_key(key);
    const unsigned long long hit = __ballot(key == vm);
    const int pv = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)hit) - 1);
    const double pinv = read_lane(rc, pv);
    fb[lane] = (lane == pv) ? 0.0 : cval * pinv;
    if (lane == 0) { *pvb = pv; *pinvb = pinv; }
}

// compile-time loop: f(std::integral_constant<int, K>) for K = 0, 1, ... (register arrays stay in registers)
template <class F, int... K>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
}

// R[lc] -= f * (R[lc] of lane pv) for the columns lc = L0 .. NC - 1 except SKIP.  The pivot row's entries come by
// readlane into SGPR pairs, GJ_BATCH columns at a time and then their FMAs: one column at a time the compiler reuses
// a single SGPR pair, and every FMA waits for its own two readlanes (~30 cycles per column).
constexpr int GJ_BATCH = 8;
template <int L0, int NC, int SKIP, int N>
__device__ __forceinline__ void gj_update(double (&R)[N], const double f, const int pv) {
    static_for([&](auto bc) {
        constexpr int l0 = L0 + GJ_BATCH * decltype(bc)::value;
        auto live = [](int lc) { return lc < NC && lc != SKIP; };
        double pr[GJ_BATCH];
#pragma unroll
        for (int j = 0; j < GJ_BATCH; ++j)
            if (live(l0 + j)) pr[j] = read_lane(R[l0 + j], pv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GJ_BATCH; ++j)
            if (live(l0 + j)) R[l0 + j] = fma(-f, pr[j], R[l0 + j]);
        __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, (NC - L0 + GJ_BATCH - 1) / GJ_BATCH>{});
}

}  // namespace
