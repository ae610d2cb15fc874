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

// compile-time loop: f(std::integral_constant<int, K>) for K = 0, 1, ... (register arrays stay in registers)
template <class F, int... K>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, K...>) {
    (f(std::integral_constant<int, K>{}), ...);
}

}  // namespace
