// Micro-benchmark: what does ONE row of a multi-wave recurrence cost in pure exchange on gfx950?
// A workgroup of NW waves iterates: [optional wave reduction] -> LDS write -> workgroup barrier -> LDS read
// -> [optional reciprocal + dependent FMAs] -> next iteration, one workgroup per CU (or two).  No useful
// arithmetic: the time per iteration is the floor under k_factorw's row.
// Build: hipcc -O3 --offload-arch=gfx950 -o row_exchange row_exchange.hip ; run: ./row_exchange
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    v += dpp<0x142>(v); v += dpp<0x143>(v);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NW, int MODE>     // MODE bit 0: wave reduction before the write; bit 1: rcp chain after the read; bit 2: 80 FMAs
__global__ void __launch_bounds__(64 * NW) k_exchange(int iters, double *out) {
    __shared__ double s_v[2][64 * NW];
    __shared__ double s_p[2][NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = 1.0 + 1e-3 * threadIdx.x, acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.1 * k;
    for (int n = 0; n < iters; ++n) {
        const int nxt = (n & 1) ^ 1;
        if (MODE & 4) {
#pragma unroll
            for (int r = 0; r < 10; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fma(acc[k], 0.999999, x);
            x += 1e-9 * (acc[0] + acc[7]);
        }
        double p = x;
        if (MODE & 1) p = wave_sum(x * 1e-3);
        s_v[nxt][threadIdx.x] = x;
        if (lane == 0) s_p[nxt][wave] = p;
        lds_barrier();
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += s_p[nxt][w];
        const double y = s_v[nxt][(threadIdx.x + 64) % (64 * NW)];
        double d = 2.0 + 1e-6 * s;
        if (MODE & 2) {
            double r = __builtin_amdgcn_rcp(d);
            r = fma(fma(-d, r, 1.0), r, r);
            r = fma(fma(-d, r, 1.0), r, r);
            d = r;
        }
        x = fma(y, 1e-6, x * d * ((MODE & 2) ? 2.0 : 0.5));
    }
    out[blockIdx.x * 64 * NW + threadIdx.x] = x + acc[3];
}

template <int NW, int MODE>
void run(const char *what, int wgs, int iters) {
    double *out;
    hipMalloc(&out, sizeof(double) * wgs * 64 * NW);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_exchange<NW, MODE>), dim3(wgs), dim3(64 * NW), 0, 0, 1000, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_exchange<NW, MODE>), dim3(wgs), dim3(64 * NW), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("NW=%d wgs=%4d %-46s %8.3f us per iteration\n", NW, wgs, what, 1e3 * ms / iters);
    hipFree(out);
}

int main() {
    const int it = 200000;
    for (int wgs : {256, 512}) {
        run<4, 0>("write + barrier + read", wgs, it);
        run<4, 1>("wave reduction + write + barrier + read", wgs, it);
        run<4, 2>("write + barrier + read + reciprocal chain", wgs, it);
        run<4, 3>("reduction + write + barrier + read + reciprocal", wgs, it);
        run<4, 7>("80 FMAs + reduction + exchange + reciprocal", wgs, it);
        run<7, 3>("reduction + write + barrier + read + reciprocal", wgs, it);
    }
    return 0;
}
