// Cycles per instruction of the Gauss-Jordan update's ingredients on one wave (development):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/readlane_cost tools/microbench/readlane_cost.hip && /tmp/readlane_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double read_lane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double *out, long long *cyc, int pv_in) {
    const int lane = threadIdx.x;
    double R[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) R[i] = out[lane * 32 + i];
    double f = out[lane];
    int pv = __builtin_amdgcn_readfirstlane(pv_in);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {            // 32 columns: readlane x2 + fma, batched by 8
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double pr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pr[j] = read_lane(R[8 * b + j], pv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) R[8 * b + j] = fma(-f, pr[j], R[8 * b + j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 1) {     // fma only (vector operands)
#pragma unroll
            for (int i = 0; i < 32; ++i) R[i] = fma(-f, R[(i + 1) & 31], R[i]);
        } else if (MODE == 2) {     // readlanes only (results summed on the scalar side would be optimised: keep as fma every 8)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double pr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pr[j] = read_lane(R[8 * b + j], pv);
                __builtin_amdgcn_sched_barrier(0);
                R[8 * b] = fma(-f, ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])), R[8 * b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {     // LDS broadcast reads + fma
            __shared__ double row[32];
            if (lane == pv) {
#pragma unroll
                for (int i = 0; i < 32; ++i) row[i] = R[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 32; ++i) R[i] = fma(-f, row[i], R[i]);
        }
        pv = (pv + 7) & 63;
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += R[i];
    out[lane] = s;
    if (lane == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 32 * 8); hipMemset(out, 0, 64 * 32 * 8);
    hipMalloc(&cyc, 64);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 5);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, 5);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, 5);
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, 5);
    }
    long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *nm[4] = {"2 readlane + fma (batched 8)", "fma only", "2 readlane (+1/8 fma, 7/8 adds on scalars?)", "LDS row write + broadcast read + fma"};
    for (int m = 0; m < 4; ++m) printf("%-45s %8.1f cycles per column (%lld per 32-column step)\n", nm[m], h[m] / 64.0 / 32.0, h[m] / 64);
    return 0;
}
