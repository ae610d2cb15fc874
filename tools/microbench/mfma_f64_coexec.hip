#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a,b,c) __builtin_amdgcn_mfma_f64_16x16x4f64((a),(b),(c),0,0,0)
// 512-thread workgroups: waves 0-3 and 4-7 pair up on the four SIMDs.  mode bits: 1 = lower waves
// run MFMA, 2 = upper waves run DP FMA, 4 = upper waves run f32 FMA, 8 = upper waves run MFMA
__global__ void __launch_bounds__(512) kco(double *out, int iters, int mode, double a0) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double s = 0;
    if (w < 4 ? (mode & 1) : (mode & 8)) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = a0 + l, b = a0 * 0.5 + l;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = MF(a, b, acc[i]);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (w >= 4 && (mode & 2)) {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = i;
        double a = a0 + l, b = a0 * 0.5;
        for (int it = 0; it < iters * 16; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else if (w >= 4 && (mode & 4)) {
        float acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = i;
        float a = (float)a0 + l, b = (float)a0 * 0.5f;
        for (int it = 0; it < iters * 16; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fmaf(a, acc[i], b);
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
float timeit(int mode, double *out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kco<<<256, 512>>>(out, iters, mode, 1.0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); kco<<<256, 512>>>(out, iters, mode, 1.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; (void)hipMalloc(&out, 8 * 512 * 256);
    const int iters = 4000;   // 16 MFMA per iter ; 256 FMA per iter in the FMA waves
    printf("MFMA alone (1 wave/SIMD)        : %.3f ms\n", timeit(1, out, iters));
    printf("DP FMA alone                    : %.3f ms\n", timeit(2, out, iters));
    printf("f32 FMA alone                   : %.3f ms\n", timeit(4, out, iters));
    printf("MFMA + DP FMA on the same SIMD  : %.3f ms\n", timeit(1 | 2, out, iters));
    printf("MFMA + f32 FMA on the same SIMD : %.3f ms\n", timeit(1 | 4, out, iters));
    printf("MFMA + MFMA on the same SIMD    : %.3f ms\n", timeit(1 | 8, out, iters));
    return 0;
}
