#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a,b,c) __builtin_amdgcn_mfma_f64_16x16x4f64((a),(b),(c),0,0,0)
template <int NACC>
__global__ void __launch_bounds__(64) kmf(double *out, int iters, double a0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = a0 * 0.5 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MF(a, b, acc[i]);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) kfma(double *out, int iters, double a0) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = a0 + threadIdx.x, b = a0 * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; (void)hipMalloc(&out, 8 * 64 * 4096);
    const int iters = 20000;
    const double clk = 2.4e9;
    for (int waves : {1024, 2048, 4096}) {
        float m1 = timeit([&] { kmf<1><<<waves, 64>>>(out, iters, 1.0); });
        float m4 = timeit([&] { kmf<4><<<waves, 64>>>(out, iters, 1.0); });
        float m16 = timeit([&] { kmf<16><<<waves, 64>>>(out, iters, 1.0); });
        float mf = timeit([&] { kfma<<<waves, 64>>>(out, iters, 1.0); });
        double wps = waves / 1024.0;   // waves per SIMD
        printf("waves/SIMD %.0f: mfma 1acc %.1f cyc/mfma/wave, 4acc %.1f, 16acc %.1f ; per SIMD (16acc) %.1f cyc/mfma ; v_fma_f64 %.2f cyc/instr/wave, per SIMD %.2f\n",
               wps, m1 * 1e-3 * clk / (iters * 4.0), m4 * 1e-3 * clk / (iters * 16.0), m16 * 1e-3 * clk / (iters * 64.0),
               m16 * 1e-3 * clk / (iters * 64.0) / wps, mf * 1e-3 * clk / (iters * 64.0), mf * 1e-3 * clk / (iters * 64.0) / wps);
    }
    return 0;
}
