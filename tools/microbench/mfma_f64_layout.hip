#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
// test 1: D = A(16x4) * B(4x16) with documented layouts
__global__ void k1(const double *A, const double *B, double *D) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
// test 2: X (16x16) in D layout regs; compute Y = X^T-as-A * X-as-B per reg: sum_r mfma(X[r], X[r]) should be X^T X
__global__ void k2(const double *X, double *Y) {
    int l = threadIdx.x;
    d4 x;
    for (int r = 0; r < 4; ++r) x[r] = X[((l >> 4) + 4 * r) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[r], x[r], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Y[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], hX[256], hY[256];
    srand(1);
    for (int i = 0; i < 64; ++i) { hA[i] = rand() / (double)RAND_MAX; hB[i] = rand() / (double)RAND_MAX; }
    for (int i = 0; i < 256; ++i) hX[i] = rand() / (double)RAND_MAX;
    double *A, *B, *D, *X, *Y;
    hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&D, 2048); hipMalloc(&X, 2048); hipMalloc(&Y, 2048);
    hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice);
    hipMemcpy(X, hX, 2048, hipMemcpyHostToDevice);
    k1<<<1, 64>>>(A, B, D); k2<<<1, 64>>>(X, Y);
    hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost); hipMemcpy(hY, Y, 2048, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
        e1 = fmax(e1, fabs(s - hD[i * 16 + j]));
        double s2 = 0; for (int k = 0; k < 16; ++k) s2 += hX[k * 16 + i] * hX[k * 16 + j];
        e2 = fmax(e2, fabs(s2 - hY[i * 16 + j]));
    }
    printf("mfma f64 16x16x4 layout: err %.3e ; D-as-A/B duality (X^T X): err %.3e\n", e1, e2);
    return 0;
}
