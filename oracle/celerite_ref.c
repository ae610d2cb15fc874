/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, loaded or called by the product
 * path (gadfly_amd); only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may use it.
 *
 * Plain-C restatement of the celerite2 C++ driver routines that gadfly's GP path
 * reaches through celerite2.GaussianProcess:
 *   get_celerite_matrices + factor    <- /root/reference/gadfly/gp.py:202   (compute)
 *   solve_lower + norm                <- /root/reference/gadfly/gp.py:350   (log_likelihood)
 *   solve_lower / solve_upper         <- /root/reference/gadfly/gp.py:370, :232 (apply_inverse, predict)
 *   matmul_lower                      <- /root/reference/gadfly/gp.py:327, :391 (dot_tril, sample)
 *   general_matmul_lower/upper        <- /root/reference/gadfly/gp.py:232   (predict at new times)
 * celerite2 (PyPI, unpinned at /root/reference/pyproject.toml:20; C++/Eigen core) is NOT in
 * /root/reference and not installed here, so this follows its published algorithm
 * (Foreman-Mackey et al. 2017; Foreman-Mackey 2018) as restated in SURVEY.md App. A.4-A.8.
 *
 * PARITY UNPINNED at the reference level (no numerical golden for this path in
 * /root/reference); pinned instead against oracle/dense.py (independent O(N^3) Cholesky)
 * and an 80-bit run of oracle/seq.py -- see tests/test_oracle.py and tests/golden/.
 *
 * Layout: row-major, U/V/W are N x J with leading dimension J (celerite2's layout).
 * Single-threaded, like celerite2's driver.  Build: oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* SURVEY.md A.4 */
void ref_get_matrices(int64_t N, int Jr, int Jc,
                      const double *ar, const double *cr, const double *ac,
                      const double *bc, const double *cc, const double *dc,
                      const double *x, const double *diag,
                      double *c, double *a, double *U, double *V)
{
    const int J = Jr + 2 * Jc;
    double asum = 0.0;
    for (int j = 0; j < Jr; ++j) { asum += ar[j]; c[j] = cr[j]; }
    for (int k = 0; k < Jc; ++k) { c[Jr + 2 * k] = cc[k]; c[Jr + 2 * k + 1] = cc[k]; }
    double acsum = 0.0;
    for (int k = 0; k < Jc; ++k) acsum += ac[k];
    asum += acsum;
    for (int64_t n = 0; n < N; ++n) {
        double *u = U + n * J, *v = V + n * J;
        a[n] = diag[n] + asum;
        for (int j = 0; j < Jr; ++j) { u[j] = ar[j]; v[j] = 1.0; }
        for (int k = 0; k < Jc; ++k) {
            const double arg = dc[k] * x[n];          /* one rounded multiply */
            const double co = cos(arg), si = sin(arg);
            u[Jr + 2 * k]     = ac[k] * co + bc[k] * si;
            u[Jr + 2 * k + 1] = ac[k] * si - bc[k] * co;
            v[Jr + 2 * k]     = co;
            v[Jr + 2 * k + 1] = si;
        }
    }
}

/* SURVEY.md A.5.  d (N) and W (N x J) are outputs.  Returns 0 or the 1-based failing row. */
int64_t ref_factor(int64_t N, int J, const double *t, const double *c, const double *a,
                   const double *U, const double *V, double *d, double *W)
{
    double *S = (double *)calloc((size_t)J * J, sizeof(double));
    double *p = (double *)malloc(sizeof(double) * J);
    double *tmp = (double *)malloc(sizeof(double) * J);
    int64_t info = 0;
    d[0] = a[0];
    if (!(d[0] > 0.0)) { info = 1; goto done; }
    for (int j = 0; j < J; ++j) W[j] = V[j] / d[0];
    for (int64_t n = 1; n < N; ++n) {
        const double dt = t[n - 1] - t[n];
        const double *wp = W + (n - 1) * J, *u = U + n * J, *v = V + n * J;
        double *w = W + n * J;
        const double dp = d[n - 1];
        for (int j = 0; j < J; ++j) p[j] = exp(c[j] * dt);
        for (int i = 0; i < J; ++i) {
            const double wi = dp * wp[i], pi = p[i];
            double *Si = S + (size_t)i * J;
            for (int j = 0; j < J; ++j) Si[j] = pi * p[j] * (Si[j] + wi * wp[j]);
        }
        for (int j = 0; j < J; ++j) tmp[j] = 0.0;
        for (int i = 0; i < J; ++i) {
            const double ui = u[i];
            const double *Si = S + (size_t)i * J;
            for (int j = 0; j < J; ++j) tmp[j] += ui * Si[j];
        }
        double dn = a[n];
        for (int j = 0; j < J; ++j) dn -= tmp[j] * u[j];
        d[n] = dn;
        if (!(dn > 0.0)) { info = n + 1; goto done; }
        for (int j = 0; j < J; ++j) w[j] = (v[j] - tmp[j]) / dn;
    }
done:
    free(S); free(p); free(tmp);
    return info;
}

/* SURVEY.md A.6.  Y, Z are N x R row-major; Z may alias Y. */
void ref_solve_lower(int64_t N, int J, int R, const double *t, const double *c,
                     const double *U, const double *W, const double *Y, double *Z)
{
    double *F = (double *)calloc((size_t)J * R, sizeof(double));
    if (Z != Y) memcpy(Z, Y, sizeof(double) * N * R);
    for (int64_t n = 1; n < N; ++n) {
        const double dt = t[n - 1] - t[n];
        const double *wp = W + (n - 1) * J, *u = U + n * J, *zp = Z + (n - 1) * R;
        double *z = Z + n * R;
        for (int j = 0; j < J; ++j) {
            const double pj = exp(c[j] * dt), wj = wp[j];
            double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) Fj[r] = pj * (Fj[r] + wj * zp[r]);
        }
        for (int j = 0; j < J; ++j) {
            const double uj = u[j];
            const double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) z[r] -= uj * Fj[r];
        }
    }
    free(F);
}

void ref_solve_upper(int64_t N, int J, int R, const double *t, const double *c,
                     const double *U, const double *W, const double *Y, double *Z)
{
    double *F = (double *)calloc((size_t)J * R, sizeof(double));
    if (Z != Y) memcpy(Z, Y, sizeof(double) * N * R);
    for (int64_t n = N - 2; n >= 0; --n) {
        const double dt = t[n] - t[n + 1];
        const double *un = U + (n + 1) * J, *w = W + n * J, *zn = Z + (n + 1) * R;
        double *z = Z + n * R;
        for (int j = 0; j < J; ++j) {
            const double pj = exp(c[j] * dt), uj = un[j];
            double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) Fj[r] = pj * (Fj[r] + uj * zn[r]);
        }
        for (int j = 0; j < J; ++j) {
            const double wj = w[j];
            const double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) z[r] -= wj * Fj[r];
        }
    }
    free(F);
}

/* Z = Y + tril(U V^T o Phi, -1) Y; Z must not alias Y. */
void ref_matmul_lower(int64_t N, int J, int R, const double *t, const double *c,
                      const double *U, const double *V, const double *Y, double *Z)
{
    double *F = (double *)calloc((size_t)J * R, sizeof(double));
    memcpy(Z, Y, sizeof(double) * N * R);
    for (int64_t n = 1; n < N; ++n) {
        const double dt = t[n - 1] - t[n];
        const double *vp = V + (n - 1) * J, *u = U + n * J, *yp = Y + (n - 1) * R;
        double *z = Z + n * R;
        for (int j = 0; j < J; ++j) {
            const double pj = exp(c[j] * dt), vj = vp[j];
            double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) Fj[r] = pj * (Fj[r] + vj * yp[r]);
        }
        for (int j = 0; j < J; ++j) {
            const double uj = u[j];
            const double *Fj = F + (size_t)j * R;
            for (int r = 0; r < R; ++r) z[r] += uj * Fj[r];
        }
    }
    free(F);
}

/* Z[m] += sum_{t2[n] <= t1[m]} (U1[m] o e^{-c (t1[m]-t2[n])}) . V2[n] Y[n]  (R = 1) */
void ref_general_matmul_lower(int64_t M, int64_t N, int J, const double *t1, const double *t2,
                              const double *c, const double *U1, const double *V2,
                              const double *Y, double *Z)
{
    double *F = (double *)calloc((size_t)J, sizeof(double));
    int64_t n = 0;
    int have = 0;
    double last = 0.0;
    for (int64_t m = 0; m < M; ++m) {
        while (n < N && t2[n] <= t1[m]) {
            if (have) {
                const double dt = last - t2[n];
                for (int j = 0; j < J; ++j) F[j] *= exp(c[j] * dt);
            }
            for (int j = 0; j < J; ++j) F[j] += V2[n * J + j] * Y[n];
            last = t2[n]; have = 1; ++n;
        }
        if (have) {
            const double dt = last - t1[m];
            double acc = 0.0;
            for (int j = 0; j < J; ++j) acc += U1[m * J + j] * exp(c[j] * dt) * F[j];
            Z[m] += acc;
        }
    }
    free(F);
}

/* Z[m] += sum_{t2[n] > t1[m]} (V1[m] o e^{-c (t2[n]-t1[m])}) . U2[n] Y[n]  (R = 1) */
void ref_general_matmul_upper(int64_t M, int64_t N, int J, const double *t1, const double *t2,
                              const double *c, const double *V1, const double *U2,
                              const double *Y, double *Z)
{
    double *F = (double *)calloc((size_t)J, sizeof(double));
    int64_t n = N - 1;
    int have = 0;
    double last = 0.0;
    for (int64_t m = M - 1; m >= 0; --m) {
        while (n >= 0 && t2[n] > t1[m]) {
            if (have) {
                const double dt = t2[n] - last;
                for (int j = 0; j < J; ++j) F[j] *= exp(c[j] * dt);
            }
            for (int j = 0; j < J; ++j) F[j] += U2[n * J + j] * Y[n];
            last = t2[n]; have = 1; --n;
        }
        if (have) {
            const double dt = t1[m] - last;
            double acc = 0.0;
            for (int j = 0; j < J; ++j) acc += V1[m * J + j] * exp(c[j] * dt) * F[j];
            Z[m] += acc;
        }
    }
    free(F);
}

/*
 * One "log-likelihood evaluation" as the BASELINE metric defines it:
 * matrix build + factor (compute) + solve_lower + reductions (log_likelihood).
 * work must hold N*(3J+3) + J doubles.  Returns 0 or the failing row; *out = loglike.
 */
int64_t ref_loglike(int64_t N, int Jr, int Jc,
                    const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc,
                    const double *t, const double *diag, const double *y,
                    double *work, double *out)
{
    const int J = Jr + 2 * Jc;
    double *c = work, *a = c + J, *U = a + N, *V = U + N * J, *W = V + N * J;
    double *d = W + N * J, *z = d + N;
    ref_get_matrices(N, Jr, Jc, ar, cr, ac, bc, cc, dc, t, diag, c, a, U, V);
    const int64_t info = ref_factor(N, J, t, c, a, U, V, d, W);
    if (info) { *out = -INFINITY; return info; }
    ref_solve_lower(N, J, 1, t, c, U, W, y, z);
    double logdet = 0.0, quad = 0.0;
    for (int64_t n = 0; n < N; ++n) { logdet += log(d[n]); quad += z[n] * z[n] / d[n]; }
    *out = -0.5 * (logdet + (double)N * log(2.0 * M_PI)) - 0.5 * quad;
    return 0;
}
