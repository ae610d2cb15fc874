/*
 * gadfly_hip.h -- C-ABI of libgadfly_hip.so, the MI355X (gfx950) replacement for the
 * celerite2 C++ driver calls that gadfly's GP hot path makes.
 *
 * The reference (/root/reference/gadfly/gp.py) reaches native code only through
 * celerite2.GaussianProcess -> pybind11 module `celerite2.driver` (third-party, NOT in
 * /root/reference).  Each entry point below names the driver function
 * it replaces and the gadfly call site that reaches it:
 *
 *   gf_build_matrices   driver.get_celerite_matrices   <- gp.py:202 (compute)
 *   gf_factor           driver.factor                  <- gp.py:202 (compute)
 *   gf_loglike_reduce   numpy glue in celerite2.core   <- gp.py:350 (_norm - 0.5 sum z^2/d)
 *   gf_solve            driver.solve_lower / solve_upper / matmul_lower
 *                                                      <- gp.py:350, :370, :232, :327, :391
 *   gf_general_matmul   driver.general_matmul_lower + general_matmul_upper
 *                                                      <- gp.py:232 (predict at new times)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP) unless marked "host"; float64 throughout;
 *   - row-major; the N x W generator matrices U, V, W and the propagator P use a leading
 *     dimension `ld` >= W (ld is a multiple of 16, pad columns hold 0 for U/V/W and 1 for P)
 *     so rows start 128-byte aligned; batch stride is N*ld;
 *   - B independent problems per call (light curves or MCMC walkers); per-array
 *     batch strides are in elements, 0 = shared by all problems;
 *   - the caller owns every buffer (torch tensors on the Python side); the library allocates
 *     nothing and keeps no global state except the thread-local last-error string;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work;
 *   - return value: 0 ok, < 0 bad arguments / launch error (see gf_last_error()).
 *     Numerical failure (non-positive pivot, celerite2's LinAlgError) is reported per problem
 *     in the device array `info` (0 or the 1-based failing row), never by the return value.
 */
#ifndef GADFLY_HIP_H
#define GADFLY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_MAX_WIDTH 256        /* largest supported celerite width W (= 2 J for gadfly) */

/* gf_solve modes */
#define GF_SOLVE_LOWER   0      /* Z = L^-1 Y          driver.solve_lower  */
#define GF_SOLVE_UPPER   1      /* Z = L^-T Y          driver.solve_upper  */
#define GF_MATMUL_LOWER  2      /* Z = L Y             driver.matmul_lower (with V := W) */

int gf_version(void);
const char *gf_last_error(void);

/* leading dimension the library wants for width W (multiple of 16); <0 if W unsupported */
int gf_leading_dim(int W);

/*
 * Matrix build + propagator rows.
 *   coefficients: ar, cr [B][Jr]; ac, bc, cc, dc [B][Jc]   (W = Jr + 2 Jc)
 *   diag_add [B]  : sum(ar) + sum(ac) + TermConvolution diagonal shift, added to diag
 *   t    [B|1][N] : times, batch stride t_bs (0 = shared)
 *   diag [B|1][N] : user diagonal (yerr^2), batch stride diag_bs; NULL = 0
 * outputs
 *   a [B][N];  U, V, P [B][N][ld];  P[n][j] = exp(c_j (t[n-1] - t[n])), P[0][:] = 1
 *   (P may be NULL when only U, V are wanted, e.g. at prediction times)
 */
int gf_build_matrices(int B, int64_t N, int Jr, int Jc, int ld,
                      const double *ar, const double *cr, const double *ac,
                      const double *bc, const double *cc, const double *dc,
                      const double *diag_add,
                      const double *t, int64_t t_bs,
                      const double *diag, int64_t diag_bs,
                      double *a, double *U, double *V, double *P, void *stream);

/*
 * Semiseparable LDL^T factor, optionally fused with the forward solve of one
 * right-hand side (the log-likelihood path).
 *   inputs : a [B][N]; U, V, P [B][N][ld]; y [B|1][N] (batch stride y_bs) or NULL
 *   outputs: d [B][N]; Wm [B][N][ld] or NULL (not stored); z [B][N] (= L^-1 y) or NULL
 *            info [B] (0 ok, else 1-based row of the first non-positive pivot)
 */
int gf_factor(int B, int64_t N, int W, int ld,
              const double *a, const double *U, const double *V, const double *P,
              const double *y, int64_t y_bs,
              double *d, double *Wm, double *z, int32_t *info, void *stream);

/*
 * out[b] = -0.5 (sum log d + N log 2pi) - 0.5 sum z^2/d ;  logdet[b] = sum log d  (NULL ok)
 * (z == NULL: out[b] = the normalisation constant only).  If info[b] != 0: out = -inf,
 * logdet = -inf.  work must hold B * gf_reduce_work(N) doubles.
 */
int64_t gf_reduce_work(int64_t N);
int gf_loglike_reduce(int B, int64_t N, const double *d, const double *z,
                      const int32_t *info, double *work, double *out, double *logdet,
                      void *stream);

/*
 * Triangular sweeps with R right-hand sides, Y and Z are [B][N][R] row-major (Z may alias Y
 * for the two solves, not for GF_MATMUL_LOWER):
 *   GF_SOLVE_LOWER : F <- P_n (F + W_{n-1} Z_{n-1}),  Z_n = Y_n - U_n F          (n ascending)
 *   GF_SOLVE_UPPER : F <- P_{n+1} (F + U_{n+1} Z_{n+1}), Z_n = Y_n - W_n F       (n descending)
 *   GF_MATMUL_LOWER: F <- P_n (F + W_{n-1} Y_{n-1}),  Z_n = Y_n + U_n F          (n ascending)
 * `scale` [B][N] or NULL: if given, the input row is first multiplied by
 *   1/scale[n] (solve modes: fuses apply_inverse's division by d) or sqrt(scale[n])
 *   (GF_MATMUL_LOWER: dot_tril's sqrt(d) factor).
 */
int gf_solve(int mode, int B, int64_t N, int W, int ld, int R,
             const double *U, const double *Wm, const double *P, const double *scale,
             const double *Y, double *Z, void *stream);

/*
 * Conditional mean at M new (sorted) times t1 given alpha = K^-1 (y - mean) at the N
 * observed times t2:
 *   mu[m] = sum_{t2[n] <= t1[m]} (U1[m] o e^{-c (t1[m]-t2[n])}) . V2[n] alpha[n]
 *         + sum_{t2[n] >  t1[m]} (V1[m] o e^{-c (t2[n]-t1[m])}) . U2[n] alpha[n]
 *   c [B][W]; U1, V1 [B][M][ld]; U2, V2, P2 [B][N][ld]; t1 [B|1][M]; t2 [B|1][N]
 *   work: B * 2 * M doubles.
 */
int gf_general_matmul(int B, int64_t M, int64_t N, int W, int ld,
                      const double *c,
                      const double *t1, int64_t t1_bs, const double *U1, const double *V1,
                      const double *t2, int64_t t2_bs, const double *U2, const double *V2,
                      const double *P2, const double *alpha,
                      double *work, double *mu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GADFLY_HIP_H */
