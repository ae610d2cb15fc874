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
 *   gf_reduce_tile /
 *   gf_loglike_finish   numpy glue in celerite2.core   <- gp.py:350 (_norm - 0.5 sum z^2/d)
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
 *     nothing and keeps no mutable global state: every option is a call argument (the only statics
 *     are the thread-local last-error string and a per-device "large-LDS attribute applied" flag);
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

/* `variant` of gf_loglike_fused / gf_chunk_sweep / gf_chunk_transition: which fused sweep runs.
 * Both agree to rounding and are parity-tested; callers pass GF_SWEEP_AUTO. */
#define GF_SWEEP_AUTO    0      /* lane-tiled sweep where the term structure allows it, else by column */
#define GF_SWEEP_COLUMN  1      /* k_factor3 / k_phi: one state column per lane, any mix of terms */
#define GF_SWEEP_TILED   2      /* k_factor7 / k_phi7: 2 x 32 lane tiling; needs Jr = 0, Jc <= 31 */
/* OR-ed into gf_chunk_sweep's `variant`: every chunk of the call starts from a ZERO state (a nominal pass);
 * the S_state / F_state slots are then outputs only and need not be initialised. */
#define GF_SWEEP_ZERO_START 0x100

int gf_version(void);
const char *gf_last_error(void);

/* leading dimension the library wants for width W (multiple of 16); <0 if W unsupported */
int gf_leading_dim(int W);

/*
 * Matrix build + propagator rows, for the N rows n_first .. n_first+N-1 of a
 * longer series (tile streaming; n_first = 0 and N = series length for a whole series).
 *   coefficients: ar, cr [B][Jr]; ac, bc, cc, dc [B][Jc]   (W = Jr + 2 Jc)
 *   diag_add [B]  : sum(ar) + sum(ac) + TermConvolution diagonal shift, added to diag
 *   t    [B|1][*] : times of the WHOLE series (indexed with the global row), batch stride t_bs
 *   diag [B|1][*] : user diagonal (yerr^2) of the whole series, batch stride diag_bs; NULL = 0
 * outputs (tile-local row index)
 *   a [B][N] (NULL ok);  U, V, P [B][N][ld];  P[n][j] = exp(c_j (t[g-1] - t[g])), g = n_first+n,
 *   P row of global row 0 = 1  (P may be NULL when only U, V are wanted, e.g. at prediction times)
 */
int gf_build_matrices(int B, int64_t N, int64_t n_first, int Jr, int Jc, int ld,
                      const double *ar, const double *cr, const double *ac,
                      const double *bc, const double *cc, const double *dc,
                      const double *diag_add,
                      const double *t, int64_t t_bs,
                      const double *diag, int64_t diag_bs,
                      double *a, double *U, double *V, double *P, void *stream);

/*
 * Semiseparable LDL^T factor of N consecutive rows, optionally fused with the
 * forward solve of one right-hand side (the log-likelihood path).
 *   inputs : a [B][N]; U, V, P [B][N][ld]; y [B|1][N] (pointer to the tile's first row,
 *            batch stride y_bs) or NULL
 *   outputs: d [B][N]; Wm [B][N][ld] or NULL (not stored); z [B][N] (= L^-1 y) or NULL
 *   state  : S_state [B][gf_state_size(W)], F_state [B][gf_state_cols(W)] or NULL.
 *            When given, the recurrence state (S + d w w^T and F + w z of the last row, before
 *            the next propagator is applied) is READ at entry and WRITTEN at exit, so a series
 *            can be streamed tile by tile through fixed-size U/V/P buffers; zero both before
 *            the first tile.  NULL = start from zero, final state not stored.
 *   info [B]: must be zeroed by the caller before the first tile; the kernel only ever writes
 *            a non-zero value (1-based GLOBAL row n_first+n+1 of the first non-positive pivot),
 *            and a problem whose info is already non-zero is skipped.
 */
int64_t gf_state_size(int W);
int gf_state_cols(int W);
int gf_factor(int B, int64_t N, int64_t n_first, int W, int ld,
              const double *a, const double *U, const double *V, const double *P,
              const double *y, int64_t y_bs,
              double *d, double *Wm, double *z,
              double *S_state, double *F_state, int32_t *info, void *stream);

/*
 * Log-likelihood fast path for W <= 64 in block-scaled coordinates (see the derivation above
 * k_build2 in gadfly_hip.hip).  Same role as gf_build_matrices + gf_factor on the
 * log-likelihood path (gp.py:202 + gp.py:350); U~, V~ are u o rho and v / rho with
 * rho = exp(-c (t_n - t_ref)); de[n] = t_ref(n) - t_ref(n-1) >= 0 at reset rows (where the
 * accumulated decay exp(-c de) is applied once) and -1 elsewhere.
 *   cmax [B] : max_j c_j of each problem; block: rows between forced resets, a power of two in
 *   1..64 chosen by the caller so that (block - 1) * cmax * cadence stays well below 28 (a row
 *   whose cmax * dt exceeds 28/(block-1) resets on its own); c [B][W] decay rate per column;
 *   n_first must be a multiple of block.
 *   outputs of gf_build_scaled: a, de [B][N]; Ut, Vt [B][N][ld]
 *   gf_factor_scaled: y (tile pointer), d, z, S_state [B][64*64], F_state [B][64], info as in
 *   gf_factor (state mandatory; zero it and info before the first tile).  Wm is not produced.
 *   Widths 64 < W <= 256 (gf_scaled_wide_supported): the same recurrence on several waves per
 *   problem (k_factor2w: two FMAs per state element and row instead of gf_factor's four operations);
 *   S_state [B][gf_state_size(W)], F_state [B][gf_state_cols(W)] as for gf_factor.
 *   The sweep prefetches rows n+1, n+2 unconditionally: a, de, y, Ut, Vt must each be readable
 *   two rows (elements) past the last row of the last problem.
 */
int gf_scaled_supported(int W);
int gf_scaled_wide_supported(int W);
int gf_build_scaled(int B, int64_t N, int64_t n_first, int Jr, int Jc, int ld,
                    const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc,
                    const double *diag_add, const double *cmax, int block,
                    const double *t, int64_t t_bs,
                    const double *diag, int64_t diag_bs,
                    double *a, double *Ut, double *Vt, double *de, void *stream);
int gf_factor_scaled(int B, int64_t N, int64_t n_first, int W, int ld, const double *c,
                     const double *a, const double *Ut, const double *Vt, const double *de,
                     const double *y, int64_t y_bs,
                     double *d, double *z, double *S_state, double *F_state,
                     int32_t *info, void *stream);

/*
 * Fully fused log-likelihood sweep (W <= 63): matrix build + factor + forward solve of N
 * consecutive rows in one kernel, generator rows produced in registers (nothing but t, y, diag
 * is read: 24 B/row).  Replaces gf_build_scaled + gf_factor_scaled on the log-likelihood path
 * (gp.py:202 + gp.py:350) whenever max|d_c| * max|t| < 1e12 (the range of the kernel's
 * FMA-reduced sincos; the caller checks).  Arguments as in gf_build_scaled / gf_factor_scaled;
 * t, diag (NULL = 0), y are the WHOLE series (global row index, batch strides t_bs, diag_bs,
 * y_bs) and must be readable three elements past row n_first + N - 1.
 *
 * gen_period: accuracy / speed of the in-register generator rows (also of gf_chunk_sweep and
 * gf_chunk_transition).  Between exact anchors the rows advance by a cached complex rotation; every
 * `gen_period` rows (a power of two, 1..64) the phasor is recomputed exactly.  Measured at a
 * condition (diagonal / pivot) of 4e5 against an 80-bit recurrence, per 8192-row tile of 2048
 * evaluations: 1 (exact rows) 2e-10 / 12.2 ms, 4: 2e-9 / 10.4 ms, 16: 1e-8 / 10.0 ms, 64: several
 * 1e-8 / 10.0 ms, i.e. error ~ 1.6e-15 * gen_period * condition.  Callers pick it from the condition
 * estimate max(a) / min(d) that gf_reduce_tile returns (1 when in doubt).  Irregular spacings are
 * always generated exactly.  variant: GF_SWEEP_AUTO (see the GF_SWEEP_* constants).
 *
 * Wide kernels: for kernels made of complex terms only (Jr = 0: every SHO term with Q >= 1/2) the
 * same entry points take 64 <= W <= 176 (k_factorw: one workgroup of ceil((W + 1) / 32) + 1 waves per
 * problem; `variant` is ignored).  gf_fused_supported tells; S_state then holds
 * gf_fused_state_size(Jr, Jc) doubles per (problem, chunk) slot instead of 64 * 64 (zeroed by the
 * caller before the first tile; opaque layout) and F_state is not used (may be NULL).
 */
int gf_fused_supported(int Jr, int Jc);
int64_t gf_fused_state_size(int Jr, int Jc);
int gf_loglike_fused(int B, int64_t N, int64_t n_first, int Jr, int Jc, int block,
                     int gen_period, int variant,
                     const double *ar, const double *cr, const double *ac,
                     const double *bc, const double *cc, const double *dc,
                     const double *diag_add, const double *cmax,
                     const double *t, int64_t t_bs, const double *diag, int64_t diag_bs,
                     const double *y, int64_t y_bs,
                     double *d, double *z, double *S_state, double *F_state,
                     int32_t *info, void *stream);

/*
 * Exact time-parallel evaluation of ONE long series (or a few): the N rows are cut into nch
 * chunks of chunk_len rows (a multiple of `block`; the last chunk may be shorter) that are swept
 * concurrently, then stitched with an exact linear-fractional combine:
 *   1. gf_chunk_sweep      nominal pass: every chunk starts from the zero state -- S_state/F_state
 *                          [B*nch] zeroed by the caller, or variant | GF_SWEEP_ZERO_START (the slots
 *                          are then outputs only); outputs dbar, zbar [B][N], rbar [B][N][64]
 *                          (pass r_out), the rows u~ [B][N][64] and reset spans de [B][N] (pass Ut_out,
 *                          de_out; Wt_out may stay NULL) and the nominal end states in S_state/F_state.
 *   2. gf_chunk_transition closed-loop transition Phi [B*nch][64*64] and the Gram sums G [B*nch][64*64],
 *                          m [B*nch][64] of the chunks chunk_first .. chunk_first + chunk_count - 1 of
 *                          every problem, from those rows (c [B][W]: the columns' decay rates).  The
 *                          start states never depend on the last chunk's map, and on the first chunk's
 *                          only through its end state: callers sweep the chunks 1 .. nch - 2 and hand
 *                          the combine zeros for the first chunk's Phi, G, m.
 *   3. gf_chunk_combine    sequential LFT combine over the chunks (64x64 pivoted solves in LDS):
 *                          S_state/F_state slot c <- TRUE start state of chunk c.
 *      gf_chunk_combine_tree  the same result in 2 (log2(P) - 1) levels (Blelloch scan over the chunk
 *                          maps) on the same arrays [B*nch] (Phi, G, m, S = nominal end X,
 *                          F = nominal end Y); P = the power of two with P / 2 < nch <= P.  The map
 *                          arrays are overwritten; Xst [B*nch][64*64] / Yst [B*nch][64] slot c <- TRUE
 *                          start state of chunk c; work: gf_chunk_combine_tree_work(B, P, nch) doubles
 *                          (the scan's slots beyond nch).
 *   4. gf_chunk_sweep      final pass from those states (r_out = NULL): d, z equal the
 *                          sequential result to rounding; reduce with gf_reduce_tile.  With
 *                          Ut_out, Wt_out [B][N][64] and de_out [B][N] the pass also stores the
 *                          factor in scaled form (rows u~, w~ = r/d and the reset spans) for
 *                          gf_chunk_linear.
 * Same argument conventions (gen_period and variant included: pass the SAME values to all
 * calls of one evaluation) and padding rules as gf_loglike_fused; the row arrays gf_chunk_transition
 * reads (Ut, rbar, dbar, zbar, de) must be readable EIGHT rows past the end (they are fetched ahead by
 * LDS-DMA).  Width 1..63, phases |d t| < 1e12.  gf_chunk_sweep works on the chunks chunk_first ..
 * chunk_first + chunk_count - 1 of every problem (state slots and rows of the others are left alone): the
 * nominal pass leaves out the last chunk, a final pass that stores no factor the first one (whose nominal
 * pass was exact).  A slot whose info entry is non-zero on entry is skipped as well.
 */
int gf_chunk_sweep(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count,
                   int Jr, int Jc, int block, int gen_period, int variant,
                   const double *ar, const double *cr, const double *ac,
                   const double *bc, const double *cc, const double *dc,
                   const double *diag_add, const double *cmax,
                   const double *t, int64_t t_bs, const double *diag, int64_t diag_bs,
                   const double *y, int64_t y_bs,
                   double *d, double *z, double *r_out, double *Ut_out, double *Wt_out,
                   double *de_out, double *S_state, double *F_state,
                   int32_t *info, void *stream);
int gf_chunk_transition(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count,
                        int Jr, int Jc, int variant, const double *c, const double *de, const double *dbar,
                        const double *zbar, const double *rbar, const double *Ut,
                        double *Phi_out, double *G_out, double *m_out, void *stream);
/*
 * The stored rows as a factor for gf_solve (any width the fused sweeps take, in particular the wide
 * kernels, which have no chunk-parallel sweeps): gf_fused_row_stride is the leading dimension of
 * Ut_out / Wt_out / r_out (64, or the padded column count of the wide sweep), and
 * gf_scaled_propagator expands the reset spans de [B][N] into the propagator rows P [B][N][ld] of
 * the scaled coordinates (1, or exp(-c de) at a reset row; c [B][W]).  gf_solve(U = Ut, Wm = Wt,
 * P, scale = d) then performs solve_lower / solve_upper / matmul_lower of the TRUE factor
 * (driver.solve_lower etc. <- gp.py:350, :370, :327).
 */
int gf_fused_row_stride(int Jr, int Jc);
int gf_scaled_propagator(int B, int64_t N, int W, int ld, const double *c, const double *de,
                         double *P_out, void *stream);
/*
 * Wide kernels (64 <= W <= 176, complex terms): the closed-loop transition sweep of step 2 on the rows
 * the nominal pass stored -- Ut [B][N][ld] (u~ rows), rbar [B][N][ld], dbar [B][N], de [B][N] (reset
 * spans), ld = gf_fused_row_stride; c [B][W] decay rates; Ut, rbar, dbar and de readable EIGHT rows
 * past the end (the rows are fetched ahead by LDS-DMA, unconditionally).  Outputs the rows h [B][N][ld]
 * and Phi [B*nch][gf_fused_state_size] in the layout of S_state ([column][row], padded; columns from
 * 16 ceil(W / 16) on are left unwritten in both).  The Gram sums
 * G = sum h h^T / dbar, m = sum h zbar / dbar and the combine of the W x W chunk maps follow in
 * gf_wide_combine.  Only the chunks chunk_first .. chunk_first + chunk_count - 1 of every problem are swept.
 */
int gf_chunk_transition_wide(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count, int Jc,
                             const double *c, const double *de, const double *dbar, const double *rbar,
                             const double *Ut, double *h_out, double *Phi_out, void *stream);
/* (W = Jr + 2 Jc <= 63, the width of the states: the 64 x 64 slots are zero beyond it, and the combines'
 * pivoted solves and matrix products are carried out on the leading W rows and columns only) */
int gf_chunk_combine(int B, int nch, int W, const double *Phi, const double *G, const double *m,
                     double *S_state, double *F_state, void *stream);
int64_t gf_chunk_combine_tree_work(int B, int P, int nch);
int gf_chunk_combine_tree(int B, int P, int nch, int W, double *Phi, double *G, double *m, double *S, double *F,
                          double *Xst, double *Yst, double *work, void *stream);

/*
 * Triangular sweeps on the stored scaled factor, time-parallel (same modes as gf_solve):
 *   gf_chunk_linear(store=0)   local pass: every chunk from a ZERO state (F_state need not be
 *                              initialised), leaves its end state there; nothing else is written.
 *   gf_chunk_linear_combine    F_state slot c <- true start state of chunk c.  GF_SOLVE_LOWER /
 *                              GF_SOLVE_UPPER need Phi [B*nch][64*64], the chunk transitions of the
 *                              TRUE factor (gf_chunk_transition run on the final pass' d, z, r
 *                              rows; the backward sweep uses its transpose); GF_MATMUL_LOWER
 *                              needs c, de and D_work [B*nch][64] instead.
 *   gf_chunk_linear(store=1)   final pass: Z rows.
 * Y, Z are [B][N][R] (Z may alias Y); F_state is [B*nch][64*R]; scale != 0 divides the input
 * rows by d (solves) or multiplies them by sqrt(d) (GF_MATMUL_LOWER), as in gf_solve.
 */
int gf_chunk_linear(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int R,
                    int scale, int store, const double *c,
                    const double *Ut, const double *Wt, const double *d, const double *de,
                    const double *Y, double *Z, double *F_state, void *stream);
int gf_chunk_linear_combine(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int R,
                            const double *c, const double *de, const double *Phi,
                            double *D_work, double *F_state, void *stream);
/*
 * Two-level form of the solves' combine (long single series: the plain combine is one 64 x 64
 * mat-vec per chunk in sequence).  gf_chunk_segment_transitions composes, once per factor, the
 * transitions of every segment of seg_len chunks: Psi_out [B*nseg][64*64], nseg = ceil(nch / seg_len).
 * gf_chunk_linear_combine_seg then does what gf_chunk_linear_combine does with a sequential depth of
 * 2 seg_len + nseg chunks; V_work is [B*nseg][64*R] scratch.  PhiT_out / PsiT_out (both or neither; may be
 * NULL): the transposes of every Phi and Psi, for the backward solve -- handed to gf_chunk_linear_combine_seg as
 * PhiT / PsiT its loads run along the lanes as the forward solve's do (NULL there: the transposes are read out of
 * Phi / Psi, one 128-byte run per lane).
 */
int gf_chunk_segment_transitions(int B, int nch, int seg_len, const double *Phi, double *Psi_out,
                                 double *PhiT_out, double *PsiT_out, void *stream);
int gf_chunk_linear_combine_seg(int mode, int B, int nch, int seg_len, int R, const double *Phi,
                                const double *Psi, const double *PhiT, const double *PsiT,
                                double *F_state, double *V_work, void *stream);

/*
 * Batched dense solve A X = B (Gauss-Jordan with partial pivoting, one launch): A [batch][n][n] row-major
 * (read only), B [batch][n][nrhs] row-major, overwritten with X; n <= 192.  What the time-parallel combine
 * of a wide kernel uses for its W x W chunk maps instead of a blocked LAPACK LU (hundreds of launches per
 * call at this size).  A singular A gives garbage in X, never an error: the maps of chunks after a failed
 * pivot are singular by construction and are not used.  celerite2 has no counterpart (its factorisation
 * is sequential: celerite2.driver.factor, /root/reference/gadfly/gp.py:202).
 */
int gf_dense_solve(int batch, int n, int nrhs, const double *A, double *B, void *stream);
/* The same solve, with logdet_out [batch] <- log det A where det A > 0 (sum of log |pivot|; the sign from the
 * pivots' signs and the parity of the row permutation), NaN where det A <= 0 or A is singular / not finite. */
int gf_dense_solve_logdet(int batch, int n, int nrhs, const double *A, double *B, double *logdet_out,
                          void *stream);

/*
 * The dense combine of the time-parallel factorisation of a WIDE kernel (64 <= W <= 176), all of it in
 * this library (gadfly_dense.hip: batched FP64-MFMA GEMM tiles, mat-vecs and copies as "job" launches,
 * gf_dense_solve once per tree level).  celerite2 has no counterpart: its `factor` is sequential
 * (celerite2.driver.factor <- /root/reference/gadfly/gp.py:202, the reference's default kernel
 * /root/reference/gadfly/core.py:430-461 has W = 172).
 *
 * gf_wide_combine: one call between the nominal pass and the final pass of gf_chunk_sweep.
 *   in : h, dbar, zbar   rows of the nominal pass / gf_chunk_transition_wide ([B*N (+2)][ld], [B*N], [B*N];
 *                        ld = gf_fused_row_stride(0, Jc));
 *        Phi_state       [B*nch][gf_fused_state_size] closed-loop transitions (gf_chunk_transition_wide);
 *        S_state         [B*nch][gf_fused_state_size] END states of the nominal pass (zero start);
 *   out: S_state         the TRUE start state of every chunk (what the final pass starts from).
 *   The start states do not depend on the LAST chunk's map at all, and on the FIRST chunk's only through
 *   its end state: h, dbar, zbar and Phi_state are read for the chunks 1 .. nch - 2 only, S_state of the last
 *   chunk is not read -- the caller may skip the last chunk in the nominal pass and both end chunks in
 *   gf_chunk_transition_wide (chunk_first = 1, chunk_count = nch - 2).
 *   Steps: pack the states into dense W' x W' maps (W' = gf_dense_width(W), zero pads; identity maps pad
 *   nch to a power of two P) -> Gram sums G_c = sum h h^T / dbar, m_c = sum h zbar / dbar -> exclusive scan
 *   over the chunk maps (Blelloch, 2 log2 P - 2 levels of 3 job launches + 1 solve) -> unpack.
 *   work: gf_wide_combine_work(B, nch, Jc) doubles.  nch >= 2; B * P / 2 <= 65535.
 * gf_wide_gram, gf_lft_tree_scan: the two middle steps on dense arrays (maps [B*P][W'][W'] / [B*P][W'],
 *   row-major, W' a multiple of 16 <= 192, pads zero; symmetric G, Xbar): the scan returns the start
 *   states X_start [B*P][W'][W'], Y_start [B*P][W']; work: gf_lft_tree_work(B, P, W') doubles.
 * gf_bgemm: C_b = D_b + op(A_b) op(B_b) for b < batch (D may be NULL), row-major, any M, N, K and leading
 *   dimensions (even leading dimensions and 16-byte aligned bases take the vector-load path); strides in
 *   elements.  Used for the composed segment transitions and multi-right-hand-side chunk chains of the
 *   solves on a wide stored factor, and the conditional covariance.
 */
int gf_dense_width(int W);
int64_t gf_wide_combine_work(int B, int nch, int Jc);
int gf_wide_combine(int B, int64_t N, int64_t chunk_len, int nch, int Jc,
                    const double *h, const double *dbar, const double *zbar,
                    const double *Phi_state, double *S_state, double *acc, double *work, void *stream);
/*
 * Log-likelihood of a chunked series WITHOUT a final pass ("two sweeps").  With (X, Y) the true start state of
 * a chunk and (G, m) its Gram sums from the nominal pass,
 *     sum log d_n     = sum log dbar_n        + log det(I - X G)
 *     sum z_n^2 / d_n = sum zbar_n^2 / dbar_n + e^T G v - 2 m^T e - m^T X m,  e = Y - X m,  v = (I - X G)^-1 e,
 * so the caller reduces the NOMINAL rows (gf_reduce_tile on dbar, zbar of all chunks) and adds these
 * corrections: acc [B][3] (gf_reduce_tile's accumulators) += the sums over the chunks chunk_first ..
 * chunk_first + chunk_count - 1.  The nominal pass and the Gram sums must then cover the LAST chunk too
 * (gf_wide_combine with acc != NULL takes the Gram sums of chunks 1 .. nch - 1; pass NULL for the start
 * states alone).  A non-positive pivot ANYWHERE in the chunk makes the correction NaN -- the sign of EVERY pivot is
 * checked, not the parity det(I - X G) gives: with X = R R^T (Cholesky with diagonal pivoting to the numerical
 * rank) the chunk's pivots are all positive exactly when M = I - R^T G R is positive definite, decided by a
 * Cholesky attempt on M -- and the caller repeats such an evaluation with a final pass, which also names the failing
 * row (celerite2's LinAlgError semantics, reference gp.py:188-192).
 * gf_chunk_corrections: the W <= 63 route (state slots [B*nch][64*64] / [B*nch][64] after gf_chunk_combine[_tree];
 * W = the celerite width); one kernel per map: that check, then log det and v from an LU of I - X G with partial
 * pivoting (the values must not come from M: rounding residue of either sign in X's null directions cancels in
 * det(I - X G) and does not in a Cholesky factor);
 * work: gf_chunk_corrections_work(B, nch) doubles; B * nch <= 65535.
 */
int64_t gf_chunk_corrections_work(int B, int nch);
int gf_chunk_corrections(int B, int nch, int W, int chunk_first, int chunk_count, const double *S_state,
                         const double *F_state, const double *G, const double *m, double *acc, double *work,
                         void *stream);
int gf_wide_gram(int B, int64_t N, int64_t chunk_len, int nch, int chunk_first, int chunk_count, int P, int Jc,
                 const double *h, const double *dbar, const double *zbar,
                 double *G_out, double *m_out, void *stream);
int64_t gf_lft_tree_work(int B, int P, int WP);
int gf_lft_tree_scan(int B, int P, int WP, const double *Phi, const double *G, const double *Xbar,
                     const double *Ybar, const double *m, double *X_start, double *Y_start,
                     double *work, void *stream);
int gf_bgemm(int batch, int trans_a, int trans_b, int M, int N, int K,
             const double *A, int lda, int64_t stride_a, const double *B, int ldb, int64_t stride_b,
             const double *D, int ldd, int64_t stride_d, double *C, int ldc, int64_t stride_c,
             void *stream);

/*
 * Log-likelihood reductions (fixed-shape tree, deterministic):
 *   gf_reduce_tile   : acc[b] = {sum log d, sum z^2/d, min d} (THREE doubles per problem) over N
 *                      rows; init != 0 overwrites acc, init == 0 accumulates (tiles in order).
 *                      z == NULL: second sum is 0.  work: B * gf_reduce_work(N) doubles.
 *                      min d gives the condition estimate max(a) / min(d) that selects the
 *                      generator period (gen_period of gf_loglike_fused).
 *   gf_loglike_finish: out[b] = -0.5 (acc0 + Ntot log 2pi) - 0.5 acc1; logdet[b] = acc0
 *                      (either may be NULL); info[b] != 0 -> -inf for both.
 */
int64_t gf_reduce_work(int64_t N);
int gf_reduce_tile(int B, int64_t N, const double *d, const double *z,
                   double *work, double *acc, int init, void *stream);
int gf_loglike_finish(int B, int64_t N, const double *acc, const int32_t *info,
                      double *out, double *logdet, void *stream);

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
 * Chunk-parallel form of gf_solve for ONE right-hand side (any width; the wide stored factor's sweeps):
 *   gf_solve_chunk(store=0)  local pass: chunk c of nch sweeps its rows from F_state slot
 *                            (b * nch + c) [ld] (zeroed by the caller) and leaves its end state
 *                            there (pending push folded, the boundary's decay left to the receiver);
 *   combine                  on the caller's side: GF_MATMUL_LOWER has diagonal transitions
 *                            D_c = product of the chunk's propagator rows -> gf_chunk_diag_scan; the
 *                            solves chain F_{c+1} = loc_c + Phi_c F_c (G_{c-1} = loc_c + Phi_c^T G_c)
 *                            with the chunk transitions of the TRUE factor (gf_chunk_transition[_wide]);
 *   gf_solve_chunk(store=1)  final pass from the true start states: Z rows.
 * Y, Z [B][N] (Z may alias Y for the solves); scale as in gf_solve.
 */
int gf_solve_chunk(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int ld,
                   const double *U, const double *Wm, const double *P, const double *scale,
                   const double *Y, double *Z, double *F_state, int store, void *stream);
/* The same with R right-hand sides (Y, Z [B][N][R]; F_state [B * nch][ld][R]; B * nch <= 65535): the
 * conditional variance / covariance on a wide stored factor (celerite2: solve_lower / solve_upper with a
 * matrix right-hand side, /root/reference/gadfly/gp.py:295-304 through ConditionalDistribution). */
int gf_solve_chunk_rhs(int mode, int B, int64_t N, int64_t chunk_len, int nch, int W, int ld, int R,
                       const double *U, const double *Wm, const double *P, const double *scale,
                       const double *Y, double *Z, double *F_state, int store, void *stream);
int gf_chunk_diag_scan(int B, int nch, int rows, int R, const double *D, double *F_state, void *stream);

/*
 * Cross-covariance block for the conditional variance / covariance (celerite2's
 * ConditionalDistribution.variance builds it densely on the host; gp.py:295-304 is gadfly's use):
 *   out[b][n][r] = k(|t[n] - ts[r]|) from the celerite coefficients, layout [B][N][R] = the
 * right-hand-side layout of gf_solve / gf_chunk_linear, so K^-1 K(t, t*) follows without a copy.
 * t [B|1][N] (stride t_bs), ts [B|1][R] (stride ts_bs), R <= 4096.  For an exposure-integrated kernel
 * the coefficient form holds for lags >= delta only: the caller patches the few closer entries.
 */
int gf_cross_covariance(int B, int64_t N, int R, int Jr, int Jc,
                        const double *ar, const double *cr, const double *ac,
                        const double *bc, const double *cc, const double *dc,
                        const double *t, int64_t t_bs, const double *ts, int64_t ts_bs,
                        double *out, void *stream);

/*
 * Power spectral density of R evenly sampled series from their one-sided FFT (hipFFT's rfft,
 * interleaved complex [R][M], M = N/2 + 1): replaces the numpy expression of
 * PowerSpectrum._fft (gadfly/psd.py:566-587),
 *   power[r][k] = (re^2 + im^2)(spec[r][first + k]) * norm,   norm = d / sqrt(2 pi) / N,
 * power is [R][M - first]; first = 1 drops the zero frequency (include_zero_freq=False, psd.py:559-561).
 */
int gf_psd_power(int R, int64_t M, int64_t first, double norm, const double *spec,
                 double *power, void *stream);

/*
 * Binned power spectrum: replaces the two scipy.stats.binned_statistic passes of
 * bin_power_spectrum (gadfly/psd.py:229-300) with spectral_binning / spectral_binning_err
 * (psd.py:186-227) as statistics.  x [M] is the ascending frequency axis the bins were drawn on
 * (log10 frequency by default), power [R][M]; bin b holds the points start[b] <= i < start[b+1]
 * (start [nb+1], int64, ascending: the host applies binned_statistic's edge rules).  Per bin and
 * series: stat = trapz(power, x) / span, err = std(power) / sqrt(n) * mean(x) / span / constant,
 * span = x_last - x_first; a one-point (or zero-span) bin gives the point itself for both, an
 * empty bin NaN.  stat, err are [R][nb].
 */
int gf_psd_bin(int R, int64_t M, int nb, const double *x, const double *power,
               const int64_t *start, double constant, double *stat, double *err, void *stream);

/*
 * Missing cadences of an evenly sampled light curve filled by linear interpolation: replaces
 * interpolate_missing_data (gadfly/interp.py:6-60; called at psd.py:495, :531 before every FFT).
 * t, f [N] ascending times and fluxes; cadences [N] int64 or NULL (then the cadence index of a
 * point is rint((t - t[0]) / dt), interp.py:41-44); dt = the median spacing per cadence
 * (interp.py:36, :40), computed by the caller.
 *   gf_interp_plan: offsets [N + 1] <- i + (number of cadences missing before point i),
 *     offsets[N] = length of the filled series; work = gf_interp_work(N) int64 words of scratch.
 *   gf_interp_fill: t_out, f_out [offsets[N]] <- the input points and the missing cadences at their
 *     grid times t[0] + index * dt (interp.py:51) with flux np.interp(x, t, f) (interp.py:54: the
 *     chord of the interval that holds x, each operation rounded on its own -- bit-identical to the
 *     numpy result), merged in time order (the reference's argsort, interp.py:57-59; with cadence
 *     numbers given the grid may drift past neighbouring time stamps, so positions are ranks).
 * Both read t[0] back to the host (one 8-byte copy, a stream synchronisation).
 */
int64_t gf_interp_work(int64_t N);
int gf_interp_plan(int64_t N, const double *t, const int64_t *cadences, double dt,
                   int64_t *offsets, int64_t *work, void *stream);
int gf_interp_fill(int64_t N, const double *t, const double *f, const int64_t *cadences, double dt,
                   const int64_t *offsets, double *t_out, double *f_out, void *stream);

/*
 * Conditional mean at M new (sorted) times t1 given alpha = K^-1 (y - mean) at the N
 * observed times t2:
 *   mu[m] = sum_{t2[n] <= t1[m]} (U1[m] o e^{-c (t1[m]-t2[n])}) . V2[n] alpha[n]
 *         + sum_{t2[n] >  t1[m]} (V1[m] o e^{-c (t2[n]-t1[m])}) . U2[n] alpha[n]
 *   c [B][W]; U1, V1 [B][M][ld]; U2, V2, P2 [B][N][ld]; t1 [B|1][M]; t2 [B|1][N]
 *   qidx [B][M] (int64): number of observed rows with t2 <= t1[m] (searchsorted, side "right");
 *   work: gf_general_matmul_work(B, M, N, W) doubles.
 * Long series are swept chunk-parallel (the recurrence is a decayed prefix sum: local pass,
 * scan of the chunk states, second pass over the chunks that contain queries).
 */
int64_t gf_general_matmul_work(int B, int64_t M, int64_t N, int W);
int gf_general_matmul(int B, int64_t M, int64_t N, int W, int ld,
                      const double *c,
                      const double *t1, int64_t t1_bs, const double *U1, const double *V1,
                      const double *t2, int64_t t2_bs, const double *U2, const double *V2,
                      const double *P2, const double *alpha, const int64_t *qidx,
                      double *work, double *mu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GADFLY_HIP_H */
