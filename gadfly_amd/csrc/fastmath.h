// Shared by the CPU validation harness (scratch/fastmath_test.c) and the HIP build: plain C99.
#ifndef FM_INLINE
#define FM_INLINE static inline
#endif
// Cody-Waite reduction by pi/2 in three 33-bit pieces (each n * piece is exact for |n| < 2^20),
// three rounds unconditionally (151 bits of pi/2), then the classic minimax kernels on
// [-pi/4, pi/4] with the double-double remainder.  Valid for |x| < 2^20 * pi/2 ~ 1.647e6.
FM_INLINE void fm_sincos(double x, double *sn, double *cs) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11;
    const double p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21;
    const double p3 = 2.02226624871116645580e-21, p3t = 8.47842766036889956997e-32;
    const double fn = rint(x * invpio2);
    double r = x - fn * p1;                  // exact product, one rounding
    double w = fn * p1t;
    double t = r;
    w = fn * p2;  r = t - w;  w = fn * p2t - ((t - r) - w);
    t = r;
    w = fn * p3;  r = t - w;  w = fn * p3t - ((t - r) - w);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double z = y0 * y0;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y0;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    const int q = ((int)fn) & 3;
    const double s_ = (q & 1) ? kc : ks;
    const double c_ = (q & 1) ? ks : kc;
    *sn = (q & 2) ? -s_ : s_;
    *cs = ((q + 1) & 2) ? -c_ : c_;
}

// exp(x) for x <= 0 (any magnitude): k = rint(x/ln2), Taylor degree 13 on |r| <= ln2/2
FM_INLINE double fm_exp(double x) {
    const double invln2 = 1.44269504088896338700e+00;
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
    if (x < -745.0) return 0.0;
    const double k = rint(x * invln2);
    const double r = fma(-k, ln2lo, fma(-k, ln2hi, x));
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
