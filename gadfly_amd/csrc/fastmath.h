// Shared by the CPU validation harness (scratch/fastmath_test.c) and the HIP build: plain C99.
#ifndef FM_INLINE
#define FM_INLINE static inline
#endif
// Argument reduction by pi/2 with FMA: fn = rint(x 2/pi); x - fn P1 is formed EXACTLY inside one fma
// (P1 = pi/2 rounded to 53 bits), the remaining fn (P2 + P3) -- pi/2 to 159 bits -- is taken off in
// double-double arithmetic (fn P2 split exactly by a second fma), so the remainder y0 + y1 carries no
// cancellation error: fn (P1 + P2 + P3) misses fn pi/2 by |fn| 2^-160, i.e. 4e-37 at |x| = 1e12, against a
// remainder that no double below 1e12 brings under ~1e-20.  The quadrant fn mod 4 is read from the low
// mantissa bits of fn + 1.5 2^52 (two's complement there: exact for |fn| < 2^51, no float -> int conversion
// whose range would cap the argument -- round 3 took (int)fn and stopped at 2^31 pi/2 = 3.37e9, below the
// 4-5e9 rad a JD-based time axis reaches with the solar p-modes, /root/reference/gadfly/gp.py:79-80).
// Then the classic minimax kernels on [-pi/4, pi/4] (|r| may exceed pi/4 by ulp(x 2/pi) / 2 <= 6e-5 when the
// rounded product picks the neighbouring fn: inside the kernels' margin).  <= 1 ulp against long-double libm
// over +-1e12, including the doubles closest to multiples of pi/2 (oracle/fastmath_check.c).  (The previous
// three-round Cody-Waite scheme with 33-bit pieces was exact without FMA but stopped at 2^20 pi/2 =
// 1.6e6 -- a 1e6-point series at one-minute cadence already exceeds it with the solar p-modes -- and
// cost five more operations.)
#define FM_SINCOS_RANGE 1.0e12
FM_INLINE void fm_sincos(double x, double *sn, double *cs) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double P1 = 1.5707963267948966e+00;       // 0x3FF921FB54442D18
    const double P2 = 6.123233995736766e-17;        // 0x3C91A62633145C07
    const double P3 = -1.4973849048591698e-33;      // 0xB91F1976B7ED8FBC
    double y0, y1, fn, qm;
    {
#ifdef __clang__
#pragma clang fp contract(off)                      // the error terms below rely on every rounding as written
#endif
        fn = rint(x * invpio2);
        qm = fn + 6755399441055744.0;               // 1.5 * 2^52: fn's integer bits land in the low mantissa
        const double r0 = fma(-fn, P1, x);          // exact difference, one rounding
        const double h = fn * P2;
        const double l = fma(fn, P2, -h);           // fn P2 = h + l exactly
        const double y0a = r0 - h;
        const double e = (((r0 - y0a) - h) - l) - fn * P3;      // what y0a misses
        y0 = y0a + e;
        y1 = (y0a - y0) + e;
    }
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
    unsigned long long qb;
    __builtin_memcpy(&qb, &qm, sizeof(qb));
    const int q = (int)(qb & 3ull);
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
