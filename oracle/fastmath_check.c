#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
/* TEST INFRASTRUCTURE: validates gadfly_amd/csrc/fastmath.h (the sincos/exp used by k_build2)
 * against long-double libm.  Build: gcc -O2 -march=x86-64-v3 -o fastmath_check fastmath_check.c -lm */
#include "../gadfly_amd/csrc/fastmath.h"
static double ulp_err(double a, double ref) {
    if (a == ref) return 0; double u = nextafter(fabs(ref), INFINITY) - fabs(ref); return fabs(a - ref) / u; }
int main() {
    srand48(1); double ms = 0, mc = 0, me = 0; long double worst = 0;
    for (long i = 0; i < 20000000; ++i) {
        double x = (drand48() * 2 - 1) * ((i % 5 == 0) ? 1.0e12 : (i % 5 == 1) ? 6.0e9 : (i % 5 == 2) ? 3.0e9 : (i % 5 == 3) ? 2.0e7 : 1.6e6);
        if (i % 7 == 0) x = rint(x / M_PI_2) * M_PI_2 + (drand48() - 0.5) * 1e-9;   // near multiples of pi/2
        if (i % 13 == 0) x = rint(x / M_PI_2) * M_PI_2;                             // the closest doubles to them
        double s, c; fm_sincos(x, &s, &c);
        long double rs = sinl((long double)x), rc = cosl((long double)x);
        double es = ulp_err(s, (double)rs), ec = ulp_err(c, (double)rc);
        if (es > ms) ms = es; if (ec > mc) mc = ec;
        double xe = -drand48() * 40.0; if (i % 11 == 0) xe = -drand48() * 740.0;
        double ee = ulp_err(fm_exp(xe), (double)expl((long double)xe));
        if (ee > me) me = ee;
    }
    printf("max ulp err: sin %.3f cos %.3f exp %.3f\n", ms, mc, me);
    return 0;
}
