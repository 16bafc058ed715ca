/* accuracy of csrc/bmmath.hpp against long-double libm (the hardware estimates emulated at 22 bits).  gcc -O2 -ffp-contract=off tests/bmmath_check.c -lm */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdbool.h>
#include "../smc.jl_amd/csrc/bmmath.hpp"
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double u53(uint64_t x) { return ((double)(x >> 11) + 0.5) * 0x1.0p-53; }
static double ulp_err(double got, long double want) {
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
    int e; frexpl(want, &e);
    long double ulp = ldexpl(1.0L, e - 53);
    return (double)fabsl(((long double)got - want) / ulp);
}
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000000;
    double el = 0, es = 0, esn = 0, ecs = 0, ez = 0;
    long wrong_sqrt = 0;
    const long double PI = 3.14159265358979323846264338327950288L;
    for (long i = 0; i < n + 64; ++i) {
        double ua, ub;
        if (i < n) {
            uint64_t a = rnd(), b = rnd();
            if ((i & 7) == 1) a >>= (rnd() % 52);            /* small uniforms: the tail of the normals */
            if ((i & 7) == 2) a = ~(a >> (rnd() % 52));      /* uniforms next to 1 */
            if ((i & 7) == 3) b >>= (rnd() % 52);
            ua = u53(a); ub = u53(b);
        } else {       /* end points and quadrant boundaries */
            const double ends[] = {0x1.0p-54, 0x1.8p-54, 1.0, 1.0 - 0x1.0p-53, 0.5, 0.25, 0.125, 0.375, 0.625, 0.75, 0.875, 0x1.0p-53, 0.70710678118654752440, 0.70710678118654757, 0.7071067811865474, 0.3535533905932738};
            ua = ends[(i - n) % 16]; ub = ends[((i - n) / 4) % 16];
        }
        const double l = bm_neg2log(ua);
        const long double lw = -2.0L * logl((long double)ua);
        double e = ulp_err(l, lw); if (e > el) el = e;
        const double q = bm_sqrt(l);
        e = ulp_err(q, sqrtl((long double)l)); if (e > es) es = e;
        if (q != sqrt(l)) ++wrong_sqrt;
        double sn, cs;
        bm_sincos2pi(ub, &sn, &cs);
        const long double th = 2.0L * PI * (long double)ub;
        /* compare against sin / cos of the EXACT angle 2 pi ub: reduce in long double the way the function does */
        long double t = 2.0L * (long double)ub; long double qd = rintl(2.0L * t); long double r = t - qd / 2.0L;
        long double S = sinl(PI * r), C = cosl(PI * r), ws, wc; int qi = (int)qd & 3;
        if (qi == 0) { ws = S; wc = C; } else if (qi == 1) { ws = C; wc = -S; } else if (qi == 2) { ws = -S; wc = -C; } else { ws = -C; wc = S; }
        (void)th;
        e = ulp_err(sn, ws); if (e > esn) esn = e;
        e = ulp_err(cs, wc); if (e > ecs) ecs = e;
        double z0, z1; bm_normal_pair(ua, ub, &z0, &z1);
        e = ulp_err(z0, sqrtl(lw) * wc); if (e > ez) ez = e;
        e = ulp_err(z1, sqrtl(lw) * ws); if (e > ez) ez = e;
    }
    printf("n %ld  max ulp: -2log %.3f  sqrt %.3f (differs from correctly rounded in %ld)  sin %.3f  cos %.3f  normals %.3f\n", n, el, es, wrong_sqrt, esn, ecs, ez);
    return 0;
}
