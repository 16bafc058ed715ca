// The three elementary functions of the Box-Muller transform, cut to what the transform needs.
//
// The RNG contract forms two normals per Philox call as  r = sqrt(-2 log ua),  z0 = r cos(2π ub),  z1 = r sin(2π ub)
// with ua, ub ∈ (0, 1] the 53-bit uniforms of philox.hpp.  The device library's log / sqrt / sincospi are general-purpose (denormals,
// infinities, NaN, huge arguments, < 1 ulp through double-double steps): 98 + 22 + 70 VALU instructions per pair - a fifth of the
// mutation kernel's instructions at n_para = 10 (profiles/r04_isa_k_mutate_reg.json).  The arguments here are never special:
//   ua ∈ [2^-54, 1]  (normal, positive)         -> log: mantissa / exponent split, atanh series in s = f / (2 + f) (the classic
//                                                  seven-coefficient minimax of the freely distributable fdlibm e_log.c), one division
//                                                  by reciprocal estimate + two Newton steps + a remainder correction;
//   x = -2 log ua ∈ [0, 75]                      -> sqrt: reciprocal-square-root estimate, one coupled step, two remainder corrections;
//   t = 2 ub ∈ (0, 2]                            -> sin(π t), cos(π t): t - q / 2 exactly (q = round(2 t) ∈ 0..4), Taylor polynomials on
//                                                  |r| <= 1/4 (truncation < 0.02 ulp), quadrant by swap / sign.
// Every multiply-add is an explicit fma, so the strict-FP and the product build give the same bits.  Measured against long-double libm
// on 2·10⁷ arguments incl. the end points (tests/bmmath_check.c, which includes THIS file with the estimates emulated at 2^-22):
// -2 log <= 0.81 ulp, sqrt correctly rounded in every case, sin / cos <= 1.07 ulp.  In the mutation kernel: 104 VALU instructions per
// pair instead of 160 (the polynomial coefficients as scalar operands).
// A normal differs from the libm-based oracle's in the last place or two - as it did with the device library - which moves a
// Metropolis decision only when the uniform lies within ~1e-15 of the acceptance ratio (tests/test_gpu_strict.py counts flips).
#pragma once

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define BM_FN __device__ inline
#define BM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define BM_RCP(x) __builtin_amdgcn_rcp(x)
#define BM_RSQ(x) __builtin_amdgcn_rsq(x)
#define BM_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define BM_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define BM_RINT(x) __builtin_rint(x)
#else   // host build of the accuracy check: the hardware estimates emulated with 22 good bits
#include <math.h>
#define BM_FN static inline
#define BM_FMA(a, b, c) fma((a), (b), (c))
static inline double bm_emul_trunc(double x) { union { double d; unsigned long long u; } v; v.d = x; v.u &= ~((1ull << 30) - 1); return v.d; }
#define BM_RCP(x) bm_emul_trunc(1.0 / (x))
#define BM_RSQ(x) bm_emul_trunc(1.0 / sqrt(x))
static inline double bm_host_mant(double x) { int e; return frexp(x, &e); }
static inline int bm_host_exp(double x) { int e; frexp(x, &e); return e; }
#define BM_FREXP_MANT(x) bm_host_mant(x)
#define BM_FREXP_EXP(x) bm_host_exp(x)
#define BM_RINT(x) rint(x)
#endif

// -2 log(u), u ∈ [2^-54, 1]
BM_FN double bm_neg2log(double u) {
    double m = BM_FREXP_MANT(u);                        // [1/2, 1)
    int e = BM_FREXP_EXP(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                 // [sqrt(1/2), sqrt(2))
    e = lo ? e - 1 : e;
    const double f = m - 1.0;                           // exact
    const double d = 2.0 + f;
    double r = BM_RCP(d);
    r = BM_FMA(BM_FMA(-d, r, 1.0), r, r);
    r = BM_FMA(BM_FMA(-d, r, 1.0), r, r);
    double s = f * r;
    s = BM_FMA(BM_FMA(-d, s, f), r, s);                 // s = f / (2 + f)
    const double z = s * s, w = z * z;
    const double t1 = w * BM_FMA(w, BM_FMA(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * BM_FMA(w, BM_FMA(w, BM_FMA(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    // log u = dk ln2_hi - ((hfsq - (s (hfsq + R) + dk ln2_lo)) - f)
    const double inner = BM_FMA(s, hfsq + R, dk * 1.90821492927058770002e-10);
    const double lg = BM_FMA(dk, 6.93147180369123816490e-01, -((hfsq - inner) - f));
    return -2.0 * lg;
}

// sqrt(x), x ∈ [0, 2^10]
BM_FN double bm_sqrt(double x) {
    const double y = BM_RSQ(x);
    double g = x * y, h = 0.5 * y;
    const double r = BM_FMA(-h, g, 0.5);
    g = BM_FMA(g, r, g);
    h = BM_FMA(h, r, h);
    double dd = BM_FMA(-g, g, x);
    g = BM_FMA(dd, h, g);
    dd = BM_FMA(-g, g, x);
    g = BM_FMA(dd, h, g);
    return x > 0.0 ? g : 0.0;                           // (ua = 1 exactly - one draw in 2^53 - gives x = 0: the estimate is infinite there)
}

// sin(2π ub), cos(2π ub), ub ∈ (0, 1]
BM_FN void bm_sincos2pi(double ub, double *sn, double *cs) {
    const double t = ub + ub;                           // (0, 2]
    const double qd = BM_RINT(t + t);                   // 0 .. 4
    const int q = (int)qd;
    const double r = BM_FMA(qd, -0.5, t);               // exact, |r| <= 1/4
    const double r2 = r * r;
    double ps = 0x1.aaec32af93359p-21;
    ps = BM_FMA(ps, r2, -0x1.6fadb9f155744p-16);
    ps = BM_FMA(ps, r2, 0x1.e8f434d018d63p-12);
    ps = BM_FMA(ps, r2, -0x1.e3074fde8871fp-8);
    ps = BM_FMA(ps, r2, 0x1.50783487ee782p-4);
    ps = BM_FMA(ps, r2, -0x1.32d2cce62bd86p-1);
    ps = BM_FMA(ps, r2, 0x1.466bc6775aae2p+1);
    ps = BM_FMA(ps, r2, -0x1.4abbce625be53p+2);
    // π r + r³ (...), π = hi + lo: the leading product carries the result's rounding, nothing else does
    const double S = BM_FMA(r, 0x1.921fb54442d18p+1, r * BM_FMA(r2, ps, 0x1.1a62633145c07p-53));
    double pc = 0x1.20c62c2f2d7f5p-18;
    pc = BM_FMA(pc, r2, -0x1.b6e24f44b128fp-14);
    pc = BM_FMA(pc, r2, 0x1.f9d38a3763cc3p-10);
    pc = BM_FMA(pc, r2, -0x1.a6d1f2a204a8cp-6);
    pc = BM_FMA(pc, r2, 0x1.e1f506891babbp-3);
    pc = BM_FMA(pc, r2, -0x1.55d3c7e3cbffap+0);
    pc = BM_FMA(pc, r2, 0x1.03c1f081b5ac4p+2);
    pc = BM_FMA(pc, r2, -0x1.3bd3cc9be45dep+2);
    const double C = BM_FMA(pc, r2, 1.0);
    // quadrant q (mod 4): 0 (S, C), 1 (C, -S), 2 (-S, -C), 3 (-C, S)
    const bool swap = (q & 1) != 0;
    const double a = swap ? C : S, b = swap ? S : C;
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}

BM_FN void bm_normal_pair(double ua, double ub, double *z0, double *z1) {
    const double r = bm_sqrt(bm_neg2log(ua));
    double s, c;
    bm_sincos2pi(ub, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}
