// philox.hpp - counter-based RNG contract of the engine (DESIGN.md "RNG contract").
//
// Philox4x32-10, key = 64-bit seed, counter = (particle id lo, particle id hi, stage, tag) with
// tag = purpose<<28 | t<<8 | q.  One call yields two uniforms in (0,1): ua from words 0,1 and ub from
// words 2,3, u = ((x >> 11) + 0.5) * 2^-53.  A Box-Muller pair is r*(cos 2πub, sin 2πub), r = sqrt(-2 ln ua).
// Streams replace the reference's Random.rand()/randn()/shuffle call sites:
//   P_MUT : src/mutation.jl:66,133 and src/helpers.jl:97 (per particle, per (mh_step, block))
//   P_RES : src/resample.jl:30,48        P_BLK : src/helpers.jl:216        P_INIT : src/initialization.jl:27,57
// Because the counter carries the GLOBAL particle id, results do not depend on how particles are sharded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bmmath.hpp"

namespace smcmi {

enum { P_MUT = 0, P_RES = 1, P_BLK = 2, P_INIT = 3 };

__host__ __device__ inline uint32_t rng_tag(uint32_t purpose, uint32_t t, uint32_t q) {
    return (purpose << 28) | ((t & 0xFFFFFu) << 8) | (q & 0xFFu);
}

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_lo / mul_hi pair: 32-bit integer
        // multiplies are quarter-rate on CDNA, and Philox is ~30 % of the mutation kernel's issue slots
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}

__host__ __device__ inline void uniform_pair(uint64_t seed, uint64_t pid, uint32_t stage, uint32_t tag, double &ua,
                                             double &ub) {
    const u32x4 o = philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), stage, tag, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    ua = u53(o.x, o.y);
    ub = u53(o.z, o.w);
}

// The Box-Muller pieces every kernel uses (one definition: all engines draw the same bits): bmmath.hpp's cut-down functions, or - with
// -DSMCMI_LIBM_BOX_MULLER, the A/B build - the device library's general-purpose ones.
#ifdef SMCMI_LIBM_BOX_MULLER
__device__ inline double bx_neg2log(double u) { return -2.0 * log(u); }
__device__ inline double bx_sqrt(double x) { return sqrt(x); }
__device__ inline void bx_sincos2pi(double ub, double *s, double *c) { sincospi(2.0 * ub, s, c); }
#else
__device__ inline double bx_neg2log(double u) { return bm_neg2log(u); }
__device__ inline double bx_sqrt(double x) { return bm_sqrt(x); }
__device__ inline void bx_sincos2pi(double ub, double *s, double *c) { bm_sincos2pi(ub, s, c); }
#endif

__device__ inline void normal_pair(uint64_t seed, uint64_t pid, uint32_t stage, uint32_t tag, double &z0, double &z1) {
    double ua, ub;
    uniform_pair(seed, pid, stage, tag, ua, ub);
    const double r = bx_sqrt(bx_neg2log(ua));
    double s, c;
    bx_sincos2pi(ub, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

}  // namespace smcmi
