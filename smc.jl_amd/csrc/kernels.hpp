// kernels.hpp - HIP kernels of the SMC stage for gfx950 (CDNA4, 64-wide wavefronts).
//
// Cloud layout in HBM: column-major n x R doubles per buffer (two ping-pong buffers), i.e. one contiguous
// array per parameter / metadata column -> every kernel reads and writes fully coalesced 8-byte lanes.
// All stage decisions are taken on the device from DevState, so a stage is a fixed launch sequence:
//   k_stage_begin -> P x k_pass<KC,false> -> k_pass<1,true> -> k_post_correct -> k_scan_weights ->
//   k_resample_gather -> k_moments -> k_prepare_mutation -> k_mutate*
// Reductions are deterministic: wavefront butterfly -> LDS -> per-block partials -> fixed-order final sum,
// and the final sum + decision of pass p is recomputed by every block in the prologue of pass p+1 (no
// single-block "decide" launches, no atomics, no grid barriers).
#pragma once
#include <hip/hip_runtime.h>

#include "devstate.hpp"
// The mutation kernels let the compiler contract a*b+c into fused multiply-adds (the product; ~3 % of the MH decisions' operands
// differ in the last bit from the uncontracted oracle's, about one decision in 10^5 flips).  -DSMCMI_STRICT_FP (libsmcmi_strict.so,
// `make libsmcmi_strict.so`) builds the same library with every contraction off: the variant the parity tests compare decision for
// decision with the CPU restatement (tests/test_gpu_strict.py).
#ifdef SMCMI_STRICT_FP
#define SMCMI_FP_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SMCMI_FP_CONTRACT _Pragma("clang fp contract(fast)")
#endif

#include "model.hpp"
#include "philox.hpp"

namespace smcmi {

constexpr int TB = 256;  // threads per block for streaming kernels (4 wavefronts)

// development aid: shader-clock stamp of (block 0, thread 0) into prof[slot] when prof != nullptr
#define SMCMI_STAMP(prof, slot)                                                                              \
    do {                                                                                                    \
        if ((prof) != nullptr && threadIdx.x == 0 && blockIdx.x == 0) {                                      \
            unsigned long long tt_;                                                                          \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory"); \
            (prof)[slot] = (long long)tt_;                                                                   \
        }                                                                                                   \
    } while (0)

// ------------------------------------------------------------------------------------------------ reductions
// Butterfly "reduce-scatter" across the 64 lanes of a wavefront: M accumulators per lane go in, and lane l
// comes out holding (in a[0]) the wavefront total of accumulator  l >> (6 - log2 M).  M-1 shuffles instead
// of 6 M for M independent all-reduces.
// Lane exchanges without the LDS crossbar (ds_bpermute: what __shfl_xor compiles to): gfx950's v_permlane32_swap /
// v_permlane16_swap exchange register halves / odd-even 16-lane rows between two VGPRs in one instruction, DPP moves cover the
// distances inside a row.  Same values, same order of the additions as the __shfl_xor formulation (bit-identical results).
template <int CTRL, int BANK>
__device__ inline double dpp_mov_f64(double old, double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
// value of lane (l ^ DIST), DIST = 1, 2, 4, 8
template <int DIST>
__device__ inline double fetch_xor(double x) {
    static_assert(DIST == 1 || DIST == 2 || DIST == 4 || DIST == 8, "row-local distances only");
    if constexpr (DIST == 1) return dpp_mov_f64<0xB1, 0xf>(x, x);            // quad_perm [1,0,3,2]
    else if constexpr (DIST == 2) return dpp_mov_f64<0x4E, 0xf>(x, x);       // quad_perm [2,3,0,1]
    else if constexpr (DIST == 8) return dpp_mov_f64<0x128, 0xf>(x, x);      // row_ror:8
    else {
        const double t = dpp_mov_f64<0x104, 0x5>(x, x);                      // banks 0, 2 (lane & 4 == 0) read lane + 4: row_shl:4
        return dpp_mov_f64<0x114, 0xA>(t, x);                                // banks 1, 3 read lane - 4: row_shr:4
    }
}
// a: the accumulator the lower lanes (lane & DIST == 0) keep, b: the one the upper lanes keep.  Returns, in every lane, its kept
// accumulator plus the partner lane's copy of the same accumulator.  DIST = 16 or 32.
template <int DIST>
__device__ inline double swap_add(double a, double b) {
    static_assert(DIST == 16 || DIST == 32, "row / half-wave exchanges only");
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (DIST == 32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    }
}
template <int DIST>
__device__ inline double xor_add(double x) {         // x + (x of lane ^ DIST)
    if constexpr (DIST >= 16) return swap_add<DIST>(x, x);
    else return x + fetch_xor<DIST>(x);
}

template <int HALF, int DIST>
struct Butterfly {
    template <int M>
    __device__ static inline void run(double (&a)[M], int lane) {
        if constexpr (DIST >= 16) {
#pragma unroll
            for (int i = 0; i < HALF; ++i) a[i] = swap_add<DIST>(a[i], a[HALF + i]);
        } else {
            const bool upper = (lane & DIST) != 0;
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                const double keep = upper ? a[HALF + i] : a[i];
                const double send = upper ? a[i] : a[HALF + i];
                a[i] = keep + fetch_xor<DIST>(send);
            }
        }
        Butterfly<HALF / 2, DIST / 2>::run(a, lane);
    }
};
template <int DIST>
struct Butterfly<0, DIST> {
    template <int M>
    __device__ static inline void run(double (&a)[M], int lane) {
        a[0] = xor_add<DIST>(a[0]);
        Butterfly<0, DIST / 2>::run(a, lane);
    }
};
template <>
struct Butterfly<0, 0> {
    template <int M>
    __device__ static inline void run(double (&)[M], int) {}
};
constexpr int ilog2(int m) { return m <= 1 ? 0 : 1 + ilog2(m / 2); }

// Block-wide deterministic reduction of M accumulators per thread; thread t < M returns accumulator t's total.
// `red` is LDS scratch of (TB/64) * M doubles.
template <int M>
__device__ inline double block_reduce_many(double (&a)[M], double *red) {
    static_assert(M >= 1 && M <= 64 && (M & (M - 1)) == 0, "M must be a power of two <= 64");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Butterfly<M / 2, 32>::run(a, lane);
    constexpr int SH = 6 - ilog2(M);
    if ((lane & ((1 << SH) - 1)) == 0) red[wave * M + (lane >> SH)] = a[0];
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < M) {
#pragma unroll
        for (int w = 0; w < TB / 64; ++w) tot += red[w * M + threadIdx.x];
    }
    __syncthreads();
    return tot;
}

// Fixed-order sum of per-block partials partials[b * m + idx], b < nb, by one whole block: thread (s, idx) adds the
// blocks b = s (mod S), S = blockDim.x / m slices, four independent loads in flight; slices are then combined in
// slice order.  Result for idx = threadIdx.x (< m).  scratch: blockDim.x doubles.  Requires m <= blockDim.x.
__device__ inline double final_sum(const double *partials, int nb, int m, double *scratch, int stride = 0) {
    if (stride == 0) stride = m;
    const int t = threadIdx.x, S = blockDim.x / m;
    const int idx = t % m, s = t / m;
    if (s < S) {
        double a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = 0.0;
        int b = s;
        for (; b + 7 * S < nb; b += 8 * S) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += partials[(long long)(b + q * S) * stride + idx];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b + q * S < nb) a[q] += partials[(long long)(b + q * S) * stride + idx];
        scratch[s * m + idx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    // combine the S slices pairwise in a fixed tree (slice s absorbs slice s + half)
    int half = 1;
    while (half * 2 < S) half *= 2;
    for (; half >= 1; half >>= 1) {
        if (s < half && s + half < S) scratch[s * m + idx] += scratch[(s + half) * m + idx];
        __syncthreads();
    }
    const double tot = (t < m) ? scratch[t] : 0.0;
    __syncthreads();
    return tot;
}

// Fixed-order total of column `colx` of partials[nb][stride] by ONE wavefront: lane l adds rows l, l + 64, ... (eight loads in
// flight), then a 6-step xor butterfly.  Every lane returns the total.  No LDS, no block barrier.
__device__ inline double wave_column_sum(const double *partials, int nb, int stride, int colx) {
    const int lane = threadIdx.x & 63;
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.0;
    int b = lane;
    for (; b + 7 * 64 < nb; b += 8 * 64) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += partials[(long long)(b + q * 64) * stride + colx];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (b + q * 64 < nb) a[q] += partials[(long long)(b + q * 64) * stride + colx];
    double v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// two columns in one round of loads
__device__ inline void wave_column_sum2(const double *partials, int nb, int stride, int c0, int c1, double &v0, double &v1) {
    const int lane = threadIdx.x & 63;
    double a[8], g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = g[q] = 0.0;
    int b = lane;
    for (; b + 7 * 64 < nb; b += 8 * 64) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { a[q] += partials[(long long)(b + q * 64) * stride + c0]; g[q] += partials[(long long)(b + q * 64) * stride + c1]; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (b + q * 64 < nb) { a[q] += partials[(long long)(b + q * 64) * stride + c0]; g[q] += partials[(long long)(b + q * 64) * stride + c1]; }
    v0 = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    v1 = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { v0 += __shfl_xor(v0, off, 64); v1 += __shfl_xor(v1, off, 64); }
}

// Fixed-order total of nb scalars using the whole block (nb can be ~1e5 mutation blocks).
__device__ inline double final_sum1(const double *partials, int nb, double *scratch /* blockDim.x doubles */) {
    const int t = threadIdx.x, T = blockDim.x;
    double a0 = 0.0, a1 = 0.0;
    int b = t;
    for (; b + T < nb; b += 2 * T) { a0 += partials[b]; a1 += partials[b + T]; }
    if (b < nb) a0 += partials[b];
    scratch[t] = a0 + a1;
    __syncthreads();
    for (int off = T >> 1; off >= 1; off >>= 1) {
        if (t < off) scratch[t] += scratch[t + off];
        __syncthreads();
    }
    const double tot = scratch[0];
    __syncthreads();
    return tot;
}

// Two-level row totals INSIDE the prepare launch.  One block totalling the 1024 x 68 rows the correction left (config 3 on one GPU)
// is bound by what a single CU can pull from memory - 557 KB at ~43 GB/s = 13 µs of a 20 µs launch, whatever the number of loads in
// flight (32 instead of 8: no change) - with 255 CUs idle.  Blocks 1..PREP_G of the same launch therefore total a contiguous chunk of
// rows each (final_sum: the order is fixed by PREP_G, not by timing) into PREP_G group rows and take a ticket; block 0 waits for the
// PREP_G tickets and adds the group rows in order.  The group rows travel as agent-scope stores / loads (no fence: see Tail2 in
// stage2.hpp); all blocks of the launch are resident (it is the only kernel running), the wait is bounded like every other one.
constexpr int PREP_G = 15;
constexpr int PREP_MIN_ROWS = 128;   // fewer rows: the hand-over (~2 µs) costs more than one CU's total
struct PrepRed {
    double *rows;            // [PREP_G][m] group rows; null: block 0 totals the rows itself
    int *tick;               // arrivals, zero between launches
};
__device__ inline void prep_reduce_block(const PrepRed &pr, const double *partials, int nb, int m, double *scratch) {
    const int g = (int)blockIdx.x - 1;
    const int per = (nb + PREP_G - 1) / PREP_G;
    const int r0 = g * per, rows = (r0 + per <= nb) ? per : (nb > r0 ? nb - r0 : 0);
    const double v = final_sum(partials + (long long)r0 * m, rows, m, scratch);
    if ((int)threadIdx.x < m)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(pr.rows) + (long long)g * m + threadIdx.x, (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the row is acknowledged before the ticket (tail_reduce, stage2.hpp)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(pr.tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// block 0: total idx = threadIdx.x < m of the group rows (0 for the other threads); false: the tickets did not arrive (50 ms)
__device__ inline bool tickets_wait(int *tick, int need) {       // all threads call; ends with a barrier; re-arms the counter
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 5000000ll) { ok = 0; break; }
        }
        __hip_atomic_store(tick, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}
__device__ inline double agent_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline void agent_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool prep_group_total(const PrepRed &pr, int m, double *out) {
    const bool ok = tickets_wait(pr.tick, PREP_G);
    double v = 0.0;
    if ((int)threadIdx.x < m) {
        unsigned long long x[PREP_G];
#pragma unroll
        for (int g = 0; g < PREP_G; ++g)
            x[g] = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(pr.rows) + (long long)g * m + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int g = 0; g < PREP_G; ++g) v += __longlong_as_double((long long)x[g]);
    }
    *out = v;
    return ok;
}

// chunk of particles owned by block b out of nb (contiguous, multiple of TB except the last)
__device__ inline void block_chunk(long long n, int nb, int b, long long &beg, long long &end) {
    long long per = (n + nb - 1) / nb;
    per = (per + TB - 1) / TB * TB;
    beg = (long long)b * per;
    end = beg + per < n ? beg + per : n;
    if (beg > n) beg = n;
}

// Loads column k = 0..nc-1 of particle i into dst[k * stride] with 8 global loads in flight: a plain run-time loop of
// load -> LDS store serialises one ~1 µs memory round trip per column.
__device__ inline void load_columns(const double *base, long long ld, long long i, int nc, double *dst, int stride) {
    for (int k0 = 0; k0 < nc; k0 += 8) {
        double tmp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) tmp[q] = (k0 + q < nc) ? base[(long long)(k0 + q) * ld + i] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (k0 + q < nc) dst[(k0 + q) * stride] = tmp[q];
    }
}

struct CloudPtrs {
    double *buf[2];     // two n x R column-major buffers
    long long n;        // local particles (leading dimension)
    int R;
};

__device__ inline double *col(const CloudPtrs &c, int which, int column) { return c.buf[which] + (long long)column * c.n; }

// broadcast lane `src`'s double to the whole wavefront through SGPRs (v_readlane_b32 x2): a few cycles, no LDS latency
__device__ inline double bcast_lane(double x, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)__double2loint(x), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)__double2hiint(x), src);
    return __hiloint2double((int)hi, (int)lo);
}

// ------------------------------------------------------------------------------------------------ ϕ predictor
// ESS(ϕ_{n-1} + δ) = A(δ)² / B(δ) with A = Σ W e^{δ e_i}, B = Σ W² e^{2 δ e_i}, e_i = loglh_i - old_loglh_i (helpers.jl:173-181).
// The mutation epilogue accumulates the power sums a_k = Σ W (e - c)^k, b_k = Σ W² (e - c)^k, k < 16 / 15 (the common factor
// e^{δ c} cancels in A²/B), so the next stage can solve the truncated-Taylor model A_K(δ)²/B_K(δ) = ESS_bar for a starting
// point that is typically within 1e-7..1e-4 of the true root - the solver then only has to certify it with a bracket.
// uniform = every weight is 1 (the stage resampled): then b_k = a_k and the ES slots hold a_0 .. a_{ES-1} instead, a model of
// twice the order exactly where the tempering step is largest.
__device__ inline void energy_terms(double (&es)[ES], double W, double like, double like_prev, double c, bool live, bool uniform) {
    const double p = (like - like_prev) - c;
    const bool ok = live && p == p && fabs(p) < 1e300;       // -Inf likelihoods carry no weight for any δ > 0
    const double w1 = ok ? (uniform ? 1.0 : W) : 0.0, w2 = w1 * w1, pp = ok ? p : 0.0;
    double pk = 1.0;
    if (uniform) {
#pragma unroll
        for (int k = 0; k < EKU; ++k) { es[k] = w1 * pk; pk *= pp; }
    } else {
#pragma unroll
        for (int k = 0; k < EKA; ++k) { es[k] = w1 * pk; if (k < EKB) es[EKA + k] = w2 * pk; pk *= pp; }
    }
    es[EACC] = 0.0;                          // the caller puts the particle's acceptance value here
}

// Block-wide fixed-order reduction of ES accumulators for any block of `nw` wavefronts; thread t < ES gets total t.
// red: nw * ES doubles of LDS.
__device__ inline double block_reduce_es(double (&a)[ES], double *red, int nw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Butterfly<ES / 2, 32>::run(a, lane);
    constexpr int SH = 6 - ilog2(ES);
    __syncthreads();
    if ((lane & ((1 << SH) - 1)) == 0) red[wave * ES + (lane >> SH)] = a[0];
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < ES)
        for (int w = 0; w < nw; ++w) tot += red[w * ES + threadIdx.x];
    __syncthreads();
    return tot;
}

__device__ constexpr double INV_FACTORIAL[32] = {1.00000000000000000e+00, 1.00000000000000000e+00, 5.00000000000000000e-01, 1.66666666666666657e-01, 4.16666666666666644e-02, 8.33333333333333322e-03, 1.38888888888888894e-03, 1.98412698412698413e-04, 2.48015873015873016e-05, 2.75573192239858925e-06, 2.75573192239858883e-07, 2.50521083854417202e-08, 2.08767569878681002e-09, 1.60590438368216133e-10, 1.14707455977297245e-11, 7.64716373181981641e-13, 4.77947733238738525e-14, 2.81145725434552060e-15, 1.56192069685862253e-16, 8.22063524662432950e-18, 4.11031762331216484e-19, 1.95729410633912626e-20, 8.89679139245057408e-22, 3.86817017063068354e-23, 1.61173757109611839e-24, 6.44695028438447359e-26, 2.47959626322479723e-27, 9.18368986379554601e-29, 3.27988923706983776e-30, 1.13099628864477181e-31, 3.76998762881590539e-33, 1.21612504155351811e-34};

// Root δ > 0 of the Taylor model G(δ) = A(δ)² - ESS_bar B(δ) by one wavefront without serial polynomial evaluation: lane (g, k)
// owns one coefficient - g = 0: a_k / k! of A, g = 1: b_k 2^k / k! of B (after a resample b_k = a_k and the slots hold 31 orders
// of a) - forms its term c x^k by binary powering, and the four sums A = Σ term, x A' = Σ k term (same for B) come from 5-step
// xor butterflies inside the 32-lane halves: a Newton step is ~40 dependent instructions.  Every lane returns the same x (NaN
// when the model is unusable); *gprime = dESS/dδ at the root of the model.
__device__ inline double predict_delta_wave(const double *es, bool uniform, double T, double *gprime, double inv_pre = -1.0, bool refresh = true) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int lane = threadIdx.x & 63, grp = lane >> 5, k = lane & 31;
    const int KT = uniform ? EKU : (grp == 0 ? EKA : EKB);   // terms of this lane's polynomial
    const double inv = inv_pre >= 0.0 ? inv_pre : INV_FACTORIAL[k];      // 1 / k! (callers on a latency budget load it ahead)
    double c = 0.0;
    if (k < KT) c = (grp == 0 ? es[k] : (uniform ? es[k] : es[EKA + k])) * inv;
    if (grp) c = ldexp(c, k);
    // broadcasts through SGPRs (v_readlane) and butterflies on DPP / v_permlane16_swap: no LDS-crossbar round trips in the Newton loop
    const double ca0 = bcast_lane(c, 0), ca1 = bcast_lane(c, 1), ca2 = bcast_lane(c, 2), cb0 = bcast_lane(c, 32), cb1 = bcast_lane(c, 33), cb2 = bcast_lane(c, 34);
    *gprime = nan;
    if (!(ca0 > 0.0) || !(cb0 > 0.0) || !(T > 0.0)) return nan;
    const double G0 = ca0 * ca0 - T * cb0;
    if (!(G0 > 0.0)) return nan;
    const double G1 = 2.0 * ca0 * ca1 - T * cb1;
    const double G2h = ca1 * ca1 + 2.0 * ca0 * ca2 - T * cb2;
    const double disc = G1 * G1 - 4.0 * G2h * G0;
    double x;
    if (disc >= 0.0 && -G1 + sqrt(disc) > 0.0) x = 2.0 * G0 / (-G1 + sqrt(disc));
    else if (G1 < 0.0) x = -G0 / G1;
    else return nan;
    const double kd = (double)k;
    double A = 0.0, SA = 0.0, B = 0.0, SB = 0.0;
    for (int it = 0; it < 4; ++it) {
        double p = 1.0, xb = x;
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) { if ((k >> bit) & 1) p *= xb; xb *= xb; }
        double term = c * p, kterm = kd * term;
        term = xor_add<16>(term); kterm = xor_add<16>(kterm);
        term = xor_add<8>(term); kterm = xor_add<8>(kterm);
        term = xor_add<4>(term); kterm = xor_add<4>(kterm);
        term = xor_add<2>(term); kterm = xor_add<2>(kterm);
        term = xor_add<1>(term); kterm = xor_add<1>(kterm);
        A = bcast_lane(term, 0); SA = bcast_lane(kterm, 0); B = bcast_lane(term, 32); SB = bcast_lane(kterm, 32);
        // the last evaluation only refreshes A, B and the slopes at the final x; without it (refresh = false: engine 2) the slope
        // dESS/dδ is the one at the last-but-one iterate, 1e-8 away - it only scales the verification tolerance
        if (it == (refresh ? 3 : 2)) {
            if (!refresh) {
                const double G = A * A - T * B, dG = (2.0 * A * SA - T * SB) / x;
                if (!(dG < 0.0) || !(A > 0.0) || !(B > 0.0)) return nan;
                const double xn = x - G / dG;
                if (!(xn > 0.0) || !(xn < 1e300)) return nan;
                *gprime = (2.0 * A * (SA / x) * B - A * A * (SB / x)) / (B * B);
                return xn;
            }
            break;
        }
        const double G = A * A - T * B, dG = (2.0 * A * SA - T * SB) / x;
        if (!(dG < 0.0) || !(A > 0.0) || !(B > 0.0)) return nan;
        const double xn = x - G / dG;
        if (!(xn > 0.0) || !(xn < 1e300)) return nan;
        x = xn;
    }
    *gprime = (2.0 * A * (SA / x) * B - A * A * (SB / x)) / (B * B);
    return x;
}

constexpr int NPR = 3;                       // rings on each side of the predicted root, as fractions of the predicted step
constexpr double PRING[NPR] = {0x1p-10, 0x1p-20, 0x1p-30};
constexpr int NRL = 2 * NPR + 1;             // lanes that own ring points in k_stage_begin
constexpr int NLANE = (NRL + 4 > KC) ? NRL + 4 : KC;   // lanes that can hold a candidate there (rings + 4 walk steps, or KC walk steps)

// ------------------------------------------------------------------------------------------------ ϕ solver
// solve_adaptive_ϕ (src/helpers.jl:9-56) as a bracketing search driven by K-candidate ESS passes:
//   SCAN    : candidates = ϕ_prop, schedule[j], schedule[j+1], ...  - the reference's `while g(ϕ_prop) >= 0` loop
//             (helpers.jl:29-32), K entries per pass; the first candidate with g < 0 becomes ϕ_prop.
//   SECTION : Roots.fzero(g, [ϕ_n1, ϕ_prop]) (helpers.jl:49) restated as a bracketing search whose K candidates per
//             pass are 3 uniform points (guaranteed shrink) plus the secant estimate and geometric rings around it
//             (2^-3 ... 2^-23 of the bracket): superlinear in practice, 4-5 passes to 1e-12 relative.
//   FINAL   : ϕ_n known.
constexpr double RING[6] = {0x1p-3, 0x1p-7, 0x1p-11, 0x1p-15, 0x1p-19, 0x1p-23};

// One decision step from the candidate totals tot[0..KC) = Σv, tot[KC..2KC) = Σv², executed by the 64 lanes of wave 0
// (lane k owns candidate k; S and tot live in LDS; `srt` is KC doubles of LDS scratch).  Same decisions as the serial
// formulation in host/hostmath.py: first candidate with g < 0 ends the scan / splits the bracket; new candidates are the
// ascending merge of 3 uniform points, the secant estimate and 6 geometric rings on each side, deduplicated.
__device__ inline void solver_decide_wave(Solver &S, const double *tot, const double *sched, int n_phi, double rtol, int *err,
                                          double *srt) {
    const int lane = threadIdx.x & 63;
    const int nv = S.n_valid, mode = S.mode;
    const bool mine = lane < nv && lane < KC;
    const double ck = mine ? S.cand[lane] : 0.0;
    const double gk = mine ? tot[lane] * tot[lane] / tot[KC + lane] - S.ess_bar : 0.0;     // ESS(ϕ) = (Σv)²/Σv²
    const unsigned long long negmask = __ballot(mine && !(gk >= 0.0));
    double lo = S.lo, hi = S.hi, glo = S.glo, ghi = S.ghi;
    int m = negmask ? (__ffsll((long long)negmask) - 1) : -1;
    if (mode == MODE_SCAN) {
        // Candidates ascend.  Those flagged in sched_mask are the reference's walk: the current ϕ_prop, then schedule[j],
        // schedule[j+1], ... (1-based j; helpers.jl:29-32): the first of them with g < 0 becomes ϕ_prop and closes the
        // bracket.  The others are predictor rings; they only refine the bracket between the two schedule points around the root.
        const int cjk = mine ? S.cj[lane] : -1;                 // walk steps from the current ϕ_prop, -1 = ring point
        const unsigned long long smask = __ballot(cjk >= 0);
        const unsigned long long sneg = negmask & smask;
        const int ms = sneg ? (__ffsll((long long)sneg) - 1) : -1;
        if (ms >= 0) {
            const double cm = __shfl(ck, ms, 64), gm = __shfl(gk, ms, 64);
            // g(ϕ_prop) = NaN ends the reference's walk (NaN >= 0 is false, helpers.jl:29) and Roots.fzero then rejects
            // [ϕ_n1, ϕ_prop] as a bracket (helpers.jl:50): the run aborts from the solver, not from check_nan_ess.
            if (gm != gm) { if (lane == 0) *err = SMCMI_ERR_BRACKET; return; }
            hi = cm; ghi = gm;
            const unsigned long long below = smask & ((1ull << ms) - 1ull);
            const int ps = below ? 63 - __clzll((long long)below) : -1;       // previous schedule candidate (g >= 0), if any
            if (ps >= 0) {
                const double cp = __shfl(ck, ps, 64), gp = __shfl(gk, ps, 64);
                if (cp > lo) { lo = cp; glo = gp; }
            }
            const int qs = __shfl(cjk, ms, 64);
            if (lane == 0) { S.phi_prop = cm; S.j += qs; }
            // ring candidates strictly between them (list positions ps+1 .. ms-1)
            const unsigned long long upto = (1ull << ms) - 1ull;
            const unsigned long long inner = ps < 0 ? upto : (upto & ~((2ull << ps) - 1ull));
            const unsigned long long rneg = negmask & inner;
            int top = ms;                                   // list position of the upper bracket end
            if (rneg) {
                top = __ffsll((long long)rneg) - 1;
                hi = __shfl(ck, top, 64); ghi = __shfl(gk, top, 64);
            }
            if (top - 1 > ps) {                             // a ring point with g >= 0 just below it
                const double cq = __shfl(ck, top - 1, 64), gq = __shfl(gk, top - 1, 64);
                if (cq > lo) { lo = cq; glo = gq; }
            }
            m = top;
        } else {
            // every schedule candidate keeps ESS above the target: lo = the last of them, continue the walk
            const int last_s = 63 - __clzll((long long)smask);
            {
                const double cl_ = __shfl(ck, last_s, 64), gl_ = __shfl(gk, last_s, 64);
                if (cl_ > lo) { lo = cl_; glo = gl_; }
            }
            const int j_new = S.j + __shfl(cjk, last_s, 64);
            const double clast = __shfl(ck, last_s, 64);
            if (j_new > n_phi) {                         // ϕ_prop == 1 and g(1) >= 0 -> ϕ_n = 1 (helpers.jl:51-53)
                if (lane == 0) { S.lo = lo; S.glo = glo; S.phi_prop = clast; S.j = j_new; S.phi_n = clast; S.mode = MODE_FINAL; }
                return;
            }
            int c = n_phi - j_new + 1;                   // continue the scan with the next chunk of the schedule
            if (c > KC) c = KC;
            if (lane < c) { S.cand[lane] = sched[j_new - 1 + lane]; S.cj[lane] = lane; }
            if (lane == 0) { S.lo = lo; S.glo = glo; S.n_valid = c; S.j = j_new + 1; S.phi_prop = sched[j_new - 1]; }
            return;                                      // stay in SCAN
        }
    } else {
        if (m >= 0) {
            hi = __shfl(ck, m, 64); ghi = __shfl(gk, m, 64);
            if (m > 0) { lo = __shfl(ck, m - 1, 64); glo = __shfl(gk, m - 1, 64); }
        } else if (nv > 0) {
            lo = __shfl(ck, nv - 1, 64); glo = __shfl(gk, nv - 1, 64);
        }
    }
    // ---- section candidates for the bracket (lo, hi)
    // Termination: the bracket is at the requested resolution, or it is so short against the length of the tempering step
    // (the scale on which ESS(ϕ) bends: it is analytic in ϕ - ϕ_{n-1}) that linear interpolation between its evaluated ends is
    // already exact to that resolution - interpolation error ~ h² g''/(8 g') ~ h² / (8 (hi - ϕ_{n-1})), taken with a 64x margin.
    const double h = hi - lo;
    int c = 0;
    const bool interp_ok = glo >= 0.0 && ghi < 0.0 && glo < 1e300 && ghi > -1e300 && 64.0 * h * h <= rtol * hi * (hi - S.phi0);
    if (h > rtol * hi && !interp_ok) {
        double t = 0.5;
        if (glo > ghi && glo < 1e300 && ghi > -1e300) t = glo / (glo - ghi);
        const double xs = lo + h * t;
        double x = 0.0;
        if (lane < 6) x = xs - h * RING[lane];
        else if (lane == 6) x = xs;
        else if (lane < 13) x = xs + h * RING[12 - lane];
        else if (lane < 16) x = lo + h * (0.25 * (lane - 12));
        // rank among the 16 raw values (stable), scatter to sorted order
        int rank = 0;
#pragma unroll
        for (int r = 0; r < KC; ++r) {
            const double xr = __shfl(x, r, 64);
            rank += (xr < x || (xr == x && r < lane)) ? 1 : 0;
        }
        if (lane < KC) srt[rank] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): the wave's LDS writes have landed (single wave, no barrier needed)
        __builtin_amdgcn_wave_barrier();
        const double xq = lane < KC ? srt[lane] : 0.0;
        const double xp = (lane > 0 && lane < KC) ? srt[lane - 1] : lo;
        const bool ok = lane < KC && xq > lo && xq < hi && (lane == 0 || xq > xp);
        const unsigned long long okm = __ballot(ok);
        c = __popcll(okm);
        if (ok) S.cand[__popcll(okm & ((1ull << lane) - 1ull))] = xq;
    }
    if (lane == 0) {
        S.lo = lo; S.hi = hi; S.glo = glo; S.ghi = ghi;
        if (c == 0) {   // bracket at the requested resolution (or no representable interior point)
            double pn = (fabs(glo) <= fabs(ghi)) ? lo : hi;
            if (glo >= 0.0 && ghi < 0.0 && glo < 1e300 && ghi > -1e300) {  // interpolate inside the certified bracket (glo == 0: lo)
                const double xs = lo + h * (glo / (glo - ghi));
                if (xs >= lo && xs <= hi) pn = xs;
            }
            S.phi_n = pn;
            S.mode = MODE_FINAL;
        } else {
            S.n_valid = c;
            S.mode = MODE_SECTION;
        }
    }
}

// Prologue shared by every pass kernel: bring the solver state of pass p into LDS.  p == 0 reads copy 0 as written by
// k_stage_begin; p > 0 reduces the partials of pass p-1 and takes the decision; block 0 publishes copy p&1.
// Returns through *S; all threads must call.  scratch: TB doubles, tot: 2 KC doubles (LDS).
__device__ inline void solver_prologue(DevState *st, const double *sched, const double *partials_prev, int nb, int p,
                                       Solver *S, double *scratch, double *tot, int force_final, int done = 0) {
    const int t = threadIdx.x;
    constexpr int NW = sizeof(Solver) / sizeof(double);
    static_assert(sizeof(Solver) % sizeof(double) == 0, "Solver must be a whole number of doubles");
    const Solver *src = &st->sol[p == 0 ? 0 : ((p - 1) & 1)];
    if (t < NW) reinterpret_cast<double *>(S)[t] = reinterpret_cast<const double *>(src)[t];
    __syncthreads();
    if (p == 0 || done) return;
    const int mode = S->mode;
    if (mode == MODE_SCAN || mode == MODE_SECTION) {
        const double v = final_sum(partials_prev, nb, 2 * KC, scratch);
        if (t < 2 * KC) tot[t] = v;
        __syncthreads();
        __shared__ int s_err;
        if (t == 0) s_err = 0;
        __syncthreads();
        if (t < 64) solver_decide_wave(*S, tot, sched, st->rp.n_phi, st->rp.phi_rtol, &s_err, scratch);
        __syncthreads();
        if (t == 0) {
            int err = s_err;
            if (force_final && S->mode != MODE_FINAL && !err && st->rp.stall_on_exhaust) {
                // out of passes: stall.  This and every later kernel does nothing until the host resumes the stage with more
                // passes (smcmi_run); the search state (copy (p-1)&1 and the partials of pass p-1) is left intact.
                if (blockIdx.x == 0) st->done = 2;
            } else if (force_final && S->mode != MODE_FINAL && !err) {   // out of passes: accept the current bracket
                if (S->mode == MODE_SECTION) { S->phi_n = (fabs(S->glo) <= fabs(S->ghi)) ? S->lo : S->hi; S->unconverged += 1; S->mode = MODE_FINAL; }
                else err = SMCMI_ERR_BRACKET;                       // still scanning the schedule
            }
            if (err && blockIdx.x == 0) { st->err = err; st->done = 1; }
            if (err) S->mode = MODE_IDLE;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && t < NW) reinterpret_cast<double *>(&st->sol[p & 1])[t] = reinterpret_cast<const double *>(S)[t];
}

// ------------------------------------------------------------------------------------------------ ESS passes
// One pass over (loglh, old_loglh, weight): for K candidate ϕ accumulate Σ v and Σ v², v = W exp((ϕ_n1-ϕ)old + (ϕ-ϕ_n1)ℓ)
// (src/helpers.jl:173-181, always the prior_weight == 0 formula: quirk Q4).
// FINAL = true is the correction step at the chosen ϕ_n (src/smc_main.jl:401-420): the incremental weight uses the
// prior-weight variant, the unnormalised weight W̃ = W w̃ is written back and w̃ goes to the history column.
// Energy shift of the stage.  The incremental weight exp(δ e), e = loglh - old_loglh, is formed as exp(δ (e - e_shift)) with
// e_shift = the largest e of the live cloud (k_stage_begin, from the maxima the previous mutation / k_energy_max left): every
// exponent is <= 0, so Σ W̃ and Σ W̃² neither overflow nor lose the whole cloud to underflow when |δ e| runs into the hundreds
// (large-sample tempered updates) - the reference normalises before squaring (helpers.jl:173-181) and survives to |δ e| ~ 745,
// unshifted sums of squares give up at ~354.  ESS, normalised weights and moments are ratios and do not see the common factor;
// log(Σ W̃ / N) gets δ e_shift back (post_load) and the stored incremental weights their factor exp(δ e_shift).
// Prior-weight corrections (pw != 0) use another exponent and stay unshifted, like the stand-alone calls (e_shift = 0).
__device__ inline double stage_shift(const DevState *st) { return st->rp.pw == 0.0 ? st->e_shift : 0.0; }

// max over the block of v (-inf for lanes without a value); smem: one double per wavefront; all threads must call
__device__ inline double block_max(double v, double *smem, int nwaves) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = smem[0];
    for (int w = 1; w < nwaves; ++w) m = fmax(m, smem[w]);
    return m;
}
__device__ inline double energy_or_ninf(double like, double like_prev, double w, bool live) {
    const double e = like - like_prev;
    return (live && w > 0.0 && fabs(e) < 1e300) ? e : -__builtin_inf();
}

// Largest energy of the live cloud per block -> emax_part[blockIdx.x] (run start; afterwards the mutation epilogue keeps it)
// (stride, buf: engine 2 takes the maxima in its mutation-row layout - rows_mut[b][RMAX_IDX] - from buffer 0)
// (SMCMI_INST_UNIT: the instantiation units - inst2.hip, inst2b.hip, inst3.hip, 76 of the 77 translation units - launch none of the non-template
// kernels of this file, stage2.hpp and stage3.hpp; compiled there all the same they were 40 % of the library and a third of its build time)
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_energy_max(CloudPtrs cl, const DevState *st, double *emax_part, int stride = 1, int buf = -1) {
    __shared__ double smem[TB / 64];
    const int R = cl.R, src = buf >= 0 ? buf : st->cur;
    const double *loglh = col(cl, src, R - 5), *old = col(cl, src, R - 3), *w = col(cl, src, R - 1);
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    double m = -__builtin_inf();
    for (long long i = beg + threadIdx.x; i < end; i += TB) m = fmax(m, energy_or_ninf(loglh[i], old[i], w[i], true));
    m = block_max(m, smem, TB / 64);
    if (threadIdx.x == 0) emax_part[(long long)blockIdx.x * stride] = m;
}
#endif

template <int K, bool FINAL>
__global__ void __launch_bounds__(TB) k_pass(CloudPtrs cl, DevState *st, const double *sched, const double *partials_prev,
                                             double *partials_out, int nb_prev, int p, double *hist_w, long long hist_ld,
                                             long long *prof = nullptr) {
    SMCMI_STAMP(prof, 0);
    __shared__ double red[(TB / 64) * 2 * K];
    __shared__ double scratch[TB];
    __shared__ double tot[2 * KC];
    __shared__ Solver S;
    // independent scalar loads first (one memory round trip): done flag, buffer index, ϕ_{n-1}, solver copy
    const int done = st->done;
    constexpr int src = 0;             // the current cloud always lives in buffer 0 (k_moments copies a resampled cloud back)
    const double phi_prev = st->phi_prev;
    const double pw = st->rp.pw, logp_old = st->rp.logp_old, esh = stage_shift(st);
    const int stage_col = st->stage - 1;
    const bool hist = FINAL && st->rp.store_history && hist_w != nullptr;
    solver_prologue(st, sched, partials_prev, nb_prev, p, &S, scratch, tot, FINAL ? 1 : 0, done);
    if (done) return;
    SMCMI_STAMP(prof, 1);
    const int mode = S.mode;
    if (FINAL ? (mode != MODE_FINAL) : (mode != MODE_SCAN && mode != MODE_SECTION)) return;
    const int nv = S.n_valid;
    const int R = cl.R;
    const double *loglh = col(cl, src, R - 5), *old = col(cl, src, R - 3);
    double *w = col(cl, src, R - 1);
    double acc[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) acc[k] = 0.0;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    const double unshift = (FINAL && hist) ? exp((S.phi_n - phi_prev) * esh) : 1.0;     // history keeps the true exp(δ e)
    for (long long i = beg + threadIdx.x; i < end; i += TB) {
        const double l = loglh[i] - esh, o = old[i], wi = w[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!FINAL && k >= nv) continue;
            const double phi = FINAL ? S.phi_n : S.cand[k];
            double inc;
            if (!FINAL || pw == 0.0) inc = exp((phi_prev - phi) * o + (phi - phi_prev) * l);
            else if (pw == 1.0) inc = exp((phi - phi_prev) * l);
            else inc = exp((phi_prev - phi) * log(exp(o - logp_old + log(1.0 - pw)) + pw) + (phi - phi_prev) * l);
            const double v = wi * inc;
            acc[k] += v;
            acc[K + k] += v * v;
            if (FINAL) {
                w[i] = v;
                if (hist) hist_w[(long long)stage_col * hist_ld + i] = inc * unshift;
            }
        }
    }
    SMCMI_STAMP(prof, 2);
    const double total = block_reduce_many<2 * K>(acc, red);
    SMCMI_STAMP(prof, 3);
    if (threadIdx.x < 2 * K) partials_out[(long long)blockIdx.x * (2 * K) + threadIdx.x] = total;
    if (!FINAL && blockIdx.x == 0 && threadIdx.x == 0) st->solver_passes += 1;
    SMCMI_STAMP(prof, 4);
}

// Correction pass that also gathers the moments (stages where no resampling is expected, smcmi_run): weighted mean and
// covariance are ratios of sums, so they can be accumulated with the UNNORMALISED weights W̃ in the same pass that creates them
// (Σ W̃ x̃ x̃ᵀ, x̃ = (1, θ - shift), as k_moments_reg) - the cloud is not read a second time and the separate moments launch
// disappears; the weights are normalised later by the mutation kernel, which reads them anyway.  Per-block partial row:
// [ΣW̃, ΣW̃², pair sums (a <= b)]; the classic 2-column partials are written too so that a stalled stage can be resumed by the
// full path.  Same solver prologue / stall semantics as k_pass<1, true>.
template <int D>
__global__ void __launch_bounds__(TB) k_correct_moments(CloudPtrs cl, DevState *st, const double *sched, const double *partials_prev,
                                                        double *partials_fin, double *partials_cm, int nb_prev, int p, double *hist_w,
                                                        long long hist_ld, double *wt_out = nullptr) {
    constexpr int DA = D + 1, NP = DA * (DA + 1) / 2, NPF = NP + 2;
    constexpr int NCH = (NPF + 63) / 64;
    __shared__ double red[(TB / 64) * 64];
    __shared__ double scratch[TB];
    __shared__ double tot[2 * KC];
    __shared__ Solver S;
    const int done = st->done;
    const double phi_prev = st->phi_prev;
    const double pw = st->rp.pw, logp_old = st->rp.logp_old, esh = stage_shift(st);
    const int stage_col = st->stage - 1;
    const bool hist = st->rp.store_history && hist_w != nullptr;
    double sh[D];
#pragma unroll
    for (int a = 0; a < D; ++a) sh[a] = st->shift[a];
    solver_prologue(st, sched, partials_prev, nb_prev, p, &S, scratch, tot, 1, done);
    if (done) return;
    if (S.mode != MODE_FINAL) return;
    const double phi = S.phi_n;
    const int R = cl.R;
    const double *loglh = col(cl, 0, R - 5), *old = col(cl, 0, R - 3);
    double *w = col(cl, 0, R - 1);
    double acc[NCH * 64];
#pragma unroll
    for (int q = 0; q < NCH * 64; ++q) acc[q] = 0.0;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    const double unshift = hist ? exp((phi - phi_prev) * esh) : 1.0;                      // history keeps the true exp(δ e)
    for (long long i = beg + threadIdx.x; i < end; i += TB) {
        const double l = loglh[i] - esh, o = old[i], wi = w[i];
        double xx[DA];
        xx[0] = 1.0;
#pragma unroll
        for (int a = 0; a < D; ++a) xx[a + 1] = col(cl, 0, a)[i] - sh[a];
        double inc;
        if (pw == 0.0) inc = exp((phi_prev - phi) * o + (phi - phi_prev) * l);
        else if (pw == 1.0) inc = exp((phi - phi_prev) * l);
        else inc = exp((phi_prev - phi) * log(exp(o - logp_old + log(1.0 - pw)) + pw) + (phi - phi_prev) * l);
        const double v = wi * inc;
        acc[0] += v;
        acc[1] += v * v;
        if (wt_out) wt_out[i] = v; else w[i] = v;        // spec stage: W stays intact until the prediction is verified
        if (hist) hist_w[(long long)stage_col * hist_ld + i] = inc * unshift;
        int q = 2;
#pragma unroll
        for (int a = 0; a < DA; ++a) {
            const double wx = v * xx[a];
#pragma unroll
            for (int b = a; b < DA; ++b) { acc[q] += wx * xx[b]; ++q; }
        }
    }
    double *out = partials_cm + (long long)blockIdx.x * NPF;
    constexpr int REM = NPF - 64 * (NCH - 1);                                   // accumulators in the last chunk
    constexpr int REMP = REM <= 1 ? 1 : REM <= 2 ? 2 : REM <= 4 ? 4 : REM <= 8 ? 8 : REM <= 16 ? 16 : REM <= 32 ? 32 : 64;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch < NCH - 1 || REMP == 64) {
            double a64[64];
#pragma unroll
            for (int q = 0; q < 64; ++q) a64[q] = acc[ch * 64 + q];
            const double t64 = block_reduce_many<64>(a64, red);
            if (threadIdx.x < 64 && ch * 64 + (int)threadIdx.x < NPF) out[ch * 64 + threadIdx.x] = t64;
            if (ch == 0 && threadIdx.x < 2) partials_fin[2 * (long long)blockIdx.x + threadIdx.x] = t64;
        } else {
            // short tail (4 sums at d = 10): a REMP-wide butterfly instead of a 64-wide one
            double ar[REMP];
#pragma unroll
            for (int q = 0; q < REMP; ++q) ar[q] = acc[ch * 64 + q];
            const double tr = block_reduce_many<REMP>(ar, red);
            if (threadIdx.x < REMP && ch * 64 + (int)threadIdx.x < NPF) out[ch * 64 + threadIdx.x] = tr;
            if (ch == 0 && threadIdx.x < 2) partials_fin[2 * (long long)blockIdx.x + threadIdx.x] = tr;
        }
    }
}

// A spec stage whose prediction could not be used (or failed verification) is resumed by the certificate-pass path: rebuild the
// plain schedule-walk candidates in solver copy 0 from the loop scalars (ϕ_prop and j are only committed by post_write, so they
// still are the values the stage started with; ESS_bar / g(ϕ_{n-1}) were stored by k_stage_begin).
#ifndef SMCMI_INST_UNIT
static __global__ void k_solver_rearm(DevState *st, const double *sched) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Solver &S = st->sol[0];
    const int j = st->j, n_phi = st->rp.n_phi;
    const double phi_prop = st->phi_prop;
    S.mode = MODE_SCAN; S.spec = 0; S.unconverged = 0;
    S.lo = st->phi_prev; S.hi = phi_prop; S.ghi = 0.0; S.phi0 = st->phi_prev;
    S.j = j; S.phi_prop = phi_prop;
    int nv = 0;
    S.cand[nv] = phi_prop; S.cj[nv] = 0; ++nv;
    for (int q = 1; j + q - 2 < n_phi && nv < KC; ++q) { S.cand[nv] = sched[j + q - 2]; S.cj[nv] = q; ++nv; }
    S.n_valid = nv;
}
#endif

// decision of the last solver pass without a correction (stand-alone smcmi_solve_phi)
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_solver_finish(DevState *st, const double *sched, const double *partials_prev, int nb_prev, int p) {
    __shared__ double scratch[TB];
    __shared__ double tot[2 * KC];
    __shared__ Solver S;
    if (st->done) return;
    solver_prologue(st, sched, partials_prev, nb_prev, p, &S, scratch, tot, 1);
}
#endif

// Stage begin (src/smc_main.jl:378-396 + src/helpers.jl:14-20): bump the stage index, fold the previous
// mutation's acceptance sums into cloud.accept, flip the cloud buffer after a resample, pick ϕ_n from the fixed
// schedule or arm the adaptive solver with its first candidates (solver copy 0).
constexpr int BT = 1024;  // threads of the stage-begin block: enough slices that the partial reduction is one round of loads
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(BT) k_stage_begin(DevState *st, const double *sched, const double *acc_partials,
                                                    int acc_nb, Records rec, const double *esum_partials = nullptr, long long *prof = nullptr,
                                                    int spec_expected = 0, const double *emax_part = nullptr, int emax_n = 0, PrepRed rr = PrepRed{},
                                                    int *host_note = nullptr) {
    __shared__ double scratch[BT];
    __shared__ double s_em[BT], s_em2[64], s_em3[8];     // energy maxima: reduced through LDS behind the barriers the sums need anyway
    // very many rows (N >= ~5e5: one row per 256 particles): blocks 1.. of THIS launch total a contiguous chunk of rows each into
    // gridDim.x - 1 group rows (the group's energy maximum behind them) and take a ticket - as the prepare launch's (PrepRed above);
    // block 0 then reads group rows instead of rows.  This used to be a launch of its own (k_reduce_rows: 4.6 µs + a launch boundary).
    const int G = (int)gridDim.x - 1;
    if (blockIdx.x > 0) {
        if (!rr.rows || st->done) return;
        const int g = (int)blockIdx.x - 1, per = (acc_nb + G - 1) / G;
        const int r0 = g * per, rows = (r0 + per <= acc_nb) ? per : (acc_nb > r0 ? acc_nb - r0 : 0);
        const double tot = final_sum(esum_partials + (long long)r0 * ES, rows, ES, scratch);
        if (threadIdx.x < ES) agent_store(rr.rows + (long long)g * ES + threadIdx.x, tot);
        double em = -__builtin_inf();
        if (emax_part) for (int b = threadIdx.x; b < rows; b += BT) em = fmax(em, emax_part[r0 + b]);
        em = block_max(em, s_em2, BT / 64);
        if (threadIdx.x == 0) agent_store(rr.rows + (long long)G * ES + g, em);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(rr.tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const bool two = rr.rows != nullptr && G > 0;
    if (two) {
        if (st->done) return;
        if (!tickets_wait(rr.tick, G)) { if (threadIdx.x == 0) { st->err = SMCMI_ERR_TIMEOUT; st->done = 1; } return; }
        esum_partials = rr.rows; emax_part = emax_part ? rr.rows + (long long)G * ES : nullptr; emax_n = G;
    }
    const int es_nb = two ? G : acc_nb;                   // rows of the energy sums this block reads
    auto ld = [&](const double *p) { return two ? agent_load(p) : *p; };
    SMCMI_STAMP(prof, 0);
    __shared__ double s_es[ES];
    __shared__ double s_sw[64];          // window of the proposed schedule: s_sw[q] = walk step q + 1 = schedule[j + q] (1-based)
    // one round of scalar loads
    const int done = st->done, stage0 = st->stage, rs = st->do_resample, n_phi = st->rp.n_phi, fixed = st->rp.use_fixed_schedule;
    const int max_stages = st->rp.max_stages, rl = st->resampled_last, j = st->j, skip_fold = st->skip_fold, stop_stage = st->rp.stop_stage;
    const double phi_n = st->phi_n, phi_prop = st->phi_prop, ess_prev = st->ess_prev, target = st->rp.tempering_target;
    const double N = (double)st->rp.n_parts, e_center = st->e_center;
    // Energy sums + acceptance sum of the previous mutation: rows of ES = 32 doubles.  Thread t owns column t % 32 of the rows
    // t / 32, t / 32 + 32, ...: a wave-load reads two whole rows (512 B, no line is fetched twice - one wavefront per COLUMN
    // fetched every line eight times and was bandwidth-bound on the one CU); all loads of a thread are issued before the first
    // use, together with the scalar loads above.  The 32 row-groups are combined in fixed order through LDS.
    const bool try_es = esum_partials != nullptr && acc_nb > 0;
    if (try_es) {
        const int colx = threadIdx.x & 31, rg = threadIdx.x >> 5;
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.0;
        int b = rg;
        for (; b + 15 * 32 < es_nb; b += 16 * 32) {
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] += ld(esum_partials + (long long)(b + q * 32) * ES + colx);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (b + q * 32 < es_nb) a[q] += ld(esum_partials + (long long)(b + q * 32) * ES + colx);
        double v = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
                   (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
        v += __shfl_xor(v, 32, 64);                      // the wave's two row-groups
        if ((threadIdx.x & 63) < 32) scratch[(threadIdx.x >> 6) * 32 + colx] = v;
    }
    // largest energy the previous mutation (or k_energy_max) left: per-block / per-shard maxima -> this stage's energy shift
    double em = -__builtin_inf();
    if (emax_part)
        for (int b = threadIdx.x; b < emax_n; b += BT) em = fmax(em, ld(emax_part + b));
    if (done) return;
    SMCMI_STAMP(prof, 1);
    s_em[threadIdx.x] = em;
    const int i = stage0 + 1;
    double swv = 2.0;                                  // stays in flight across the reduction's loads; stored to LDS after them
    if (threadIdx.x < 64) {
        const int jj = j - 1 + (int)threadIdx.x;       // 0-based index of walk step threadIdx.x + 1
        if (!fixed && jj < n_phi) swv = sched[jj];
    }
    const double ph_fixed = (fixed && i <= n_phi) ? sched[i - 1] : 0.0;
    // Σ accept over blocks of the previous mutation (update_acceptance_rate!, src/particle.jl:466-468)
    const bool have_es = try_es && stage0 > 1 && !fixed && !skip_fold;
    __syncthreads();
    if (have_es && threadIdx.x < ES) {
        double tsum = 0.0;
#pragma unroll
        for (int g = 0; g < BT / 64; ++g) tsum += scratch[g * 32 + threadIdx.x];
        s_es[threadIdx.x] = tsum;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 128) {        // wave 1: 1024 -> 64 maxima
        const int c = threadIdx.x - 64;
        double m = s_em[c];
#pragma unroll
        for (int q = 1; q < BT / 64; ++q) m = fmax(m, s_em[c + 64 * q]);
        s_em2[c] = m;
    }
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 72) {         // 64 -> 8
        const int c = threadIdx.x - 64;
        double m = s_em2[c * 8];
#pragma unroll
        for (int q = 1; q < 8; ++q) m = fmax(m, s_em2[c * 8 + q]);
        s_em3[c] = m;
    }
    double asum = 0.0;
    if (!have_es && acc_nb > 0) asum = final_sum1(acc_partials, acc_nb, scratch);
    if (threadIdx.x < 64) s_sw[threadIdx.x] = swv;
    __syncthreads();
    SMCMI_STAMP(prof, 2);
    // Wavefront 0 finishes: the scalar bookkeeping is lane 0's, the candidate set is built by all lanes (no serial loops).
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    if (have_es) asum = s_es[EACC];
    if (lane == 0) {
        if (acc_nb > 0 && stage0 > 1 && !skip_fold) {
            const double a = asum / N;
            st->accept = a;
            rec.accept[stage0 - 1] = a;
        }
        if (rs) st->do_resample = 0;
        if (skip_fold) st->skip_fold = 0;
    }
    if (phi_n >= 1.0) { if (lane == 0) { st->done = 1; st->e_shift = 0.0; } return; }
    // intermediate save point: stage `stage0` is complete (its acceptance rate folded above); the host downloads what it wants
    // and continues the same chain with smcmi_run(continue_run = 1)
    if (stop_stage > 0 && stage0 >= stop_stage) { if (lane == 0) { st->done = 5; st->e_shift = 0.0; } return; }
    if (i > max_stages) { if (lane == 0) { st->err = SMCMI_ERR_CAPACITY; st->done = 1; } return; }
    Solver &S = st->sol[0];
    if (lane == 0) { st->stage = i; st->phi_prev = phi_n; S.unconverged = 0; }
    // host-mapped progress word (fixed schedules enqueued without selection kernels, smcmi_run): the host stays a bounded number of
    // stages ahead of this stage index, so a stage that resamples after all leaves few idle launches behind it
    if (host_note && lane == 0) __hip_atomic_store(host_note, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (emax_part && lane == 0) {
        double m = s_em3[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmax(m, s_em3[w]);
        if (fabs(m) < 1e300) st->e_shift = m;              // no live particle with a finite energy: keep the previous shift
    }
    if (fixed) {
        if (lane == 0) {
            st->phi_n = ph_fixed;
            S.phi_n = ph_fixed; S.mode = MODE_FINAL; S.j = j; S.phi_prop = phi_prop; S.n_valid = 0;
        }
        return;
    }
    double ess_now, ess_bar;   // ESS of the current weights = ESS(ϕ_n1)
    if (rl) { ess_bar = target * N; ess_now = N; }
    else { ess_bar = target * ess_prev; ess_now = ess_prev; }
    if (lane == 0) {
        if (rl) st->resampled_last = 0;
        S.ess_bar = ess_bar;
        S.lo = phi_n; S.phi0 = phi_n;
        S.glo = ess_now - ess_bar;
        S.hi = phi_prop; S.ghi = 0.0;
        S.j = j; S.phi_prop = phi_prop;
    }
    // predictor: root of the Taylor model (every lane computes the same scalar), then rings around it
    double pd = __longlong_as_double(0x7ff8000000000000ll), gp = pd;
    if (have_es) {
        pd = predict_delta_wave(s_es, rs != 0, ess_bar, &gp);
        const double ec = e_center + s_es[1] / s_es[0];        // weighted mean energy: centre for the next epilogue
        if (lane == 0 && fabs(ec) < 1e300) st->e_center = ec;
    }
    if (lane == 0) st->pred_delta = pd;
    SMCMI_STAMP(prof, 3);
    // Scan candidates, ascending: the reference's walk (step 0 = the current ϕ_prop, step q = schedule[j + q - 1], 1-based;
    // helpers.jl:29-32) merged with the ring points.  With a prediction only the steps around it are evaluated (ϕ_prop, the last
    // step below the prediction, the first two above it): ESS(ϕ) falls with ϕ, so the skipped steps in between keep it above
    // the target just like their neighbours.  Lanes 0..NRL-1 own the ring points, the next four the walk steps; each kept value
    // finds its slot by counting the kept values below it.
    const int q_end = n_phi - j + 1;                         // last existing walk step
    const double ph = phi_n + pd;
    bool use_pred = pd > 0.0 && ph < 1.0;
    // first walk step above the prediction among steps 0..62 (lane l looks at step l)
    const double wl = lane == 0 ? phi_prop : s_sw[lane - 1];
    const unsigned long long above = __ballot(lane <= 62 && lane <= q_end && wl > ph);
    int qstar = above ? (__ffsll((long long)above) - 1) : (q_end <= 62 ? q_end : -1);
    if (qstar < 0) use_pred = false;                         // prediction beyond the staged window: plain walk
    if (lane == 0) { S.spec = 0; S.gprime = gp; }
    if (spec_expected) {
        // Predict -> correct -> verify: the stage was enqueued WITHOUT a certificate pass.  ϕ_n is the root of the Taylor model
        // (accurate to ~1e-14 of the step with 16 power sums per weight order); the reference's schedule walk (helpers.jl:29-32:
        // advance ϕ_prop while ESS(ϕ_prop) >= target) is decided by comparing the walk steps with it - ESS falls with ϕ - and
        // k_prepare_mutation verifies the ESS the correction actually produced.  No usable prediction: stall, the host resumes
        // the stage with the certificate-pass path (candidates are prepared below as usual).
        const bool none_above = !above;
        // (a predicted root at or beyond 1 is the last stage: no walk step lies above it, ϕ_n = ϕ_prop = 1, and the verification is
        // ESS(1) >= ESS_bar - the stage that used to stall once per run)
        const bool beyond = pd > 0.0 && ph >= 1.0 && qstar >= 0 && none_above;
        if ((use_pred || beyond) && gp < 0.0) {
            const double step_q = __shfl(wl, qstar, 64);                 // first walk step above the prediction (or the last step)
            if (lane == 0) {
                S.mode = MODE_FINAL; S.spec = 1; S.n_valid = 0;
                S.phi_prop = step_q; S.j = j + qstar;
                S.phi_n = none_above ? step_q : ph;                      // walk exhausted with ESS still above the target: ϕ_n = ϕ_prop (= 1)
            }
            return;
        }
        if (lane == 0) st->done = 4;
    }
    double x = 0.0;
    int cq = -1;
    bool keep = false;
    if (use_pred) {
        if (lane < 2 * NPR + 1) {
            const double r = lane < NPR ? -PRING[lane < NPR ? lane : 0] : (lane == NPR ? 0.0 : PRING[2 * NPR - lane]);
            x = ph + pd * r;
            keep = x > phi_n && x < 1.0;
        }
        const unsigned long long ringm = __ballot(keep);
        if (!ringm) use_pred = false;
        const int nr = __popcll(ringm);
        if (lane == NRL) { cq = 0; keep = true; }
        if (lane == NRL + 1) { cq = qstar - 1; keep = cq > 0; }
        if (lane == NRL + 2) { cq = qstar; keep = cq > 0; }
        const int nq3 = 1 + (qstar - 1 > 0 ? 1 : 0) + (qstar > 0 ? 1 : 0);
        if (lane == NRL + 3) { cq = qstar + 1; keep = cq <= q_end && nq3 < KC - nr; }
        if (lane >= NRL && lane <= NRL + 3 && keep) x = cq == 0 ? phi_prop : s_sw[cq - 1];
    }
    if (!use_pred) {                                         // plain walk: steps 0 .. min(q_end, KC - 1)
        cq = lane; keep = lane < KC && lane <= q_end;
        x = keep ? wl : 0.0;
    } else {
        // drop ring points that coincide with a lower-lane ring point or with any walk step
        bool dup = false;
#pragma unroll
        for (int o = 0; o < NLANE; ++o) {
            const double xo = __shfl(x, o, 64);
            const bool ko = (bool)__shfl((int)keep, o, 64);
            if (ko && lane < NRL && xo == x && (o < lane || o >= NRL)) dup = true;
        }
        if (dup) keep = false;
    }
    int rank = 0;
#pragma unroll
    for (int o = 0; o < NLANE; ++o) {
        const double xo = __shfl(x, o, 64);
        const bool ko = (bool)__shfl((int)keep, o, 64);
        if (ko && xo < x) ++rank;
    }
    const int nv = __popcll(__ballot(keep));
    if (keep) { S.cand[rank] = x; S.cj[rank] = use_pred ? (lane >= NRL ? cq : -1) : cq; }
    if (lane == 0) { S.n_valid = nv; S.mode = MODE_SCAN; }
    SMCMI_STAMP(prof, 5);
}
#endif

// Inclusive scan of W̃/ΣW̃ over chunk `vb` (cumsum(weights ./ sum(weights)), src/resample.jl:29,47): thread t owns IPT
// consecutive items so the running sum follows particle order; `carry` = sum of the preceding chunks.
__device__ inline void scan_chunk(const double *w, long long n, int n_chunks, int vb, double carry, double total, double *cum,
                                  double *s_tot /* TB doubles of LDS */) {
    long long beg, end;
    block_chunk(n, n_chunks, vb, beg, end);
    constexpr int IPT = 4;
    for (long long base = beg; base < end; base += (long long)TB * IPT) {
        const long long i0 = base + (long long)threadIdx.x * IPT;
        double v[IPT], run = 0.0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) { v[k] = (i0 + k < end) ? w[i0 + k] : 0.0; run += v[k]; v[k] = run; }
        s_tot[threadIdx.x] = run;
        __syncthreads();
        for (int off = 1; off < TB; off <<= 1) {   // Hillis-Steele inclusive scan of the 256 thread totals
            const double add = (threadIdx.x >= off) ? s_tot[threadIdx.x - off] : 0.0;
            __syncthreads();
            s_tot[threadIdx.x] += add;
            __syncthreads();
        }
        const double excl = (threadIdx.x > 0 ? s_tot[threadIdx.x - 1] : 0.0) + carry;
#pragma unroll
        for (int k = 0; k < IPT; ++k)
            if (i0 + k < end) cum[i0 + k] = (excl + v[k]) / total;
        carry += s_tot[TB - 1];
        __syncthreads();
    }
}

// After the correction pass: ESS, log-MDD increment, resample decision, step-size adaptation
// (src/smc_main.jl:427-455, src/particle.jl:362-366); copies the solver's (ϕ_n, j, ϕ_prop) back to the loop scalars and,
// on resample stages, forms the exclusive prefix of the per-block weight sums for the resampling scan.
// Launched with one block (stand-alone / sharded callers, chunk offsets only) or with one block per weight chunk and `cum`
// set (smcmi_run): then every block recomputes the decision from the same partial sums (no mutable state is read for it),
// block 0 alone does the bookkeeping, and on resample stages each block scans its own chunk right here - the separate scan
// launch (a ~4 µs no-op on 95 % of the stages) disappears.
// Scalars of the post-correction bookkeeping, loaded by every thread up front (one memory round trip with the partial sums).
struct PostIn {
    int done, smode, i, jj, resamples;
    double N, a, tg, c0, thr, phi_n, phi_prop, logz, dlz;
};
__device__ inline PostIn post_load(const DevState *st, int sol_slot) {
    const Solver &S = st->sol[sol_slot];
    PostIn p;
    p.done = st->done; p.smode = S.mode; p.i = st->stage; p.jj = S.j; p.resamples = st->resamples;
    p.N = (double)st->rp.n_parts; p.a = st->accept; p.tg = st->rp.target; p.c0 = st->c; p.thr = st->rp.threshold;
    p.phi_n = S.phi_n; p.phi_prop = S.phi_prop; p.logz = st->logz;
    p.dlz = (S.phi_n - st->phi_prev) * stage_shift(st);          // log of the common factor the shifted weights left out
    return p;
}
// ESS, log-MDD increment, resample decision, step-size adaptation, records (src/smc_main.jl:427-455); one thread.
// Returns the resample decision (-1: ESS is NaN, run aborted).
__device__ inline int post_write(DevState *st, const Records &rec, const PostIn &p, double s1, double s2) {
    const double ess = s1 * s1 / s2;
    const bool bad = isnan(ess);
    const int rs = (!bad && ess < p.thr) ? 1 : 0;
    const double c1 = p.c0 * (0.95 + 0.10 * exp(16.0 * (p.a - p.tg)) / (1.0 + exp(16.0 * (p.a - p.tg))));
    st->phi_n = p.phi_n; st->phi_prop = p.phi_prop; st->j = p.jj;
    st->sumw = s1; st->sumw2 = s2; st->ess = ess; st->ess_prev = ess;
    rec.phi[p.i - 1] = p.phi_n;
    rec.ess[p.i - 1] = ess;
    if (bad) { st->err = SMCMI_ERR_NAN_ESS; st->done = 1; return -1; }     // check_nan_ess, helpers.jl:270-305
    st->logz = p.logz + (log(s1 / p.N) + p.dlz);
    st->do_resample = rs;
    rec.resampled[p.i - 1] = rs;
    if (rs) { st->resamples = p.resamples + 1; st->resampled_last = 1; }
    st->c = c1;
    rec.c[p.i - 1] = c1;
    return rs;
}

#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_post_correct(DevState *st, const double *partials, int nb, double *chunk_off,
                                                     Records rec, int sol_slot, CloudPtrs cl = CloudPtrs{}, double *cum = nullptr) {
    __shared__ double scratch[TB];
    __shared__ double s_tot[2];
    __shared__ int s_rs;
    const PostIn pin = post_load(st, sol_slot);
    const int done = pin.done, smode = pin.smode;
    const double thr = pin.thr;
    const double v = final_sum(partials, nb, 2, scratch);
    if (done) return;
    if (smode != MODE_FINAL) {       // cannot happen unless a pass kernel flagged an error
        if (threadIdx.x == 0) { st->err = SMCMI_ERR_BRACKET; st->done = 1; }
        return;
    }
    if (threadIdx.x < 2) s_tot[threadIdx.x] = v;
    if (threadIdx.x == 0) s_rs = 0;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x != 0) {          // other blocks: the decision only
        const double ess = s_tot[0] * s_tot[0] / s_tot[1];
        s_rs = (!isnan(ess) && ess < thr) ? 1 : 0;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int rs = post_write(st, rec, pin, s_tot[0], s_tot[1]);
        if (rs > 0) s_rs = 1;
    }
    __syncthreads();
    if (s_rs && (chunk_off || cum)) {
        // exclusive prefix of partials[2 b] in block order: 256 threads x 4 blocks each, then a sequential carry
        const int t = threadIdx.x;
        double loc[4], run = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int b = t * 4 + q; loc[q] = run; run += (b < nb) ? partials[2 * (long long)b] : 0.0; }
        scratch[t] = run;
        __syncthreads();
        if (t == 0) { double carry = 0.0; for (int q = 0; q < TB; ++q) { const double x = scratch[q]; scratch[q] = carry; carry += x; } }
        __syncthreads();
        if (cum) {
            // fused selection scan: this block's chunk(s); offsets come from the prefix every block just rebuilt identically
            __shared__ double s_off[4 * TB];
#pragma unroll
            for (int q = 0; q < 4; ++q) s_off[t * 4 + q] = scratch[t] + loc[q];
            __syncthreads();
            const double *w = col(cl, 0, cl.R - 1);
            const double total = s_tot[0];
            for (int vb = blockIdx.x; vb < nb; vb += gridDim.x) scan_chunk(w, cl.n, nb, vb, s_off[vb], total, cum, scratch);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int b = t * 4 + q; if (b < nb) chunk_off[b] = scratch[t] + loc[q]; }
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------ resampling
// Inclusive scan of W̃/ΣW̃ (cumsum(weights ./ sum(weights)), src/resample.jl:29,47) in the block chunks of the
// correction pass; chunk offsets come from k_post_correct.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_scan_weights(CloudPtrs cl, const DevState *st, const double *chunk_off,
                                                     double *cum, int force, int n_chunks) {
    __shared__ double s_tot[TB];
    if (!force && (st->done || !st->do_resample)) return;
    const double *w = col(cl, st->cur, cl.R - 1);
    const double total = st->sumw;
    for (int vb = blockIdx.x; vb < n_chunks; vb += gridDim.x) scan_chunk(w, cl.n, n_chunks, vb, chunk_off[vb], total, cum, s_tot);
}
#endif

// Selection for output slot k: ancestor = first j with cum[j] > thr (src/resample.jl:51-70 systematic walk, :33-41
// multinomial findfirst; fall-through, which the reference turns into index 0 / nothing and is reachable only by
// round-off, clamps to the last index), then cloud.particles = particles[new_inds, :] and reset_weights!
// (src/smc_main.jl:440-442): the thread copies its ancestor's R-1 columns into the other cloud buffer and sets W = 1.
// full != nullptr: rows come from an all-gathered n_cum x R cloud (multi-GPU) and go to the current buffer.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_resample_gather(CloudPtrs cl, const DevState *st, const double *cum, long long n_cum,
                                                        long long slot0, long long n_parts_total, int method,
                                                        unsigned long long seed, unsigned stage, const double *offsets,
                                                        long long *anc, const double *full, int force, long long full_shard_n = 0,
                                                        int dst_buf = -1) {
    if (!force && (st->done || !st->do_resample)) return;
    if (!force) stage = (unsigned)st->stage;
    for (long long k = (long long)blockIdx.x * TB + threadIdx.x; k < cl.n; k += (long long)gridDim.x * TB) {
    const long long slot = slot0 + k;
    double ua, ub;
    if (method == SMCMI_RESAMPLE_MULTINOMIAL) {
        if (offsets) ua = offsets[slot];
        else uniform_pair(seed, (unsigned long long)slot, stage, rng_tag(P_RES, 0, 0), ua, ub);
    } else {
        if (offsets) ua = offsets[0];
        else uniform_pair(seed, 0ull, stage, rng_tag(P_RES, 0, 0), ua, ub);
        ua = ((double)slot + ua) / (double)n_parts_total;       // (i - 1 + offset) / n_parts
    }
    long long lo = 0, hi = n_cum;   // upper_bound: first index with cum > thr
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] > ua) hi = mid; else lo = mid + 1;
    }
    const long long a = lo < n_cum ? lo : n_cum - 1;
    if (anc) anc[k] = a;
    const int src = st->cur, dst = dst_buf >= 0 ? dst_buf : (full ? src : (src ^ 1)), R = cl.R;
    // source row: the local buffer, an n_cum x R column-major cloud, or (full_shard_n > 0) the concatenation of per-rank
    // [R][full_shard_n] shard buffers as an all-gather delivers them
    const double *from = full ? full : cl.buf[src];
    long long ldf = full ? n_cum : cl.n, a_row = a;
    if (full && full_shard_n > 0) {
        from = full + (a / full_shard_n) * (long long)R * full_shard_n;
        ldf = full_shard_n;
        a_row = a % full_shard_n;
    }
    for (int c0 = 0; c0 < R - 1; c0 += 8) {          // 8 indexed loads in flight, then the coalesced stores
        double tmp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) tmp[q] = (c0 + q < R - 1) ? from[(long long)(c0 + q) * ldf + a_row] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (c0 + q < R - 1) col(cl, dst, c0 + q)[k] = tmp[q];
    }
    col(cl, dst, R - 1)[k] = 1.0;
    }
}
#endif

// Bridge resampling of a tempered update (src/smc_main.jl:266-279): n_out rows of the old cloud `src` are drawn by its
// weights (cum = cumsum(weights ./ sum(weights)) over the n_src old particles) and written to rows [0, n_out) of `dst`
// WITH their old weights (update_cloud! copies whole rows; the weights are only reset after the second resample, :322).
// Systematic search range is start_ind:n_parts of the *output* length (resample.jl:54, quirk Q5) -> lim = min(n_out, n_src).
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_bridge_gather(CloudPtrs src, int src_buf, const double *cum, long long n_src,
                                                      CloudPtrs dst, int dst_buf, long long n_out, int method,
                                                      unsigned long long seed, unsigned stage, const double *offsets,
                                                      long long *anc) {
    const long long k = (long long)blockIdx.x * TB + threadIdx.x;
    if (k >= n_out) return;
    double ua, ub;
    long long lim = n_src;
    if (method == SMCMI_RESAMPLE_MULTINOMIAL) {
        if (offsets) ua = offsets[k];
        else uniform_pair(seed, (unsigned long long)k, stage, rng_tag(P_RES, 0, 0), ua, ub);
    } else {
        if (offsets) ua = offsets[0];
        else uniform_pair(seed, 0ull, stage, rng_tag(P_RES, 0, 0), ua, ub);
        ua = ((double)k + ua) / (double)n_out;
        lim = n_out < n_src ? n_out : n_src;
    }
    long long lo = 0, hi = lim;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] > ua) hi = mid; else lo = mid + 1;
    }
    const long long a = lo < lim ? lo : lim - 1;
    if (anc) anc[k] = a;
    for (int c = 0; c < src.R; ++c) col(dst, dst_buf, c)[k] = col(src, src_buf, c)[a];
}
#endif

// zero_bad_loglh_weights! (src/particle.jl:392-396): weight 0 where loglh == -Inf
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_zero_bad_weights(CloudPtrs cl, const DevState *st) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= cl.n) return;
    const int d = cl.R - 5;
    if (col(cl, st->cur, d)[i] == SMCMI_NEG_INF) col(cl, st->cur, cl.R - 1)[i] = 0.0;
}
#endif
// normalize_weights! (src/particle.jl:362-366): W *= n_parts, W /= sum(W); st->sumw holds the fixed-order sum
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_normalize_weights(CloudPtrs cl, const DevState *st, double n_parts) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= cl.n) return;
    double *w = col(cl, st->cur, cl.R - 1);
    w[i] = (w[i] * n_parts) / st->sumw;
}
#endif

// Systematic resampling, sharded: ancestor (global row) of the first and of the last output slot of every shard r - the rows a
// shard must receive form the contiguous range [out[2r], out[2r+1]] (thresholds ascend with the slot).  Same threshold and
// upper_bound as k_resample_gather.  out[0] = -1 when this stage does not resample.  One wavefront, lane r = shard r.
#ifndef SMCMI_INST_UNIT
static __global__ void k_anc_ranges(const DevState *st, const double *cum, long long N, long long n_local, int world, unsigned long long seed,
                             long long *out) {
    const int r = threadIdx.x;
    if (st->done || !st->do_resample) { if (r == 0) out[0] = -1; return; }
    if (r >= world) return;
    double ua, ub;
    uniform_pair(seed, 0ull, (unsigned)st->stage, rng_tag(P_RES, 0, 0), ua, ub);
    for (int e = 0; e < 2; ++e) {
        const long long slot = e == 0 ? (long long)r * n_local : (long long)(r + 1) * n_local - 1;
        const double thr = ((double)slot + ua) / (double)N;
        long long lo = 0, hi = N;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (cum[mid] > thr) hi = mid; else lo = mid + 1;
        }
        out[2 * r + e] = lo < N ? lo : N - 1;
    }
}
#endif

// ------------------------------------------------------------------------------------------------ moments
// One pass over (θ, W̃): normalise the weights (normalize_weights!, src/particle.jl:362-366: W*N then /ΣW; or 1 after a
// resample), write them to the weight column and the W history, and accumulate the augmented second-moment matrix
// Σ w x̃ x̃ᵀ, x̃ = (1, θ - shift), from which weighted_mean / weighted_cov follow (src/particle.jl:481-483, 526-529).
// Particles are staged through LDS in tiles so every (a,b) pair is accumulated from on-chip data.
constexpr int MT = 256;                      // particles per LDS tile
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_moments(CloudPtrs cl, DevState *st, double *partials, double *hist_W,
                                                long long hist_ld, int standalone) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (!standalone && st->done) return;
    const int d = cl.R - 5, da = d + 1, npairs = da * (da + 1) / 2;
    const int ldx = MT + 1;                  // padded row: pair threads reading different rows hit different banks
    double *xs = sm;                         // da rows: row 0 = 1, row a+1 = θ_a - shift_a
    double *wv = xs + (long long)da * ldx;   // weights of the tile
    unsigned char *pa = (unsigned char *)(wv + ldx), *pb = pa + npairs;
    for (int p = threadIdx.x; p < npairs; p += TB) {   // decode pair index -> (a <= b)
        int a = 0, rem = p;
        while (rem >= da - a) { rem -= da - a; ++a; }
        pa[p] = (unsigned char)a; pb[p] = (unsigned char)(a + rem);
    }
    const int resampled = standalone ? 0 : st->do_resample;
    const int src = resampled ? 1 : 0;       // a resampled cloud was gathered into buffer 1; it is copied back to buffer 0 here
    double *w = col(cl, 0, cl.R - 1);
    const double N = (double)st->rp.n_parts, sumw = st->sumw;
    const int stage_col = st->stage - 1;
    const bool hist = !standalone && st->rp.store_history && hist_W != nullptr;
    const int slices = npairs >= TB ? 1 : TB / npairs;
    constexpr int RMAX = (NPAIR_MAX + TB - 1) / TB;
    double acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.0;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    for (long long base = beg; base < end; base += MT) {
        __syncthreads();
        const long long i = base + threadIdx.x;
        double wi = 0.0;
        if (i < end) {
            if (standalone) wi = w[i];
            else {
                wi = resampled ? 1.0 : (w[i] * N) / sumw;
                w[i] = wi;
                if (hist) hist_W[(long long)stage_col * hist_ld + i] = wi;
            }
        }
        wv[threadIdx.x] = wi;
        xs[threadIdx.x] = 1.0;
        for (int a = 0; a < d; ++a) {
            const double th = (i < end) ? col(cl, src, a)[i] : 0.0;
            if (resampled && i < end) col(cl, 0, a)[i] = th;
            xs[(a + 1) * ldx + threadIdx.x] = (i < end) ? th - st->shift[a] : 0.0;
        }
        if (resampled && i < end)
            for (int c = d; c < d + 4; ++c) col(cl, 0, c)[i] = col(cl, 1, c)[i];
        __syncthreads();
        if (slices > 1) {
            const int p = threadIdx.x % npairs, s = threadIdx.x / npairs;
            if (s < slices) {
                const double *xa = xs + pa[p] * ldx, *xb = xs + pb[p] * ldx;
                for (int q = s; q < MT; q += slices) acc[0] += wv[q] * xa[q] * xb[q];
            }
        } else {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int p = threadIdx.x + r * TB;
                if (p < npairs) {
                    const double *xa = xs + pa[p] * ldx, *xb = xs + pb[p] * ldx;
                    double s = 0.0;
                    for (int q = 0; q < MT; ++q) s += wv[q] * xa[q] * xb[q];
                    acc[r] += s;
                }
            }
        }
    }
    __syncthreads();
    double *out = partials + (long long)blockIdx.x * npairs;
    if (slices > 1) {
        double *sl = xs;   // reuse
        const int p = threadIdx.x % npairs, s = threadIdx.x / npairs;
        if (s < slices) sl[s * npairs + p] = acc[0];
        __syncthreads();
        if (threadIdx.x < npairs) {
            double t = 0.0;
            for (int s2 = 0; s2 < slices; ++s2) t += sl[s2 * npairs + threadIdx.x];
            out[threadIdx.x] = t;
        }
    } else {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int p = threadIdx.x + r * TB;
            if (p < npairs) out[p] = acc[r];
        }
    }
}
#endif

// Register-resident variant for d <= 12: one thread streams particles (coalesced 8-byte column reads) and keeps all
// (d+1)(d+2)/2 augmented pair sums in VGPRs with compile-time indices - no LDS traffic in the main loop, 1 FMA per pair
// per particle, so the pass is HBM-bound.  Same outputs (normalised weights, W history, per-block pair partials in the
// same (a <= b) order) as k_moments.
template <int D>
__global__ void __launch_bounds__(TB) k_moments_reg(CloudPtrs cl, DevState *st, double *partials, double *hist_W,
                                                    long long hist_ld, int standalone, const double *fin_partials = nullptr,
                                                    int nb_fin = 0, int sol_slot = 0, Records rec = Records{}) {
    constexpr int DA = D + 1, NP = DA * (DA + 1) / 2;
    constexpr int NCH = (NP + 63) / 64;                  // chunks of 64 accumulators for the block reduction
    __shared__ double red[(TB / 64) * 64];
    // fin_partials != nullptr: the stage was enqueued WITHOUT k_post_correct / k_resample_gather because the host expects no
    // resampling (smcmi_run).  Every block re-derives the post-correction decision from the correction partials (same fixed-order
    // sums as k_post_correct), block 0 does the bookkeeping; if the decision is to resample after all, nothing is written and the
    // run stalls (done = 3) until the host enqueues the selection path for this stage.
    double sumw_f = 0.0;
    if (fin_partials) {
        __shared__ double s_fin[2];
        const PostIn pin = post_load(st, sol_slot);
        const double v = final_sum(fin_partials, nb_fin, 2, red);
        if (pin.done) return;
        if (pin.smode != MODE_FINAL) {
            if (threadIdx.x == 0 && blockIdx.x == 0) { st->err = SMCMI_ERR_BRACKET; st->done = 1; }
            return;
        }
        if (threadIdx.x < 2) s_fin[threadIdx.x] = v;
        __syncthreads();
        const double s1 = s_fin[0], s2 = s_fin[1];
        const double ess = s1 * s1 / s2;
        if (!isnan(ess) && ess < pin.thr) {              // selection needed: stall, the host resumes with the full path
            if (threadIdx.x == 0 && blockIdx.x == 0) st->done = 3;
            return;
        }
        if (threadIdx.x == 0 && blockIdx.x == 0) post_write(st, rec, pin, s1, s2);
        if (isnan(ess)) return;
        sumw_f = s1;
        __syncthreads();
    }
    if (!standalone && !fin_partials && st->done) return;
    const int resampled = (standalone || fin_partials) ? 0 : st->do_resample;
    const int src = resampled ? 1 : 0;       // a resampled cloud was gathered into buffer 1; it is copied back to buffer 0 here
    double *w = col(cl, 0, cl.R - 1);
    const double N = (double)st->rp.n_parts, sumw = fin_partials ? sumw_f : st->sumw;
    const int stage_col = st->stage - 1;
    const bool hist = !standalone && st->rp.store_history && hist_W != nullptr;
    double sh[D];
#pragma unroll
    for (int a = 0; a < D; ++a) sh[a] = st->shift[a];
    double acc[NCH * 64];
#pragma unroll
    for (int p = 0; p < NCH * 64; ++p) acc[p] = 0.0;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    for (long long i0 = beg + threadIdx.x; i0 < end; i0 += 2 * TB) {
        // two particles per trip: both particles' column loads are issued before any arithmetic (each trip is otherwise one
        // exposed ~1 µs memory round trip at this occupancy)
        double xx[2][DA], ww[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = i0 + (long long)u * TB;
            const bool in = i < end;
            ww[u] = in ? w[i] : 0.0;
            xx[u][0] = 1.0;
#pragma unroll
            for (int a = 0; a < D; ++a) xx[u][a + 1] = in ? col(cl, src, a)[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = i0 + (long long)u * TB;
            if (i >= end) continue;
            double wi = ww[u];
            if (!standalone) {
                wi = resampled ? 1.0 : (wi * N) / sumw;
                w[i] = wi;
                if (hist) hist_W[(long long)stage_col * hist_ld + i] = wi;
            }
            if (resampled) {
                double meta[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) meta[c] = col(cl, 1, D + c)[i];
#pragma unroll
                for (int a = 0; a < D; ++a) col(cl, 0, a)[i] = xx[u][a + 1];
#pragma unroll
                for (int c = 0; c < 4; ++c) col(cl, 0, D + c)[i] = meta[c];
            }
#pragma unroll
            for (int a = 0; a < D; ++a) xx[u][a + 1] -= sh[a];
            int p = 0;
#pragma unroll
            for (int a = 0; a < DA; ++a) {
                const double wx = wi * xx[u][a];
#pragma unroll
                for (int b = a; b < DA; ++b) { acc[p] += wx * xx[u][b]; ++p; }
            }
        }
    }
    double *out = partials + (long long)blockIdx.x * NP;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double a64[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) a64[q] = acc[ch * 64 + q];
        const double tot = block_reduce_many<64>(a64, red);
        if (threadIdx.x < 64 && ch * 64 + (int)threadIdx.x < NP) out[ch * 64 + threadIdx.x] = tot;
    }
}

// Fixed-order reduction of the moment partials: totals[p] for the (d+1)(d+2)/2 pairs, 1024 threads per block:
// thread (s, idx) sums blocks b = s (mod 16) of pair p0 + idx; grid = ceil(npairs / 64).
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(1024) k_moments_reduce(const DevState *st, const double *partials, int nb, int npairs,
                                                         double *totals, int standalone) {
    __shared__ double scratch[1024];
    if (!standalone && st->done) return;
    const int p0 = blockIdx.x * 64, m = (npairs - p0) < 64 ? (npairs - p0) : 64;
    const int t = threadIdx.x, idx = t % 64, s = t / 64;
    double a0 = 0.0, a1 = 0.0;
    if (idx < m) {
        int b = s;
        for (; b + 16 < nb; b += 32) { a0 += partials[(long long)b * npairs + p0 + idx]; a1 += partials[(long long)(b + 16) * npairs + p0 + idx]; }
        if (b < nb) a0 += partials[(long long)b * npairs + p0 + idx];
    }
    scratch[s * 64 + idx] = a0 + a1;
    __syncthreads();
    if (t < m) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += scratch[q * 64 + t];
        totals[p0 + t] = tot;
    }
}
#endif

// totals of the augmented pair sums -> θ_bar (st->mean), R (st->cov); the shift moves to the new mean
__device__ inline void moments_from_totals(DevState *st, const double *totals, int d, int t, int nt) {
    const int da = d + 1;
    const double sw = totals[0];
    for (int a = t; a < d; a += nt) st->mean[a] = st->shift[a] + totals[a + 1] / sw;
    for (int e = t; e < d * d; e += nt) {
        int a = e / d, b = e % d;
        if (a > b) { const int tmp = a; a = b; b = tmp; }
        const int ra = a + 1, rb = b + 1;
        const int p = ra * da - ra * (ra - 1) / 2 + (rb - ra);
        const double m1a = totals[a + 1] / sw, m1b = totals[b + 1] / sw;
        st->cov[e] = totals[p] / sw - m1a * m1b;
    }
    __syncthreads();
    for (int a = t; a < d; a += nt) st->shift[a] = st->mean[a];
}
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(64) k_finalize_moments(DevState *st, const double *totals, int d) {
    moments_from_totals(st, totals, d, threadIdx.x, 64);
}
#endif

// generic fixed-order reduction of block partials into out[0..m) (1 block)
// Optional: also reduce the per-block energy maxima of this shard into its slot of `emax_slots` (the other shards' slots get 0:
// the sum all-reduce that follows then is a gather of the maxima, k_stage_begin takes the largest).
__device__ inline void emax_publish(const double *emax_part, int nb, double *emax_slots, int rank, int world, double *smem) {
    double em = -__builtin_inf();
    for (int b = threadIdx.x; b < nb; b += TB) em = fmax(em, emax_part[b]);
    em = block_max(em, smem, TB / 64);
    if ((int)threadIdx.x < world) emax_slots[threadIdx.x] = ((int)threadIdx.x == rank) ? fmax(em, -1e300) : 0.0;
}
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_reduce_partials(const double *partials, int nb, int m, double *out, const double *emax_part = nullptr,
                                                        int emax_nb = 0, double *emax_slots = nullptr, int rank = 0, int world = 1) {
    __shared__ double scratch[TB];
    __shared__ double smem[TB / 64];
    if (emax_part) emax_publish(emax_part, emax_nb, emax_slots, rank, world, smem);
    if (m == 0) return;                  // (the maxima alone: the sharded driver's run start)
    if (m == 1) {
        const double tot = final_sum1(partials, nb, scratch);
        if (threadIdx.x == 0) out[0] = tot;
        return;
    }
    const double tot = final_sum(partials, nb, m, scratch);
    if (threadIdx.x < m) out[threadIdx.x] = tot;
}
#endif

// First level of a two-level row reduction for very many partial rows (N >= ~3e5: one row per 256 particles): block g totals the
// rows of its contiguous chunk in fixed order -> out[g][m]; the consumer then totals gridDim.x rows instead of nb.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_reduce_rows(const double *partials, int nb, int m, double *out, const double *emax_in = nullptr,
                                                    double *emax_out = nullptr) {
    __shared__ double scratch[TB];
    __shared__ double smem[TB / 64];
    const int per = (nb + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, rows = (r0 + per <= nb) ? per : (nb > r0 ? nb - r0 : 0);
    const double tot = final_sum(partials + (long long)r0 * m, rows, m, scratch);
    if (threadIdx.x < m) out[(long long)blockIdx.x * m + threadIdx.x] = tot;
    if (emax_in) {                      // the chunk's energy maximum rides along (one value per row)
        double em = -__builtin_inf();
        for (int b = threadIdx.x; b < rows; b += TB) em = fmax(em, emax_in[r0 + b]);
        em = block_max(em, smem, TB / 64);
        if (threadIdx.x == 0) emax_out[blockIdx.x] = em;
    }
}
#endif

// per-chunk weight sums in the layout k_post_correct / k_scan_weights expect (partials[2 b])
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_weight_chunk_sums(CloudPtrs cl, const DevState *st, double *partials) {
    __shared__ double red[(TB / 64) * 2];
    const double *w = col(cl, st->cur, cl.R - 1);
    double acc[2] = {0.0, 0.0};
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    for (long long i = beg + threadIdx.x; i < end; i += TB) { const double v = w[i]; acc[0] += v; acc[1] += v * v; }
    const double tot = block_reduce_many<2>(acc, red);
    if (threadIdx.x < 2) partials[2 * (long long)blockIdx.x + threadIdx.x] = tot;
}
#endif
#ifndef SMCMI_INST_UNIT
static __global__ void k_chunk_offsets(DevState *st, const double *partials, int nb, double *chunk_off, double base,
                                int set_sum) {
    double run = base;
    for (int b = 0; b < nb; ++b) { chunk_off[b] = run; run += partials[2 * (long long)b]; }
    if (set_sum) st->sumw = run;
}
#endif

// θ_bar, R from the moment partials; free subset + symmetrisation (src/smc_main.jl:457-465); random blocks
// (generate_free_blocks/all_blocks, src/helpers.jl:215-260, Fisher-Yates on Philox); then per block the scaled
// covariance c²Σ_b and its Cholesky factor - done ONCE per stage instead of per particle (src/mutation.jl:81,
// src/helpers.jl:90-94,135-155).  One block of 256 threads; all matrix work happens in LDS, DevState is written once.
// from_totals: 0 = use st->mean / st->cov as given (stand-alone mutation), 1 = reduce `partials` (nb blocks x npairs,
// fixed order) first, 2 = `partials` already holds the npairs totals.
// Cholesky of a db x db matrix (db <= DBM) held one row per lane in registers: column j's pivot and multipliers are
// broadcast with v_readlane (no LDS round trips); same k-ascending subtraction order per entry as the oracle's loop.
// Returns false when a pivot is not positive.  On return lane i holds row i of L in r[0..i].
// RSQ = true (engine 2): the pivot's reciprocal square root (hardware estimate + two Newton steps, ~1 ulp) replaces the
// correctly rounded sqrt and division - the column's critical path is a third as long (the factorisation is ten dependent
// columns, every block of a launch waits for it); L differs from the oracle's in the last bit or two.
// FULL = true: db == DBM, no per-column run-time test.
template <int DBM, bool RSQ = false, bool FULL = false>
__device__ inline bool chol_rows_in_regs(double (&r)[DBM], int db, int lane) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < DBM; ++j) {
        if (FULL || j < db) {
            const double ajj = bcast_lane(r[j], j);
            if (!(ajj > 0.0)) ok = false;
            double ljj, lij;
            if constexpr (RSQ) {
                double y = __builtin_amdgcn_rsq(ajj);               // ~1e-8 relative
                const double h0 = 0.5 * ajj;
                y = y * (1.5 - h0 * y * y);                         // Newton: y <- y (3 - a y²) / 2
                y = y * (1.5 - h0 * y * y);
                ljj = ajj * y;
                lij = (lane == j) ? ljj : r[j] * y;
            } else {
                ljj = sqrt(ajj);
                lij = (lane == j) ? ljj : r[j] / ljj;               // column j of L (valid for lanes >= j)
            }
#pragma unroll
            for (int k = j + 1; k < DBM; ++k) {
                const double lkj = bcast_lane(lij, k);
                r[k] -= lij * lkj;                                   // meaningful for lanes i >= k, k < db
            }
            r[j] = lij;
        }
    }
    return ok;
}

constexpr int PT = 1024;  // threads of the single prepare block

// Random numbers of one stage's mutation, generated AHEAD of it: the draws depend only on (seed, particle id, stage, step), so
// while block 0 of k_prepare_mutation does its serial set-up on one CU the rest of the chip is idle - blocks 1.. of the same
// launch fill it with the Philox + Box-Muller work (~40 % of the mutation kernel's instructions), which the mutation kernel
// then just loads.  Layout: zbuf[(t ZS + slot) n + i], t = mh_step n_blocks + block, ZS = D + 2 slots: MH uniform, mixture
// uniform, D normals (zero beyond the block length).  Same tags and the same expressions as the in-kernel path -> same bits.
constexpr int RA_T = 256;
constexpr int RA_SKIP = 16;     // blocks 1..15 of k_prepare_mutation draw nothing (block 0 prepares, the others may total rows): see rng_ahead_block

struct RngAhead {
    double *zbuf;            // null: disabled
    long long n, gid0;
    int D;
    int t_ahead;             // proposals t < t_ahead are drawn here, the others inside the mutation kernel (see ensure_zbuf)
};
__device__ inline void rng_ahead_block(const DevState *st, const ModelDev *md, unsigned long long seed, const RngAhead &ra) {
    // RA_T particles per block although the launch has PT threads per block (block 0 needs them): with one proposal per particle
    // 4 wavefronts per CU spread the draws over the whole chip instead of 16 per CU on a quarter of it
    // chunk c of RA_T particles is drawn by block c + RA_SKIP: workgroups go round-robin over the 8 XCDs, so the block that draws a
    // chunk sits on the XCD whose L2 the mutation block c (same chunking, blockIdx = c) will read the numbers from
    if (blockIdx.x < RA_SKIP) return;
    // the block's other wavefronts take the other (MH step, parameter block) proposals of the same particles: with 3 MH steps
    // (config 4) 12 wavefronts per block draw 870 instructions each instead of 4 drawing 2 610 - the draws are independent
    // functions of (seed, particle, stage, proposal), so who draws them changes no bit
    const int tq = (int)(threadIdx.x / RA_T), tn = (int)(blockDim.x / RA_T);
    const long long i = (long long)(blockIdx.x - RA_SKIP) * RA_T + (threadIdx.x % RA_T);
    if (i >= ra.n) return;
    const int nf = md->n_free, nb = st->rp.n_blocks, n_steps = st->rp.n_mh_steps, D = ra.D, ZS = D + 2;
    const unsigned stage = (unsigned)st->stage;
    const unsigned long long pid = (unsigned long long)(ra.gid0 + i);
    const int sub = (nf + nb - 1) / nb;
    const int t_end = n_steps * nb < ra.t_ahead ? n_steps * nb : ra.t_ahead;
    for (int tt = tq; tt < t_end; tt += tn) {
            const unsigned t = (unsigned)tt;
            const int b = tt % nb;
            const int db = (b < nb - 1) ? sub : nf - sub * (nb - 1);
            double step_prob, u_dummy, uc, unext;
            if (t == 0) uniform_pair(seed, pid, stage, rng_tag(P_MUT, 0xFFFFFu, 0), step_prob, u_dummy);
            else uniform_pair(seed, pid, stage, rng_tag(P_MUT, t - 1, 0), u_dummy, step_prob);
            uniform_pair(seed, pid, stage, rng_tag(P_MUT, t, 0), uc, unext);
            double *zt = ra.zbuf + (long long)t * ZS * ra.n + i;
            zt[0] = step_prob;
            zt[ra.n] = uc;
            for (int q = 0; 2 * q < D; ++q) {
                double z0 = 0.0, z1 = 0.0;
                if (2 * q < db) {
                    double ua, ub, sn, cs;
                    uniform_pair(seed, pid, stage, rng_tag(P_MUT, t, 1 + q), ua, ub);
                    const double rr = bx_sqrt(bx_neg2log(ua));
                    bx_sincos2pi(ub, &sn, &cs);
                    z0 = rr * cs;
                    z1 = (2 * q + 1 < db) ? rr * sn : 0.0;
                }
                zt[(long long)(2 + 2 * q) * ra.n] = z0;
                if (2 * q + 1 < D) zt[(long long)(3 + 2 * q) * ra.n] = z1;
            }
        }
}

#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(PT) k_prepare_mutation(DevState *st, const ModelDev *md, const double *partials, int nb_part,
                                                         unsigned long long seed, int from_totals, int gen_blocks,
                                                         int standalone, long long *prof = nullptr, RngAhead ra = RngAhead{}, int sol_slot = 0,
                                                         Records rec = Records{}, PrepRed pr = PrepRed{}, int *host_note = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    __shared__ double scratch[PT];
    if (blockIdx.x > 0) {
        if (blockIdx.x <= PREP_G) {          // blocks 1..PREP_G total a chunk of the rows each (see PrepRed)
            if (pr.rows && (standalone || !st->done)) {
                const int dd = md->d + 1, np = dd * (dd + 1) / 2;
                prep_reduce_block(pr, partials, nb_part, from_totals == 3 ? np + 2 : np, scratch);
            }
            return;
        }
        // the idle CUs draw the mutation's random numbers (see RngAhead)
        if (ra.zbuf && !st->done) rng_ahead_block(st, md, seed, ra);
        return;
    }
    SMCMI_STAMP(prof, 0);
    __shared__ double mu_f[MAXD];
    __shared__ int bfree[MAXD], bptr[MAXD + 1], fi[MAXD], fi_j[MAXD];
    __shared__ int s_fail;
    if (!standalone && st->done) return;
    const int d = md->d, nf = md->n_free, t = threadIdx.x, da = d + 1, npairs = da * (da + 1) / 2;
    double *tot = psm;                       // [npairs] (rounded up to 64)
    double *covl = tot + ((npairs + 63) / 64) * 64;   // [d*d]
    double *sig_f = covl + d * d;            // [nf*nf]
    double *A = sig_f + nf * nf;             // [nf*nf] scaled block covariance
    double *Ls = A + nf * nf;                // [nf*nf] factor of the current block
    if (t == 0) s_fail = 0;
    if (t < nf) fi[t] = md->free_inds[t];
    double c = st->c;
    SMCMI_STAMP(prof, 1);
    if (from_totals == 3) {
        // `partials` are the rows of k_correct_moments: [ΣW̃, ΣW̃², pair sums].  This block first takes the post-correction
        // decision k_post_correct would take (the stage was enqueued without it); if selection is needed after all the run
        // stalls (done = 3) and the host resumes the stage with the full path.
        __shared__ double s_cm[2];
        __shared__ double s_c;
        __shared__ int s_go;
        const PostIn pin = post_load(st, sol_slot);
        const int npf = npairs + 2;
        double v;
        if (pr.rows) {
            if (!prep_group_total(pr, npf, &v)) { if (t == 0) { st->err = SMCMI_ERR_TIMEOUT; st->done = 1; } return; }
        } else v = final_sum(partials, nb_part, npf, scratch);
        if (t < 2) s_cm[t] = v;
        else if (t < npf) tot[t - 2] = v;
        if (t == 0) s_go = 0;
        __syncthreads();
        if (t == 0) {
            if (pin.smode != MODE_FINAL) { st->err = SMCMI_ERR_BRACKET; st->done = 1; }
            else {
                const double ess = s_cm[0] * s_cm[0] / s_cm[1];
                const Solver &Sv = st->sol[sol_slot];
                bool verified = true;
                if (Sv.spec) {
                    // the correction ran at the PREDICTED ϕ_n: accept it only if the ESS it produced puts the true root within
                    // phi_rtol (|ESS - ESS_bar| / |dESS/dϕ| <= rtol ϕ_n), or, for ϕ_n = 1 by exhaustion, ESS >= ESS_bar
                    if (Sv.phi_n < 1.0) verified = fabs(ess - Sv.ess_bar) <= fabs(Sv.gprime) * fmax(st->rp.phi_rtol, SPEC_VERIFY_RTOL) * Sv.phi_n;
                    else verified = ess >= Sv.ess_bar * (1.0 - 1e-13);
                }
                if (!verified) st->done = 4;
                else if (!isnan(ess) && ess < pin.thr) {
                    st->done = Sv.spec ? 4 : 3;   // (a spec stage keeps W̃ in scratch: redo it in full)
                    // a host that runs ahead unsynchronised (fixed schedules) learns of the stall without waiting for the stream
                    if (host_note) __hip_atomic_store(host_note + 1, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                else if (post_write(st, rec, pin, s_cm[0], s_cm[1]) == 0) { s_go = 1; s_c = st->c; }
            }
        }
        __syncthreads();
        if (!s_go) return;
        c = s_c;
    }
    if (from_totals) {
        if (from_totals == 3) {
        } else if (from_totals == 1) {
            // all (d+1)(d+2)/2 pair sums in one round of loads: thread (slice, pair) adds the blocks of its slice, eight
            // loads in flight, slices combined in a fixed tree (final_sum); npairs <= PT for d <= 43, else 64-wide chunks
            if (npairs <= PT) {
                double v;
                if (pr.rows) {
                    if (!prep_group_total(pr, npairs, &v)) { if (t == 0) { st->err = SMCMI_ERR_TIMEOUT; st->done = 1; } return; }
                } else v = final_sum(partials, nb_part, npairs, scratch);
                if (t < npairs) tot[t] = v;
                __syncthreads();
            } else {
                for (int p0 = 0; p0 < npairs; p0 += 64) {
                    const int m = (npairs - p0) < 64 ? (npairs - p0) : 64;
                    const double v = final_sum(partials + p0, nb_part, m, scratch, npairs);
                    if (t < m) tot[p0 + t] = v;
                    __syncthreads();
                }
            }
        } else {
            for (int p = t; p < npairs; p += PT) tot[p] = partials[p];
            __syncthreads();
        }
        const double sw = tot[0];
        for (int e = t; e < d * d; e += PT) {
            int a = e / d, b = e % d;
            if (a > b) { const int tmp = a; a = b; b = tmp; }
            const int ra = a + 1, rb = b + 1;
            const int p = ra * da - ra * (ra - 1) / 2 + (rb - ra);
            const double v = tot[p] / sw - (tot[a + 1] / sw) * (tot[b + 1] / sw);
            covl[e] = v;
            st->cov[e] = v;
        }
        for (int a = t; a < d; a += PT) {
            const double mn = st->shift[a] + tot[a + 1] / sw;
            scratch[a] = mn;                       // d <= MAXD <= TB
            st->mean[a] = mn;
            st->shift[a] = mn;
        }
    } else {
        for (int e = t; e < d * d; e += PT) covl[e] = st->cov[e];
        for (int a = t; a < d; a += PT) scratch[a] = st->mean[a];
    }
    SMCMI_STAMP(prof, 2);
    const int nb = gen_blocks ? st->rp.n_blocks : st->n_blocks;
    if (gen_blocks) {
        if (t >= 64 && t < 128) {            // wave 1 shuffles while wave 0 finishes the covariance
            const unsigned stage = (unsigned)st->stage;
            const int i0 = t - 64;               // lane i draws the Fisher-Yates partner of position i (independent Philox calls)
            if (i0 < nf) {
                bfree[i0] = i0;
                int jx = 0;
                if (i0 >= 1) {
                    double ua, ub;
                    uniform_pair(seed, 0ull, stage, rng_tag(P_BLK, (unsigned)i0, 0), ua, ub);
                    jx = (int)(ua * (double)(i0 + 1));
                    if (jx > i0) jx = i0;
                }
                fi_j[i0] = jx;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (t == 64) {
            for (int i = nf - 1; i >= 1; --i) {
                const int jx = fi_j[i];
                const int tmp = bfree[i]; bfree[i] = bfree[jx]; bfree[jx] = tmp;
            }
            const int sub = (nf + nb - 1) / nb;
            for (int b = 0; b < nb; ++b) bptr[b] = b * sub;
            bptr[nb] = nf;
        }
    } else {
        for (int i = t; i < nf; i += PT) bfree[i] = st->blocks_free[i];
        for (int b = t; b <= nb; b += PT) bptr[b] = st->block_ptr[b];
    }
    __syncthreads();
    SMCMI_STAMP(prof, 3);
    // R_fr = (R[f,f] + R[f,f]')/2, θ_bar_fr
    for (int e = t; e < nf * nf; e += PT) {
        const int a = fi[e / nf], b = fi[e % nf];
        sig_f[e] = (covl[a * d + b] + covl[b * d + a]) / 2.0;
    }
    for (int a = t; a < nf; a += PT) mu_f[a] = scratch[fi[a]];
    __syncthreads();
    for (int i = t; i < nf; i += PT) {
        const int f = bfree[i];
        st->blocks_free[i] = f;
        st->blocks_all[i] = fi[f];
        st->mu_b[i] = mu_f[f];
        st->sd_draw[i] = sqrt(c * c * sig_f[f * nf + f]);
        st->sd_dens[i] = sqrt(sig_f[f * nf + f]);
    }
    SMCMI_STAMP(prof, 4);
    int off = 0, max_db = 0;
    for (int b = 0; b < nb; ++b) {
        const int p0 = bptr[b], db = bptr[b + 1] - p0;
        if (db > max_db) max_db = db;
        for (int e = t; e < db * db; e += PT) {
            A[e] = c * c * sig_f[bfree[p0 + e / db] * nf + bfree[p0 + e % db]];
            Ls[e] = 0.0;
        }
        __syncthreads();
        // Right-looking Cholesky inside ONE wavefront (lane i owns row i, no block barriers): after column j is final the
        // trailing rows are updated A[i][k] -= L[i][j] L[k][j]; every entry therefore receives its subtractions in the same
        // k-ascending order as the textbook left-looking loop of the oracle (bitwise the same factor).
        if (t < 64 && db <= 12) {
            double r[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) r[k] = (t < db && k < db) ? A[t * db + k] : 0.0;
            const bool ok = chol_rows_in_regs<12>(r, db, t);
            if (!ok && t == 0) s_fail = 1;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (t < db && k <= t) Ls[t * db + k] = r[k];
        } else if (t < 64) {
            for (int jx = 0; jx < db; ++jx) {
                const double ajj = A[jx * db + jx];
                if (!(ajj > 0.0)) { if (t == 0) s_fail = 1; break; }
                const double ljj = sqrt(ajj);
                double lij = 0.0;
                if (t == jx) Ls[jx * db + jx] = ljj;
                if (t > jx && t < db) { lij = A[t * db + jx] / ljj; Ls[t * db + jx] = lij; }
                (void)lij;
                __builtin_amdgcn_wave_barrier();
                // trailing update of the lower triangle, one (i, k) pair per lane
                const int m = db - 1 - jx, npr = m * (m + 1) / 2;
                for (int pq = t; pq < npr; pq += 64) {
                    int r = 0, rem = pq;
                    while (rem > r) { rem -= r + 1; ++r; }
                    const int i2 = jx + 1 + r, k2 = jx + 1 + rem;
                    A[i2 * db + k2] -= Ls[i2 * db + jx] * Ls[k2 * db + jx];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (s_fail) break;
        for (int e = t; e < db * db; e += PT) st->L[off + e] = Ls[e];
        if (t < 64) {                                  // log of the diagonal in parallel, summed in index order
            const double lg = (t < db) ? log(Ls[t * db + t]) : 0.0;
            double ld = 0.0;
            for (int i = 0; i < db; ++i) ld += __shfl(lg, i, 64);
            if (t == 0) { st->logdet[b] = 2.0 * ld; st->l_off[b] = off; }
        }
        off += db * db;
        __syncthreads();
    }
    SMCMI_STAMP(prof, 5);
    for (int b = t; b <= nb; b += PT) st->block_ptr[b] = bptr[b];
    if (t == 0) {
        st->n_blocks = nb;
        st->max_db = max_db;
        if (s_fail) { st->err = SMCMI_ERR_POSDEF; st->done = 1; }   // PosDefException aborts the run (mutation.jl:81)
        if (!standalone) {
            st->mut_c = c; st->mut_alpha = st->rp.alpha; st->mut_phi = st->phi_n; st->mut_steps = st->rp.n_mh_steps;
            st->mut_stage = (unsigned)st->stage;
        }
    }
    SMCMI_STAMP(prof, 6);
}
#endif

// ------------------------------------------------------------------------------------------------ mutation
// One thread = one particle: n_mh_steps x n_blocks random-walk-mixture Metropolis-Hastings moves
// (src/mutation.jl:56-138) with the mixture draw (src/helpers.jl:87-100), proposal densities (:128-164), bounds check,
// log-prior and the device likelihood.  Per-thread vectors live in LDS ([k][thread], conflict-free) because block
// membership is a run-time index.  MODE 0 = full move; 1 = propose only (host-callback split); 2 = accept only.
struct MutArgs {
    unsigned long long seed;
    long long gid0;            // global id of local particle 0
    double *proposals;         // MODE 1/2: n x d proposals (column-major), logprior', q0-q1
    double *prop_logprior, *prop_qdiff;
    long long prop_chunk;      // MODE 1/2, > 0: `proposals` is chunk-major - particles [c m, (c + 1) m) form one contiguous column-major block of
                               // len x d (len = m, the last chunk's is shorter), so a chunk crosses PCIe as ONE linear copy and reaches the host
                               // callback as the m x d matrix its contract names, without a pack (callback.hpp host_mutation); column d of a
                               // chunk's block holds the proposals' log-priors (prop_logprior is not used), so a chunk is ONE copy; 0: n x d
    const double *lik_new, *lik_old_new;   // MODE 2
    int *acc_count;            // MODE 1/2: accepted block lengths so far
    int block, step, last;     // MODE 1/2
    long long *prof;           // development only: per-phase shader-clock stamps of block 0 / middle block, wave 0
    int debug;                 // development only (tools/kbench.py): bit0 skip normals, bit1 skip prior/likelihood, bit2 skip matvec
    double *esum;              // in-run MODE 0: per-block energy power sums for the next stage's ϕ predictor ([blocks][ES]) or null
    const double *zbuf;        // in-run register kernel: random numbers drawn ahead by k_prepare_mutation (RngAhead layout) or null
    int z_ahead;               //   ... for the proposals t < z_ahead (the later ones are drawn in the kernel)
    int normalize;             // in-run register kernel after k_correct_moments: the weight column still holds W̃ - apply
                               // normalize_weights! (src/particle.jl:362-366) here and write the W history column
    double *hist_W;
    long long hist_ld;
    const double *wt;          // normalize: read W̃ from here instead of the weight column (spec stages)
    double *emax;              // in-run MODE 0: per-block largest loglh - old_loglh of the mutated cloud (next stage's energy shift) or null
    const double *mix;         // register kernel, α < 1: the blocks' dense mixture matrices from k_mix_prepare ([n_blocks][MixDense::DOUBLES])
    const int *mixpos;         //                         and parameter positions ([n_blocks][D])
    int stage_consts;          // k_mutate<0, 1>: the launch reserved sizeof(MutStage) + 64 bytes behind the per-thread vectors (n_para <= 13)
};

// LS = 4 (lgss_kalman family only, MODE 0): FOUR lanes per particle - a 256-thread block carries 64 particles, the four lanes of a
// quad run the per-particle part (draw, densities, prior, decision) redundantly on the same values (same counters -> same random
// numbers; nothing to exchange) and share the Kalman filter (model.hpp kalman_lgss_quad).  Per wavefront: θ, θ' [d][16], then
// {draw, solve scratch [d][16]} overlaid with the filter's 16 transposition slots (dead while the filter runs: the accepted
// proposal is copied from θ').  Lane 0 of a quad stores the particle and feeds the block sums.
// what the proposal of the lane-split kernel reads per block and per parameter, staged in LDS once per launch (in DevState / ModelDev
// they are global loads inside rolled loops: a dependent ~0.3-1 µs round trip per iteration with one wavefront per SIMD)
struct MutStage {
    double L[13 * 13], mu_b[13], sd_draw[13], sd_dens[13], logdet[13], lo[13], hi[13], prior_a[13], prior_b[13], prior_k[13];
    int block_ptr[14], blocks_all[13], l_off[13], fixed[13], prior_family[13];
};
constexpr int mutate_wave_bytes_ls4(int d) {
    return (2 * d * 16 * 8 + ((2 * d * 16 * 8 > 16 * KALMAN4_SLOT_BYTES) ? 2 * d * 16 * 8 : 16 * KALMAN4_SLOT_BYTES) + 15) / 16 * 16;
}
// The Metropolis-Hastings moves of ONE particle (src/mutation.jl:56-138; mixture draw helpers.jl:87-100, proposal densities :128-164,
// bounds, log-prior, device likelihood) for any n_para, with the per-particle vectors in LDS columns th / tn / y / v ([k][T], column tid):
// the body of the generic mutation kernel k_mutate below (engine 1, host-callback split) and of engine 2's kernel for n_para > 10
// (stage2.hpp k2w_mutate).  The stage's proposal arrives as plain arrays (a_*: packed block factors, block means, marginal scales,
// log-determinants, block structure) - DevState, or an LDS staging of it.  MODE / LS as k_mutate.  `src`: cloud buffer to read the
// particle from.  Every lane of the block calls (the lane-split Kalman filter needs whole wavefronts); results for `live` lanes.
template <int MODE, int LS>
__device__ __forceinline__ void mutate_generic(const CloudPtrs &cl, const ModelDev *md, const MutArgs &ma, double *th, double *tn, double *y, double *v, const int T,
                                               const int tid, const int quad_lane, const long long i, const bool live, const int src, const unsigned long long pid,
                                               const unsigned stage, const double c_alpha, const double phi_n, const int nb, const int n_steps, const int d,
                                               const double *a_L, const double *a_mu, const double *a_sdd, const double *a_sdn, const double *a_logdet,
                                               const int *a_bptr, const int *a_ball, const int *a_loff, const ModelView &mv, double &like, double &lprior,
                                               double &like_prev, double &accept) {
    if (live) {
        like = col(cl, src, d)[i]; lprior = col(cl, src, d + 1)[i]; like_prev = col(cl, src, d + 2)[i];
        load_columns(cl.buf[src], cl.n, i, d, th + tid, T);
        for (int k = 0; k < d; ++k) tn[k * T + tid] = th[k * T + tid];
    }
    // lgss_kalman on both vintages with one thread per particle: the filter whose structure values travel through DPP operands
    // (model.hpp kalman_lgss_wave) - called by every lane, like the lane-split one.
    const bool kalman_wave = LS == 1 && MODE == 0 && d == 13 && md->lik[0].family == SMCMI_LIK_LGSS_KALMAN &&
                             (md->lik[1].family == SMCMI_LIK_NONE || md->lik[1].family == SMCMI_LIK_LGSS_KALMAN);
    auto TN = [&](int k) { return tn[k * T + tid]; };
    // proposal k of this particle in the MODE 1/2 buffer: proposals[k * p_ld + p_off]
    long long p_ld = cl.n, p_off = i;
    if (MODE != 0 && ma.prop_chunk > 0 && live) {
        const long long c0 = (i / ma.prop_chunk) * ma.prop_chunk;
        p_ld = (cl.n - c0 < ma.prop_chunk) ? cl.n - c0 : ma.prop_chunk;
        p_off = c0 * (d + 1) + (i - c0);                // (a chunk's block: d proposal columns + the log-prior column)
    }
    // (the loops are uniform and `live` is tested inside them: the lane-split filter needs every lane of the wavefront)
    {
        const int s_beg = (MODE == 0) ? 0 : ma.step, s_end = (MODE == 0) ? n_steps : ma.step + 1;
        for (int step = s_beg; step < s_end; ++step) {
            const int b_beg = (MODE == 0) ? 0 : ma.block, b_end = (MODE == 0) ? nb : ma.block + 1;
            for (int b = b_beg; b < b_end; ++b) {
                const int p0 = a_bptr[b], db = a_bptr[b + 1] - p0;
                const double *L = a_L + a_loff[b];
                const unsigned t = (unsigned)(step * nb + b);
                // MH uniform for this decision: drawn "before" the proposal (quirk Q3)
                double step_prob = 0.0, u_dummy;
                bool inb = false;
                if (!live) {}
                else if (t == 0) uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, 0xFFFFFu, 0), step_prob, u_dummy);
                else uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t - 1, 0), u_dummy, step_prob);
                double q0 = 0.0, q1 = 0.0, prior_new = SMCMI_NEG_INF, like_new = SMCMI_NEG_INF, like_old_data = SMCMI_NEG_INF;
                if (MODE != 2 && live) {
                    // ---- mvnormal_mixture_draw
                    double uc, unext;
                    uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t, 0), uc, unext);
                    for (int e = 0; e < db; e += 2) {
                        double z0, z1;
                        normal_pair(ma.seed, pid, stage, rng_tag(P_MUT, t, 1 + e / 2), z0, z1);
                        y[e * T + tid] = z0;
                        if (e + 1 < db) y[(e + 1) * T + tid] = z1;
                    }
                    const int comp = (uc < c_alpha) ? 0 : (uc < c_alpha + (1.0 - c_alpha) / 2.0 ? 1 : 2);
                    if (comp == 1) {
                        for (int e = 0; e < db; ++e)
                            y[e * T + tid] = th[a_ball[p0 + e] * T + tid] + a_sdd[p0 + e] * y[e * T + tid];
                    } else {
                        for (int e = db - 1; e >= 0; --e) {   // in place: row e only needs z_0..z_e
                            double s = 0.0;
#pragma unroll 4
                            for (int k = 0; k <= e; ++k) s += L[e * db + k] * y[k * T + tid];
                            const double center = (comp == 0) ? th[a_ball[p0 + e] * T + tid] : a_mu[p0 + e];
                            y[e * T + tid] = center + s;
                        }
                    }
                    // ---- compute_proposal_densities
                    const double cst = (double)db * LOG2PI + a_logdet[b];
                    double quad = 0.0, quad_s = 0.0, quad_d = 0.0, ind_pdf = 1.0;
                    if (LS == 4) {
                        // the three forward substitutions L⁻¹(θ_b - ϑ_b), L⁻¹(θ_b - θ̄_b), L⁻¹(ϑ_b - θ̄_b) share one sweep over L (each
                        // sum in the literal order: same bits): a third of the loads, three independent chains
                        double *v2 = v + (long long)d * T, *v3 = v2 + (long long)d * T;
                        for (int e = 0; e < db; ++e) {
                            const double te = th[a_ball[p0 + e] * T + tid], ye = y[e * T + tid], me = a_mu[p0 + e];
                            double s1 = te - ye, s2 = te - me, s3 = ye - me;
#pragma unroll 4
                            for (int k = 0; k < e; ++k) {
                                const double l = L[e * db + k];
                                s1 -= l * v[k * T + tid]; s2 -= l * v2[k * T + tid]; s3 -= l * v3[k * T + tid];
                            }
                            const double dg = L[e * db + e];
                            const double e1 = s1 / dg, e2 = s2 / dg, e3 = s3 / dg;
                            v[e * T + tid] = e1; v2[e * T + tid] = e2; v3[e * T + tid] = e3;
                            quad += e1 * e1; quad_s += e2 * e2; quad_d += e3 * e3;
                            const double sii = a_sdn[p0 + e];
                            const double z = (te - ye) / sii;
                            ind_pdf = ind_pdf / (sii * sqrt(2.0 * M_PI)) * exp(-0.5 * z * z);
                        }
                    } else {
                    for (int e = 0; e < db; ++e) {            // L⁻¹(θ_b - ϑ_b): forward == reverse density
                        double s = th[a_ball[p0 + e] * T + tid] - y[e * T + tid];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad += ve * ve;
                    }
                    for (int e = 0; e < db; ++e) {
                        const double sii = a_sdn[p0 + e];
                        const double z = (th[a_ball[p0 + e] * T + tid] - y[e * T + tid]) / sii;
                        ind_pdf = ind_pdf / (sii * sqrt(2.0 * M_PI)) * exp(-0.5 * z * z);
                    }
                    for (int e = 0; e < db; ++e) {            // log N(θ_b; θ̄_b, c²Σ)
                        double s = th[a_ball[p0 + e] * T + tid] - a_mu[p0 + e];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad_s += ve * ve;
                    }
                    for (int e = 0; e < db; ++e) {            // log N(ϑ_b; θ̄_b, c²Σ)
                        double s = y[e * T + tid] - a_mu[p0 + e];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad_d += ve * ve;
                    }
                    }
                    const double lp_sym = -(cst + quad) / 2.0;
                    q0 = c_alpha * exp(lp_sym);
                    q1 = q0;
                    q0 += (1.0 - c_alpha) / 2.0 * ind_pdf;
                    q1 += (1.0 - c_alpha) / 2.0 * ind_pdf;
                    q0 += (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_s) / 2.0);
                    q1 += (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_d) / 2.0);
                    q0 = log(q0);
                    q1 = log(q1);
                    if (q0 == __builtin_huge_val() && q1 == __builtin_huge_val()) q0 = 0.0;
                    // ---- para_new
                    for (int e = 0; e < db; ++e) tn[a_ball[p0 + e] * T + tid] = y[e * T + tid];
                    inb = in_bounds(mv, TN);
                    if (inb) prior_new = logprior(mv, TN);
                    if (MODE == 1) {
                        for (int k = 0; k < d; ++k) ma.proposals[(long long)k * p_ld + p_off] = TN(k);
                        if (ma.prop_chunk > 0) ma.proposals[(long long)d * p_ld + p_off] = prior_new;
                        else ma.prop_logprior[i] = prior_new;
                        ma.prop_qdiff[i] = q0 - q1;
                    }
                }
                if (MODE == 1) continue;
                if (MODE == 0 && (LS == 4 || kalman_wave)) {
                    // every lane calls the filter (its structure values travel through DPP operands); lanes without a proposal inside
                    // the bounds run it on the particle's current θ, lanes without a particle on ρ = 0, σ = 0.5, and drop the result
                    double thv[13];
                    for (int k = 0; k < 13; ++k) thv[k] = (live && inb) ? TN(k) : (live ? th[k * T + tid] : (k < 8 ? 0.0 : (k < 12 ? 0.5 : 0.0)));
                    const lds_bytes slot = (lds_bytes)(y) + tid * KALMAN4_SLOT_BYTES;
                    // (LS = 4: the draw in y is dead from here on - θ' holds it at the block's positions.  The quad filter is inlined - one
                    // register allocation with the kernel, nothing saved around a call - hence ONE call site: the second trip is the old
                    // vintage when it is not a prefix of the data)
                    const int n_pass = (md->lik_prefix > 0 || md->lik[1].family == SMCMI_LIK_NONE) ? 1 : 2;      // (uniform)
                    double r_new = 0.0, r_old = 0.0;
#pragma nounroll
                    for (int pass = 0; pass < n_pass; ++pass) {
                        const LikDev &lk = md->lik[pass];
                        KalmanLL r;
                        if constexpr (LS == 4) r = kalman_lgss_quad(thv, lk.data, lk.cols, pass == 0 ? md->lik_prefix : 0, lk.aux, slot, quad_lane);
                        else {
                            KalmanTheta kt;
                            for (int k = 0; k < 13; ++k) kt.v[k] = thv[k];
                            r = kalman_lgss_wave(kt, lk.data, lk.cols, pass == 0 ? md->lik_prefix : 0, lk.aux);
                        }
                        if (pass == 0) { r_new = r.ll; if (md->lik_prefix > 0) r_old = r.ll_mid; }
                        else r_old = r.ll;
                    }
                    if (live && inb) {
                        like_new = r_new; like_old_data = r_old;
                        if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                    }
                } else if (MODE != 2) {
                    if (live && inb) {
                        if (md->lik_prefix > 0) {
                            // old data = a prefix of the data, same state-space structure: both log-likelihoods from one filter pass
                            double thv[13];
                            for (int k = 0; k < 13; ++k) thv[k] = TN(k);
                            const KalmanLL r = kalman_lgss2(thv, md->lik[0].data, md->lik[0].cols, md->lik_prefix, md->lik[0].aux, md->lik[0].par[0]);
                            like_new = r.ll; like_old_data = r.ll_mid;
                        } else {
                            like_new = loglik(md->lik[0], d, TN);
                            like_old_data = (md->lik[1].family == SMCMI_LIK_NONE) ? 0.0 : loglik(md->lik[1], d, TN);
                        }
                        if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                    }
                } else if (live) {
                    for (int k = 0; k < d; ++k) tn[k * T + tid] = ma.proposals[(long long)k * p_ld + p_off];
                    prior_new = ma.prop_chunk > 0 ? ma.proposals[(long long)d * p_ld + p_off] : ma.prop_logprior[i];
                    q0 = ma.prop_qdiff[i];
                    q1 = 0.0;
                    if (prior_new == SMCMI_NEG_INF) {            // out of bounds: ParamBoundsError => everything -Inf
                        like_new = like_old_data = SMCMI_NEG_INF;
                    } else {
                        like_new = ma.lik_new[i];
                        like_old_data = ma.lik_old_new ? ma.lik_old_new[i] : 0.0;
                        if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                    }
                }
                const double eta = exp(phi_n * (like_new - like) + (1.0 - phi_n) * (like_old_data - like_prev) +
                                       (prior_new - lprior) + (q0 - q1));
                if (!live) {}
                else if (step_prob < eta) {
                    if (MODE == 2) { for (int k = 0; k < d; ++k) th[k * T + tid] = tn[k * T + tid]; }
                    else if (LS == 4) { for (int e = 0; e < db; ++e) { const int k = a_ball[p0 + e]; th[k * T + tid] = tn[k * T + tid]; } }
                    else for (int e = 0; e < db; ++e) th[a_ball[p0 + e] * T + tid] = y[e * T + tid];
                    like = like_new; lprior = prior_new; like_prev = like_old_data;
                    accept += (double)db;
                } else if (MODE != 2) {
                    for (int e = 0; e < db; ++e) tn[a_ball[p0 + e] * T + tid] = th[a_ball[p0 + e] * T + tid];
                }
            }
        }
    }
}

template <int MODE, int LS = 1>
__global__ void __launch_bounds__(256, 1) k_mutate(CloudPtrs cl, const DevState *st, const ModelDev *md, MutArgs ma, double *acc_partials,
                         int standalone) {
    static_assert(LS == 1 || (LS == 4 && MODE == 0), "lane-split mutation: MODE 0 only");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (!standalone && st->done) return;
    const int T = (LS == 4) ? 16 : blockDim.x;                    // stride of the per-particle LDS vectors
    const int tid = (LS == 4) ? ((threadIdx.x & 63) >> 2) : threadIdx.x;   // column in them
    const int quad_lane = threadIdx.x & 3;
    const bool lead = (LS == 1) || quad_lane == 0;
    const int d = md->d, nf = md->n_free;
    double *wave_base = (LS == 4) ? sm + (long long)(threadIdx.x >> 6) * (mutate_wave_bytes_ls4(13) / 8) : sm;
    double *th = wave_base;                // current θ           [d][T]
    double *tn = th + (long long)d * T;    // proposed θ          [d][T]
    double *y = tn + (long long)d * T;     // z / draw            [d][T]
    double *v = y + (long long)d * T;      // triangular-solve scratch [d][T]
    double *red = (LS == 4) ? sm + (long long)(blockDim.x >> 6) * (mutate_wave_bytes_ls4(13) / 8) : v + (long long)d * T;    // [blockDim.x/64]
    const long long i = (LS == 4) ? (long long)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2) : (long long)blockIdx.x * T + tid;
    const bool live = i < cl.n;
    constexpr int src = 0;
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    const unsigned stage = st->mut_stage;
    const double c_alpha = st->mut_alpha, phi_n = st->mut_phi;
    const int nb = st->n_blocks, n_steps = st->mut_steps;
    double like = 0.0, lprior = 0.0, like_prev = 0.0, accept = 0.0;
    // LS = 1: the same staging when the host reserved the room (MODE 0, n_para <= 13: ma.stage_consts)
    const bool staged = (LS == 4) || (MODE == 0 && ma.stage_consts && d <= 13);
    MutStage *S = staged ? (MutStage *)(red + 8) : nullptr;
    if (staged) {
        for (int e = threadIdx.x; e < d * d; e += blockDim.x) S->L[e] = st->L[e];
        if ((int)threadIdx.x < d) {
            const int e = threadIdx.x;
            S->mu_b[e] = st->mu_b[e]; S->sd_draw[e] = st->sd_draw[e]; S->sd_dens[e] = st->sd_dens[e]; S->logdet[e] = st->logdet[e];
            S->lo[e] = md->lo[e]; S->hi[e] = md->hi[e]; S->prior_a[e] = md->prior_a[e]; S->prior_b[e] = md->prior_b[e]; S->prior_k[e] = md->prior_k[e];
            S->blocks_all[e] = st->blocks_all[e]; S->l_off[e] = st->l_off[e]; S->fixed[e] = md->fixed[e]; S->prior_family[e] = md->prior_family[e];
        }
        if ((int)threadIdx.x <= d) S->block_ptr[threadIdx.x] = st->block_ptr[threadIdx.x];
        __syncthreads();
    }
    // (otherwise the same arrays are read where they are)
    const double *a_L = staged ? S->L : st->L, *a_mu = staged ? S->mu_b : st->mu_b, *a_sdd = staged ? S->sd_draw : st->sd_draw;
    const double *a_sdn = staged ? S->sd_dens : st->sd_dens, *a_logdet = staged ? S->logdet : st->logdet;
    const int *a_bptr = staged ? S->block_ptr : st->block_ptr, *a_ball = staged ? S->blocks_all : st->blocks_all, *a_loff = staged ? S->l_off : st->l_off;
    const ModelView mv{d, staged ? S->fixed : md->fixed, staged ? S->prior_family : md->prior_family, staged ? S->lo : md->lo, staged ? S->hi : md->hi,
                       staged ? S->prior_a : md->prior_a, staged ? S->prior_b : md->prior_b, staged ? S->prior_k : md->prior_k};
    mutate_generic<MODE, LS>(cl, md, ma, th, tn, y, v, T, tid, quad_lane, i, live, src, pid, stage, c_alpha, phi_n, nb, n_steps, d, a_L, a_mu, a_sdd, a_sdn, a_logdet,
                             a_bptr, a_ball, a_loff, mv, like, lprior, like_prev, accept);
    if (MODE == 1) return;
    double acc_val = 0.0;
    if (live && lead) {
        for (int k = 0; k < d; ++k) col(cl, src, k)[i] = th[k * T + tid];
        col(cl, src, d)[i] = like;
        col(cl, src, d + 1)[i] = lprior;
        col(cl, src, d + 2)[i] = like_prev;
        if (MODE == 0) {
            acc_val = accept / (double)nf;                      // quirk Q2: normalised by n_free only
            col(cl, src, d + 3)[i] = acc_val;
        } else {
            int cnt = ((ma.step == 0 && ma.block == 0) ? 0 : ma.acc_count[i]) + (int)accept;
            ma.acc_count[i] = cnt;
            acc_val = (double)cnt / (double)nf;
            if (ma.last) col(cl, src, d + 3)[i] = acc_val;
        }
    }
    // Σ accept over the block (update_acceptance_rate!, src/particle.jl:466-468), fixed order
    const int btid = threadIdx.x, nwv = blockDim.x >> 6;
    const bool counted = live && lead;               // lanes 1..3 of a quad (LS = 4) carry copies: they add nothing to the block sums
    double a1[1] = {acc_val};
    Butterfly<0, 32>::run(a1, btid & 63);
    if ((btid & 63) == 0) red[btid >> 6] = a1[0];
    // (MODE 2: the accept launch of a stage's LAST proposal leaves them as well - the host-closure path then has the predictor's rings
    // around its certificate pass and the shifted weights, like every other path)
    const bool leaves_sums = MODE == 0 || (MODE == 2 && ma.last);
    if (leaves_sums && ma.emax) {                    // largest energy of the mutated cloud (energy shift of the next stage)
        __shared__ double emx[4];
        const double wl = (counted && !st->do_resample) ? col(cl, src, d + 4)[i] : 1.0;
        const double em = block_max(energy_or_ninf(like, like_prev, wl, counted), emx, nwv);
        if (btid == 0) ma.emax[blockIdx.x] = em;
    }
    if (leaves_sums && ma.esum) {                    // energy power sums of the mutated cloud (ϕ predictor of the next stage)
        double es[ES];
        energy_terms(es, counted ? col(cl, src, d + 4)[i] : 0.0, like, like_prev, st->e_center, counted, st->do_resample != 0);
        es[EACC] = acc_val;
        const double tot = block_reduce_es(es, sm, nwv);             // θ staging area is dead by now (nwv ES <= d T)
        if (btid < ES) ma.esum[(long long)blockIdx.x * ES + btid] = tot;
    }
    __syncthreads();
    if (btid == 0) {
        double s = 0.0;
        for (int w = 0; w < nwv; ++w) s += red[w];
        acc_partials[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------ mixture proposal, dense form
// The α < 1 proposal (src/helpers.jl:87-100 draw, :128-164 densities) of the register-resident kernels, arranged so that nothing
// is indexed by block position at run time and nothing divides per particle.  Per random block b with factor L (c²Σ_b = L Lᵀ,
// block order e) and W = L⁻¹, the block's constants are expanded, in NATURAL parameter order k, to
//   LsT[e][k] = L[pos(k)][e]         draw of the full-covariance components:  Σ_e LsT[e][k] z_e      (zero rows/columns outside b)
//   WsT[k][e] = W[e][pos(k)]         L⁻¹ r for r in parameter order:          v_e = Σ_k WsT[k][e] r_k
//   mu_p, sdd_p, isd_p = θ̄_b, sd_draw, 1/sd_dens in parameter order (0 outside b), pos(k), ind_c = Π_e 1/(sd_dens_e √(2π)),
//   cst = d_b log 2π + log|c²Σ_b|.
// A proposal is then two dense D x D sweeps with compile-time register indices (the draw; the two solves L⁻¹(θ-ϑ), L⁻¹(θ-θ̄) in
// two halves of the rows, L⁻¹(ϑ-θ̄) being their difference), the diagonal component's z_pos(k) through a private LDS column,
// one exponential for the diagonal density and one logarithm of the forward/reverse ratio - against three forward substitutions
// with 3 d divisions, d exponentials + d square roots + 2 d divisions in the diagonal density, two d x d select chains for the
// gather / scatter of θ_b and two logarithms in the literal restatement (k_mutate MODE 0 keeps that one).  Deviation: the sums run
// in a different order (differences ~1e-16 cond(L) relative in the log density ratio; a literal/dense run pair flips an MH
// decision with probability ~1e-9 per proposal).
template <int D, class PD, class PI>
struct MixDenseT {
    PD LsT, WsT;                  // [D*D] each
    PD mu_p, sdd_p, isd_p;        // [D]
    PD sc;                        // [0] = ind_c, [1] = cst
    PI pos;                       // [D] position of parameter k in the block, or -1
    static constexpr int DOUBLES = 2 * D * D + 3 * D + 2;
    __device__ MixDenseT(PD buf, PI ibuf) : LsT(buf), WsT(buf + D * D), mu_p(buf + 2 * D * D), sdd_p(mu_p + D), isd_p(sdd_p + D), sc(isd_p + D), pos(ibuf) {}
};
template <int D> using MixDense = MixDenseT<D, double *, int *>;          // LDS / global memory, writable (mix_expand)
// the same block read through the constant address space: wave-uniform scalar loads, the matrix entries arrive as SGPR operands
// of the FMAs (no LDS traffic, no vector registers) - for matrices a previous launch wrote (k_mix_prepare)
using mix_cdp = const double __attribute__((address_space(4))) *;
using mix_cip = const int __attribute__((address_space(4))) *;
template <int D> using MixDenseC = MixDenseT<D, mix_cdp, mix_cip>;
// W_b = L_b⁻¹ for every block of the stage: wave w inverts blocks w, w + NW, ...; lane j builds column j by forward substitution
// (L read from LDS at uniform addresses).  Lraw / Wraw: packed d_b x d_b row-major at loff[b].  Ends with a barrier.
template <int D, int T>
__device__ inline void mix_invert_factors(const double *Lraw, double *Wraw, const int *loff, const int *bptr, int nb, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int b = wave; b < nb; b += T / 64) {
        const int db = bptr[b + 1] - bptr[b];
        const double *L = Lraw + loff[b];
        double w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            w[i] = 0.0;
            if (i < db) {
                double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) s -= L[i * db + k] * w[k];
                w[i] = (i >= lane) ? s / L[i * db + i] : 0.0;
            }
        }
        if (lane < db) {
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (i < db) Wraw[loff[b] + i * db + lane] = w[i];
        }
    }
    __syncthreads();
}
// expansion of block b (all T threads; the caller has synchronised the previous readers; ends with a barrier)
template <int D, int T>
__device__ inline void mix_expand(const MixDense<D> &M, const double *Lb, const double *Wb, const int *ball_b, const double *mub_b, const double *sdd_b,
                                  const double *sdn_b, int db, double logdet, int tid) {
    for (int k = tid; k < D; k += T) M.pos[k] = -1;
    __syncthreads();
    for (int e = tid; e < db; e += T) M.pos[ball_b[e]] = e;
    __syncthreads();
    for (int idx = tid; idx < D * D; idx += T) {
        const int a = idx / D, k = idx % D;                  // LsT: a = block position e, k = parameter
        const int pk = M.pos[k];
        M.LsT[idx] = (pk >= 0 && a <= pk) ? Lb[pk * db + a] : 0.0;
        const int pa = M.pos[a];                             // WsT: a = parameter, k = block position e
        M.WsT[idx] = (pa >= 0 && pa <= k && k < db) ? Wb[k * db + pa] : 0.0;
    }
    for (int k = tid; k < D; k += T) {
        const int pk = M.pos[k];
        M.mu_p[k] = pk >= 0 ? mub_b[pk] : 0.0;
        M.sdd_p[k] = pk >= 0 ? sdd_b[pk] : 0.0;
        M.isd_p[k] = pk >= 0 ? 1.0 / sdn_b[pk] : 0.0;
    }
    if (tid == 0) {
        double ic = 1.0;
        for (int e = 0; e < db; ++e) ic = ic / (sdn_b[e] * sqrt(2.0 * M_PI));     // the reference's running product (helpers.jl:150-154)
        M.sc[0] = ic;
        M.sc[1] = (double)db * LOG2PI + logdet;
    }
    __syncthreads();
}
// rows [E0, E1) of the two solves and their contribution to the three quadratic forms
template <int D, int E0, int E1, class MX>
__device__ inline void mix_solve_rows(const MX &M, const double (&x)[D], const double (&xn)[D], double &quad, double &quad_s, double &quad_d) {
SMCMI_FP_CONTRACT
    constexpr int NE = E1 - E0;
    if constexpr (NE > 0) {
        double v1[NE], v2[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { v1[e] = 0.0; v2[e] = 0.0; }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double xk = x[k];
            asm volatile("" : "+v"(xk));                   // (a fresh value: otherwise x - xn and x - θ̄ of all k stay live from the first use on)
            const double r1 = xk - xn[k], r2 = xk - M.mu_p[k];
#pragma unroll
            for (int e = 0; e < NE; ++e) { const double wv = M.WsT[k * D + E0 + e]; v1[e] += wv * r1; v2[e] += wv * r2; }
#pragma unroll
            for (int e = 0; e < NE; ++e) asm volatile("" : "+v"(v1[e]), "+v"(v2[e]));      // one matrix row live at a time
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double v3 = v2[e] - v1[e];
            quad += v1[e] * v1[e]; quad_s += v2[e] * v2[e]; quad_d += v3 * v3;
        }
    }
}
// compute_proposal_densities (src/helpers.jl:128-164) in the dense form: the two mixture densities (not yet logarithms) of the move
// x (θ, para_subset) <-> xn (ϑ, para_draw); ssq = Σ_k ((x_k - xn_k) / sd_dens_k)² over the block, the diagonal component's exponent
// (its standard deviations are the UNSCALED Σ_ii: quirk Q1).  q0 carries N(θ_b; θ̄_b, c²Σ_b), q1 carries N(ϑ_b; θ̄_b, c²Σ_b).
template <int D, class MX>
__device__ inline void mix_densities(const MX &M, const double (&x)[D], const double (&xn)[D], double ssq, double c_alpha, double &q0, double &q1) {
SMCMI_FP_CONTRACT
    double quad = 0.0, quad_s = 0.0, quad_d = 0.0;
    constexpr int HALF = (D + 1) / 2;
    mix_solve_rows<D, 0, HALF, MX>(M, x, xn, quad, quad_s, quad_d);
    mix_solve_rows<D, HALF, D, MX>(M, x, xn, quad, quad_s, quad_d);
    const double cst = M.sc[1];
    const double common = c_alpha * exp(-(cst + quad) / 2.0) + (1.0 - c_alpha) / 2.0 * (M.sc[0] * exp(-0.5 * ssq));
    q0 = common + (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_s) / 2.0);
    q1 = common + (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_d) / 2.0);
}
// One proposal: x (current θ, parameter order) -> xn, and log q(θ|ϑ) - log q(ϑ|θ) as the MH ratio uses it (NaN when both
// densities vanish, like the reference's log 0 - log 0).  z: the block's standard normals (block order, zero beyond d_b);
// zt: this thread's private LDS column (stride T), D entries.
template <int D, int T, class MX>
__device__ inline double mix_propose(const MX &M, const double (&x)[D], const double (&z)[D], double uc, double c_alpha, bool force_diag,
                                     double *zt, double (&xn)[D]) {
SMCMI_FP_CONTRACT
    const int comp = (uc < c_alpha) ? 0 : (uc < c_alpha + (1.0 - c_alpha) / 2.0 ? 1 : 2);
    const bool diag_draw = comp == 1 || force_diag;
#pragma unroll
    for (int e = 0; e < D; ++e) zt[e * T] = z[e];
    double u[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) {
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] += M.LsT[e * D + k] * z[e];
#pragma unroll
        for (int k = 0; k < D; ++k) asm volatile("" : "+v"(u[k]));      // one matrix row live at a time
    }
    double ssq = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int pk = M.pos[k];
        const double zp = zt[(pk >= 0 ? pk : 0) * T];                       // z_pos(k): the diagonal component's draw (sdd_p = 0 outside b)
        const double centre = (comp == 2 && pk >= 0) ? M.mu_p[k] : x[k];
        xn[k] = diag_draw ? x[k] + M.sdd_p[k] * zp : centre + u[k];
        const double zz = (x[k] - xn[k]) * M.isd_p[k];
        ssq += zz * zz;
    }
    double q0, q1;
    mix_densities<D, MX>(M, x, xn, ssq, c_alpha, q0, q1);
    return log(q0 / q1);
}

// The dense mixture matrices of every block of the stage, once per stage by one block (the register kernel's blocks hold 256
// particles each: inverting and expanding per block costs more than the mutation itself once the cloud is large).
template <int D>
__global__ void __launch_bounds__(256) k_mix_prepare(const DevState *st, int nb, int nf, double *mix, int *mixpos) {
    constexpr int T = 256;
    __shared__ double Lraw[D * D], Wraw[D * D], mub[D], sdd[D], sdn[D], logdet[D];
    __shared__ int ball[D], bptr[D + 1], loff[D];
    const int tid = threadIdx.x;
    for (int e = tid; e < nf * nf; e += T) Lraw[e] = st->L[e];
    for (int e = tid; e < nf; e += T) { mub[e] = st->mu_b[e]; sdd[e] = st->sd_draw[e]; sdn[e] = st->sd_dens[e]; ball[e] = st->blocks_all[e]; }
    for (int b = tid; b < nb; b += T) { loff[b] = st->l_off[b]; logdet[b] = st->logdet[b]; }
    for (int b = tid; b <= nb; b += T) bptr[b] = st->block_ptr[b];
    __syncthreads();
    mix_invert_factors<D, T>(Lraw, Wraw, loff, bptr, nb, tid);
    for (int b = 0; b < nb; ++b) {
        const MixDense<D> G(mix + (long long)b * MixDense<D>::DOUBLES, mixpos + b * D);
        const int p0 = bptr[b], db = bptr[b + 1] - p0;
        mix_expand<D, T>(G, Lraw + loff[b], Wraw + loff[b], ball + p0, mub + p0, sdd + p0, sdn + p0, db, logdet[b], tid);
    }
}

// Development / parity aid (smcmi_debug_proposal_densities): compute_proposal_densities (src/helpers.jl:128-164) of ONE move through the
// dense form the register kernels use - factorisation of c²Σ, L⁻¹, mix_expand, mix_densities - for a block that is the whole vector
// (positions = identity).  out[0] = q0, out[1] = q1 (logarithms; the reference's `q0 == Inf && q1 == Inf -> q0 = 0` included), out[2] = 1
// if c²Σ is not positive definite.
template <int D>
__global__ void __launch_bounds__(256) k_debug_mix_densities(const double *para_draw, const double *para_subset, const double *mu, const double *Sigma, int d,
                                                             double c, double alpha, double *out) {
    constexpr int T = 256;
    __shared__ double Lraw[D * D], Wraw[D * D], mub[D], sdd[D], sdn[D], mixbuf[MixDense<D>::DOUBLES], logdet_s;
    __shared__ int ball[D], bptr[2], loff[1], mixpos[D], fail;
    const int tid = threadIdx.x;
    for (int e = tid; e < d * d; e += T) Lraw[e] = 0.0;
    for (int e = tid; e < d; e += T) { mub[e] = mu[e]; ball[e] = e; sdd[e] = sqrt(c * c * Sigma[e * d + e]); sdn[e] = sqrt(Sigma[e * d + e]); }
    if (tid == 0) { bptr[0] = 0; bptr[1] = d; loff[0] = 0; fail = 0; }
    __syncthreads();
    if (tid == 0) {                                          // c²Σ = L Lᵀ, row by row (MvNormal(θ̄_b, c²Σ_b), src/mutation.jl:81)
        double ld = 0.0;
        for (int i = 0; i < d && !fail; ++i)
            for (int j = 0; j <= i; ++j) {
                double sacc = c * c * Sigma[i * d + j];
                for (int k = 0; k < j; ++k) sacc -= Lraw[i * d + k] * Lraw[j * d + k];
                if (i == j) {
                    if (!(sacc > 0.0)) { fail = 1; break; }
                    Lraw[i * d + i] = sqrt(sacc);
                    ld += 2.0 * log(Lraw[i * d + i]);
                } else Lraw[i * d + j] = sacc / Lraw[j * d + j];
            }
        logdet_s = ld;
    }
    __syncthreads();
    if (fail) { if (tid == 0) { out[0] = out[1] = __builtin_nan(""); out[2] = 1.0; } return; }
    mix_invert_factors<D, T>(Lraw, Wraw, loff, bptr, 1, tid);
    const MixDense<D> M(mixbuf, mixpos);
    mix_expand<D, T>(M, Lraw, Wraw, ball, mub, sdd, sdn, d, logdet_s, tid);
    if (tid == 0) {
        double x[D], xn[D], ssq = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            x[k] = k < d ? para_subset[k] : 0.0;
            xn[k] = k < d ? para_draw[k] : 0.0;
            const double zz = (x[k] - xn[k]) * M.isd_p[k];
            ssq += zz * zz;
        }
        double q0, q1;
        mix_densities<D, MixDense<D>>(M, x, xn, ssq, alpha, q0, q1);
        q0 = log(q0); q1 = log(q1);
        if (q0 == __builtin_huge_val() && q1 == __builtin_huge_val()) q0 = 0.0;
        out[0] = q0; out[1] = q1; out[2] = 0.0;
    }
}

// Register-resident mutation for models with D = n_para <= 10 (MODE 0 of k_mutate, same arithmetic in the same order).
// Everything per-particle lives in VGPRs with compile-time indices: the parameter vector x[D], and the block's z / draw /
// solve vectors (padded to D entries; identity-padded factor).  Block membership is a run-time (but wave-uniform) index,
// so gathering θ_b and scattering the proposal back are D x D chains of uniform selects instead of memory indexing -
// at ~1.5 wavefronts per SIMD (N = 1e5) every LDS / global round trip is exposed latency, a select is not.
// The block factor L, the block constants and the model constants are staged once into LDS (uniform-address reads).
template <int D, bool ALPHA1>
__global__ void __launch_bounds__(256, 3) k_mutate_reg(CloudPtrs cl, const DevState *st, const ModelDev *md, MutArgs ma,
                                                   double *acc_partials, int standalone, int nb, int nf) {
SMCMI_FP_CONTRACT
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int T = blockDim.x, tid = threadIdx.x;
    const bool profme = ma.prof != nullptr && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);
    long long *profp = ma.prof + (blockIdx.x == 0 ? 0 : 16);
    // (-DSMCMI_ISA_MARKS, profiles/isa_phases.py: the phase boundaries as comments in the ISA - the per-phase instruction census)
#ifdef SMCMI_ISA_MARKS
#define SMCMI_MARK(slot) asm volatile("; SMCMI_MARK " #slot ::: "memory")
#else
#define SMCMI_MARK(slot) (void)0
#endif
#define SMCMI_PROF(slot)                                                                                    \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        SMCMI_MARK(slot);                                                                                   \
        if (profme) { unsigned long long tt_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory"); profp[slot] = (long long)tt_; } \
    } while (0)
    SMCMI_PROF(0);
    double *Ls = sm;                                // [D*D] row-major with stride D, identity-padded
    double *mu_s = Ls + D * D;                      // [D]
    double *sdd_s = mu_s + D, *sdn_s = sdd_s + D;   // [D] each
    double *red = sdn_s + D;                        // [4]
    double *m_lo = red + 4;
    double *m_hi = m_lo + D, *m_a = m_hi + D, *m_b = m_a + D, *m_k = m_b + D;
    double *l_par = m_k + D;                        // [2 * LIK_PAR_MAX]
    double *l_dat = l_par + 2 * LIK_PAR_MAX;        // [LIK_LDS_CAP]
    double *Lraw = l_dat + LIK_LDS_CAP;             // [D*D] packed block factors as k_prepare_mutation wrote them
    double *logdet_s = Lraw + D * D;                // [D]
    double *mub_raw = logdet_s + D, *sdd_raw = mub_raw + D, *sdn_raw = sdd_raw + D;   // [D] each, block order
    int *ball_s = (int *)(sdn_raw + D);             // [D]
    int *m_fix = ball_s + D + (D & 1), *m_fam = m_fix + D;
    int *bptr_s = m_fam + D, *loff_s = bptr_s + D + 1, *ball_raw = loff_s + D;
    // ---- round 1: every uniform input of the launch, issued back to back (a dependent global round trip costs ~1 µs
    // at this occupancy, so the kernel is organised as: one round of parameter loads, one round of particle loads,
    // compute, one round of stores).  nb / nf are launch arguments so the copy extents do not depend on loaded data.
    const int done = st->done, n_steps = st->mut_steps;
    const unsigned stage = st->mut_stage;
    const double c_alpha = st->mut_alpha, phi_n = st->mut_phi;
    const double e_center = st->e_center;
    const bool es_uniform = st->do_resample != 0;       // this stage resampled: all weights are 1
    const double nrm_N = (double)st->rp.n_parts, nrm_sumw = st->sumw;       // only used with ma.normalize
    const int nrm_col = st->stage - 1, nrm_hist = st->rp.store_history;
    for (int e = tid; e < nf * nf; e += T) Lraw[e] = st->L[e];
    for (int e = tid; e < nf; e += T) {
        mub_raw[e] = st->mu_b[e]; sdd_raw[e] = st->sd_draw[e]; sdn_raw[e] = st->sd_dens[e]; ball_raw[e] = st->blocks_all[e];
    }
    for (int b = tid; b < nb; b += T) { loff_s[b] = st->l_off[b]; logdet_s[b] = st->logdet[b]; }
    for (int b = tid; b <= nb; b += T) bptr_s[b] = st->block_ptr[b];
    for (int k = tid; k < D; k += T) {
        m_lo[k] = md->lo[k]; m_hi[k] = md->hi[k]; m_a[k] = md->prior_a[k]; m_b[k] = md->prior_b[k]; m_k[k] = md->prior_k[k];
        m_fix[k] = md->fixed[k]; m_fam[k] = md->prior_family[k];
    }
    for (int k = tid; k < 2 * LIK_PAR_MAX; k += T) l_par[k] = md->lik[k / LIK_PAR_MAX].par[k % LIK_PAR_MAX];
    const LikDev ld0 = md->lik[0], ld1 = md->lik[1];
    const int has_other = md->has_other_priors;
    if (!standalone && done) return;
    SMCMI_PROF(1);
    // ---- round 2: the particle and the likelihood data
    const long long i = (long long)blockIdx.x * T + tid;
    const bool live = i < cl.n;
    constexpr int src = 0;
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    double like = 0.0, lprior = 0.0, like_prev = 0.0, accept = 0.0;
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = 0.0;
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = col(cl, src, k)[i];
        like = col(cl, src, D)[i]; lprior = col(cl, src, D + 1)[i]; like_prev = col(cl, src, D + 2)[i];
    }
    double w_part = (live && (ma.esum || ma.normalize)) ? ((ma.normalize && ma.wt) ? ma.wt[i] : col(cl, src, D + 4)[i]) : 0.0;
    if (ma.normalize) {
        w_part = (w_part * nrm_N) / nrm_sumw;                             // W·N then /ΣW̃, two roundings like the reference
        if (live) {
            col(cl, src, D + 4)[i] = w_part;
            if (ma.hist_W && nrm_hist) ma.hist_W[(long long)nrm_col * ma.hist_ld + i] = w_part;
        }
    }
    ModelView mv{D, m_fix, m_fam, m_lo, m_hi, m_a, m_b, m_k};
    LikView lv[2];
    {
        int used = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const LikDev &ld = q == 0 ? ld0 : ld1;
            const long long nd = ld.rows * ld.cols, na = ld.aux_rows * ld.aux_cols;
            const bool fits = ld.family >= 0 && ld.family != SMCMI_LIK_CAPM_LITERAL && used + nd + na <= LIK_LDS_CAP;   // (capm_literal reads its data as scalars)
            if (fits) {
                for (long long k = tid; k < nd; k += T) l_dat[used + k] = ld.data[k];
                for (long long k = tid; k < na; k += T) l_dat[used + nd + k] = ld.aux[k];
            }
            lv[q] = LikView{ld.family, l_par + q * LIK_PAR_MAX, ld.c0, fits ? l_dat + used : ld.data, ld.rows, ld.cols,
                            fits ? l_dat + used + nd : ld.aux, ld.aux_rows, ld.aux_cols};
            if (fits) used += (int)(nd + na);
        }
    }
    SMCMI_PROF(2);
    auto XN = [&](int k) { return x[k]; };
    __shared__ double mixzt[ALPHA1 ? 1 : 256 * D];             // private z columns of the diagonal component's draw
    for (int step = 0; step < n_steps; ++step) {
        for (int b = 0; b < nb; ++b) {
            if (nb > 1 || step == 0) __syncthreads();   // raw copies (first pass) / previous block's readers (later passes)
            const int p0 = bptr_s[b], db = bptr_s[b + 1] - p0;
            if (nb > 1 || step == 0) {              // expand this block's constants to the padded D x D form
                const double *L = Lraw + loff_s[b];
                if constexpr (ALPHA1) {
                    // α = 1 only ever draws θ_b + L z.  Store the factor with its rows scattered to NATURAL parameter
                    // order, M[k][:] = L[e][:] for k = blocks_all[e] (zero rows elsewhere): the proposal becomes
                    // x_k + Σ_e M[k][e] z_e with compile-time indices - no gather / scatter of the parameter vector at all
                    // (the added zero terms do not change any sum).
                    for (int e = tid; e < D * D; e += T) Ls[e] = 0.0;
                    __syncthreads();
                    for (int e = tid; e < db * db; e += T) {
                        const int r = e / db, cidx = e % db;
                        if (cidx <= r) Ls[cidx * D + ball_raw[p0 + r]] = L[r * db + cidx];   // transposed: Ls[e][k] = M[k][e]
                    }
                }
                __syncthreads();
            }
            SMCMI_PROF(3);
            if (!live) continue;
            const unsigned t = (unsigned)(step * nb + b);
            double step_prob, u_dummy;     // MH uniform for this decision: drawn "before" the proposal (quirk Q3)
            double uc, unext;
            double z[D];
#ifdef SMCMI_COUNT_RNG_AHEAD   // profiles/isa_count.py: instruction mix of the path that loads the draws (dead-codes the in-kernel RNG)
            constexpr bool z_only = true;
#else
            constexpr bool z_only = false;
#endif
            if (z_only || (ma.zbuf && (int)t < ma.z_ahead)) {
                // drawn ahead by the idle CUs during k_prepare_mutation (RngAhead): D + 2 coalesced loads
                const double *zt = ma.zbuf + (long long)t * (D + 2) * cl.n + i;
                step_prob = zt[0];
                uc = zt[cl.n];
#pragma unroll
                for (int e = 0; e < D; ++e) z[e] = zt[(long long)(2 + e) * cl.n];
            } else {
            if (t == 0) uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, 0xFFFFFu, 0), step_prob, u_dummy);
            else uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t - 1, 0), u_dummy, step_prob);
            // ---- mvnormal_mixture_draw (src/helpers.jl:87-100)
            uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t, 0), uc, unext);
            {
                // Box-Muller for the block, written stage by stage over all pairs so the independent log / sqrt / sincospi
                // chains can be interleaved by the scheduler (at <= 2 wavefronts per SIMD dependent FP64 latency is exposed)
                constexpr int NP2 = (D + 1) / 2;
                constexpr int GRP = 3;                     // pairs interleaved at a time (more raises register pressure past 2 waves/SIMD)
                double ua[NP2], ub[NP2], rr[NP2], sn[NP2], cs[NP2];
#pragma unroll
                for (int q = 0; q < NP2; ++q) {
                    ua[q] = 0.5; ub[q] = 0.0;
                    if (2 * q < db) uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t, 1 + q), ua[q], ub[q]);
                }
#ifdef SMCMI_ISA_MARKS
                __builtin_amdgcn_sched_barrier(0);
                SMCMI_MARK(31);
#endif
#pragma unroll
                for (int g0 = 0; g0 < NP2; g0 += GRP) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = g0; q < g0 + GRP && q < NP2; ++q) rr[q] = bx_neg2log(ua[q]);
#pragma unroll
                    for (int q = g0; q < g0 + GRP && q < NP2; ++q) rr[q] = bx_sqrt(rr[q]);
#pragma unroll
                    for (int q = g0; q < g0 + GRP && q < NP2; ++q) bx_sincos2pi(ub[q], &sn[q], &cs[q]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NP2; ++q) {
                    z[2 * q] = (2 * q < db) ? rr[q] * cs[q] : 0.0;
                    if (2 * q + 1 < D) z[2 * q + 1] = (2 * q + 1 < db) ? rr[q] * sn[q] : 0.0;
                }
#pragma unroll
                for (int e = 0; e < D; ++e) asm volatile("" : "+v"(z[e]));   // materialise the normals here (see the matvec note)
            }
            }
            SMCMI_PROF(4);
            double prior_new = SMCMI_NEG_INF, like_new = SMCMI_NEG_INF, like_old_data = SMCMI_NEG_INF;
            double q0 = 0.0, q1 = 0.0;
            double xo[D];
            if constexpr (ALPHA1) {
                double zz2 = 0.0;
#pragma unroll
                for (int e = 0; e < D; ++e) zz2 += z[e] * z[e];
                // q0 = q1 = log N(θ_b; ϑ_b, c²Σ) bit for bit (sign-symmetric quadratic form), other mixture terms have weight 0:
                // q0 - q1 == 0 unless exp() underflows to 0 (log-density < -745.13: the reference gets NaN and rejects).
                q1 = (-((double)db * LOG2PI + logdet_s[b] + zz2) / 2.0 < -745.1332191019412) ? __builtin_nan("") : 0.0;
                SMCMI_PROF(5);
                // column sweep: D independent accumulators (one per parameter), one factor column live at a time; every
                // accumulator still adds its terms in ascending e, i.e. the same order as the row-wise triangular product
                double sacc[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { xo[k] = x[k]; sacc[k] = 0.0; }
#pragma unroll
                for (int e = 0; e < D; ++e) {
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc[k] += Ls[e * D + k] * z[e];
                    // pin the accumulators here: otherwise the optimiser sinks all FMAs to the use site and keeps the whole
                    // factor (100 LDS values = 200 VGPRs) live, which costs the second wavefront per SIMD
#pragma unroll
                    for (int k = 0; k < D; ++k) asm volatile("" : "+v"(sacc[k]));
                }
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xo[k] + sacc[k];
                SMCMI_PROF(6);
                if (ma.debug & 2) { prior_new = lprior - 0.1 * zz2; like_new = like - 0.2; like_old_data = 0.0; }
                else if (in_bounds_s<D>(mv, XN)) {
                    prior_new = logprior_s<D>(mv, XN, has_other);
                    like_new = loglik_s<D>(lv[0], XN);
                    if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                    like_old_data = (lv[1].family == SMCMI_LIK_NONE) ? 0.0 : loglik_s<D>(lv[1], XN);
                }
            } else {
            // mixture draw + proposal densities in parameter order (mix_propose above): no gather / scatter, no division
            double zz2 = 0.0;
#pragma unroll
            for (int e = 0; e < D; ++e) zz2 += z[e] * z[e];
            SMCMI_PROF(5);
            double xn[D];
            // (this block's dense matrices, written by k_mix_prepare, read as wave-uniform scalars)
            const MixDenseC<D> MX((mix_cdp)(unsigned long long)(ma.mix + (long long)b * MixDenseC<D>::DOUBLES), (mix_cip)(unsigned long long)(ma.mixpos + b * D));
            q0 = mix_propose<D, 256>(MX, x, z, uc, c_alpha, (ma.debug & 4) != 0, mixzt + tid, xn);       // q0 - q1 as one number
            SMCMI_PROF(6);
#pragma unroll
            for (int k = 0; k < D; ++k) { xo[k] = x[k]; x[k] = xn[k]; }
            if (ma.debug & 2) { prior_new = lprior - 0.1 * zz2; like_new = like - 0.2; like_old_data = 0.0; }
            else if (in_bounds_s<D>(mv, XN)) {
                prior_new = logprior_s<D>(mv, XN, has_other);
                like_new = loglik_s<D>(lv[0], XN);
                if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                like_old_data = (lv[1].family == SMCMI_LIK_NONE) ? 0.0 : loglik_s<D>(lv[1], XN);
            }
            }
            SMCMI_PROF(7);
            const double eta = exp(phi_n * (like_new - like) + (1.0 - phi_n) * (like_old_data - like_prev) +
                                   (prior_new - lprior) + (q0 - q1));
            if (step_prob < eta) {
                like = like_new; lprior = prior_new; like_prev = like_old_data;
                accept += (double)db;
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xo[k];
            }
        }
    }
    SMCMI_PROF(8);
    double acc_val = 0.0;
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) col(cl, src, k)[i] = x[k];
        col(cl, src, D)[i] = like;
        col(cl, src, D + 1)[i] = lprior;
        col(cl, src, D + 2)[i] = like_prev;
        acc_val = accept / (double)nf;                      // quirk Q2: normalised by n_free only
        col(cl, src, D + 3)[i] = acc_val;
    }
    double a1[1] = {acc_val};
    Butterfly<0, 32>::run(a1, tid & 63);
    __syncthreads();
    // largest energy of the mutated cloud (energy shift of the next stage): wave maxima ride along with the acceptance sums
    __shared__ double emx[4];
    double em = energy_or_ninf(like, like_prev, (ma.esum && !es_uniform) ? w_part : 1.0, live);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) em = fmax(em, __shfl_xor(em, off, 64));
    if ((tid & 63) == 0) { red[tid >> 6] = a1[0]; emx[tid >> 6] = em; }
    if (ma.esum) {                                   // energy power sums of the mutated cloud (ϕ predictor of the next stage)
        double es[ES];
        energy_terms(es, w_part, like, like_prev, e_center, live, es_uniform);
        es[EACC] = acc_val;
        const double tot = block_reduce_es(es, l_dat, T / 64);       // likelihood data in LDS is dead by now
        if (tid < ES) ma.esum[(long long)blockIdx.x * ES + tid] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0, m = emx[0];
        for (int w = 0; w < T / 64; ++w) s += red[w];
        for (int w = 1; w < T / 64; ++w) m = fmax(m, emx[w]);
        acc_partials[blockIdx.x] = s;
        if (ma.emax) ma.emax[blockIdx.x] = m;
    }
    SMCMI_PROF(9);
#undef SMCMI_PROF
#undef SMCMI_MARK
#undef SMCMI_MARK
}

// ------------------------------------------------------------------------------------------------ initial draw
// One prior draw of parameter k (rand(::ParameterVector), src/initialization.jl:27) on the RNG contract: outer attempt in the counter's
// stage field, bounds redraw r and parameter k in the tag.  Normal / Uniform: one Philox call, tag(P_INIT, r, k).  The Gamma family
// (Gamma, Beta, InverseGamma, RootInverseGamma - the priors DSGE parameter vectors carry) by Marsaglia & Tsang's method: iteration m
// (< 32) of unit gamma g (< 2: Beta needs two) reads its normal from tag(P_INIT, r | (32 g + m) << 14, k) (Box-Muller, cosine branch)
// and its acceptance uniform u1 - with u2 for the shape < 1 boost G(a) = G(a + 1) u2^(1/a) - from tag(P_INIT, same, k | 64); r < 16384.
// (the CPU restatement used by the tests follows the same rules); NaN = no acceptance in 32 iterations (probability < 1e-40).
__device__ inline double gamma_unit_draw(unsigned long long seed, unsigned long long pid, unsigned attempt, unsigned r, unsigned k, unsigned g, double shape) {
    const bool boost = shape < 1.0;
    const double a = boost ? shape + 1.0 : shape;
    const double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    for (unsigned m = 0; m < 32u; ++m) {
        const unsigned t = (r & 0x3FFFu) | ((32u * g + m) << 14);
        double ua, ub, u1, u2;
        uniform_pair(seed, pid, attempt, rng_tag(P_INIT, t, k), ua, ub);
        const double x = sqrt(-2.0 * log(ua)) * cos(6.283185307179586476925286766559 * ub);
        double v = 1.0 + c * x;
        if (!(v > 0.0)) continue;
        v = v * v * v;
        uniform_pair(seed, pid, attempt, rng_tag(P_INIT, t, k | 64u), u1, u2);
        const double x2 = x * x;
        if (u1 < 1.0 - 0.0331 * x2 * x2 || log(u1) < 0.5 * x2 + dd * (1.0 - v + log(v))) {
            double G = dd * v;
            if (boost) G *= exp(log(u2) / shape);
            return G;
        }
    }
    return __builtin_nan("");
}
__device__ inline double prior_draw(unsigned long long seed, unsigned long long pid, unsigned attempt, unsigned r, unsigned k, int fam, double a, double b) {
    if (fam == SMCMI_PRIOR_NORMAL || fam == SMCMI_PRIOR_UNIFORM) {
        double ua, ub;
        uniform_pair(seed, pid, attempt, rng_tag(P_INIT, r, k), ua, ub);
        return fam == SMCMI_PRIOR_NORMAL ? a + b * (sqrt(-2.0 * log(ua)) * cos(6.283185307179586476925286766559 * ub)) : a + (b - a) * ua;
    }
    switch (fam) {
    case SMCMI_PRIOR_GAMMA: return b * gamma_unit_draw(seed, pid, attempt, r, k, 0u, a);                       // Gamma(shape a, scale b)
    case SMCMI_PRIOR_BETA: {
        const double g1 = gamma_unit_draw(seed, pid, attempt, r, k, 0u, a), g2 = gamma_unit_draw(seed, pid, attempt, r, k, 1u, b);
        return g1 / (g1 + g2);
    }
    case SMCMI_PRIOR_INVGAMMA: return b / gamma_unit_draw(seed, pid, attempt, r, k, 0u, a);                   // InverseGamma(shape a, scale b)
    case SMCMI_PRIOR_ROOTINVGAMMA: return sqrt(a * b * b / (2.0 * gamma_unit_draw(seed, pid, attempt, r, k, 0u, 0.5 * a)));   // sqrt(ν τ² / χ²_ν)
    default: return __builtin_nan("");
    }
}
// the redraw loop of one parameter (rand(::ParameterVector) redraws until the value is inside the bounds)
__device__ inline double prior_draw_in_bounds(unsigned long long seed, unsigned long long pid, unsigned attempt, unsigned k, int fam, double a, double b, double lo,
                                              double hi) {
    const unsigned r_max = (fam == SMCMI_PRIOR_NORMAL || fam == SMCMI_PRIOR_UNIFORM) ? 100000u : 16383u;
    for (unsigned r = 0;; ++r) {
        const double x = prior_draw(seed, pid, attempt, r, k, fam, a, b);
        if ((lo < x && x < hi) || r >= r_max) return x;
    }
}
// initial_draw! / one_draw (src/initialization.jl:23-119): prior draws with bounds rejection, re-draw until the
// log-likelihood is finite.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_init_prior(CloudPtrs cl, const DevState *st, const ModelDev *md,
                                                   unsigned long long seed, long long gid0, int *fail_flag) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= cl.n) return;
    const int d = md->d, dst = st->cur;
    const unsigned long long pid = (unsigned long long)(gid0 + i);
    double thl[MAXD];
    auto TH = [&](int k) { return thl[k]; };
    double ll = 0.0, lp = 0.0;
    for (unsigned attempt = 0;; ++attempt) {
        if (attempt > 100000u) { *fail_flag = 1; break; }
        for (int k = 0; k < d; ++k) {
            if (md->fixed[k]) { thl[k] = md->prior_a[k]; continue; }
            thl[k] = prior_draw_in_bounds(seed, pid, attempt, (unsigned)k, md->prior_family[k], md->prior_a[k], md->prior_b[k], md->lo[k], md->hi[k]);
        }
        if (in_bounds(*md, TH)) {
            ll = loglik(md->lik[0], d, TH);
            lp = logprior(*md, TH);
            if (ll == SMCMI_NEG_INF || ll != ll) ll = lp = SMCMI_NEG_INF;
        } else ll = lp = SMCMI_NEG_INF;
        if (!isinf(ll)) break;
    }
    for (int k = 0; k < d; ++k) col(cl, dst, k)[i] = thl[k];
    col(cl, dst, d)[i] = ll;
    col(cl, dst, d + 1)[i] = lp;
    col(cl, dst, d + 2)[i] = 0.0;
    col(cl, dst, d + 3)[i] = 0.0;
    col(cl, dst, d + 4)[i] = 1.0;
}
#endif

// Prior draw of the particles whose attempt[i] >= 0 - the draws k_init_prior makes on outer attempt attempt[i] (same Philox tags, same
// bounds redraws) - with the log-prior; the likelihood comes from the host (callback.hpp): initial_draw! with a user closure.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TB) k_draw_prior(CloudPtrs cl, const ModelDev *md, unsigned long long seed, long long gid0, const int *attempt) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= cl.n || attempt[i] < 0) return;
    const int d = md->d;
    const unsigned long long pid = (unsigned long long)(gid0 + i);
    double thl[MAXD];
    auto TH = [&](int k) { return thl[k]; };
    for (int k = 0; k < d; ++k) {
        if (md->fixed[k]) { thl[k] = md->prior_a[k]; continue; }
        thl[k] = prior_draw_in_bounds(seed, pid, (unsigned)attempt[i], (unsigned)k, md->prior_family[k], md->prior_a[k], md->prior_b[k], md->lo[k], md->hi[k]);
    }
    for (int k = 0; k < d; ++k) col(cl, 0, k)[i] = thl[k];
    col(cl, 0, d)[i] = 0.0;
    col(cl, 0, d + 1)[i] = in_bounds(*md, TH) ? logprior(*md, TH) : SMCMI_NEG_INF;
    col(cl, 0, d + 2)[i] = 0.0;
    col(cl, 0, d + 3)[i] = 0.0;
    col(cl, 0, d + 4)[i] = 1.0;
}
#endif

// initialize_likelihoods! (src/initialization.jl:153-186): retire loglh to old_loglh, then evaluate the (new-data) likelihood
// and the prior at every particle.  Out-of-bounds parameters give -Inf (the reference would throw ParamBoundsError here).
// KIND 0: any family, one thread per particle (TB threads).  The lgss_kalman family takes the values the mutation compares its proposals
// with from the filter the mutation runs: KIND 4 four lanes per particle (model.hpp kalman_lgss_quad; 64 particles per 256-thread block,
// dynamic LDS 64 slots), KIND 1 one thread per particle through kalman_lgss_wave - whole wavefronts call either (their structure values
// travel through DPP operands), lanes without a particle on rho = 0, sigma = 0.5.
template <int KIND>
static __global__ void __launch_bounds__(256, 1) k_initialize_likelihoods(CloudPtrs cl, const ModelDev *md) {
    static_assert(TB == 256, "one block size for the three kinds");
    if constexpr (KIND == 0) {
        const long long i = (long long)blockIdx.x * TB + threadIdx.x;
        if (i >= cl.n) return;
        const int d = md->d;
        double thl[MAXD];
        for (int k = 0; k < d; ++k) thl[k] = col(cl, 0, k)[i];
        auto TH = [&](int k) { return thl[k]; };
        col(cl, 0, d + 2)[i] = col(cl, 0, d)[i];
        double ll = SMCMI_NEG_INF, lp = SMCMI_NEG_INF;
        if (in_bounds(*md, TH)) { ll = loglik(md->lik[0], d, TH); lp = logprior(*md, TH); }
        col(cl, 0, d)[i] = ll;
        col(cl, 0, d + 1)[i] = lp;
    } else {
        extern __shared__ __attribute__((aligned(16))) double sm[];
        const int q = threadIdx.x & 3, slot_i = threadIdx.x >> 2;
        const long long i = KIND == 4 ? (long long)blockIdx.x * 64 + slot_i : (long long)blockIdx.x * TB + threadIdx.x;
        const bool live = i < cl.n;
        double thl[13];
        for (int k = 0; k < 13; ++k) thl[k] = live ? col(cl, 0, k)[i] : (k >= 8 && k < 12 ? 0.5 : 0.0);
        auto TH = [&](int k) { return thl[k]; };
        const bool inb = live && in_bounds(*md, TH);
        KalmanLL r;
        if constexpr (KIND == 4) r = kalman_lgss_quad(thl, md->lik[0].data, md->lik[0].cols, 0, md->lik[0].aux, (lds_bytes)sm + slot_i * KALMAN4_SLOT_BYTES, q);
        else {
            KalmanTheta kt;
            for (int k = 0; k < 13; ++k) kt.v[k] = thl[k];
            r = kalman_lgss_wave(kt, md->lik[0].data, md->lik[0].cols, 0, md->lik[0].aux);
        }
        if (live && (KIND != 4 || q == 0)) {
            col(cl, 0, 13 + 2)[i] = col(cl, 0, 13)[i];
            col(cl, 0, 13)[i] = inb ? r.ll : SMCMI_NEG_INF;
            col(cl, 0, 13 + 1)[i] = inb ? logprior(*md, TH) : SMCMI_NEG_INF;
        }
    }
}

// device-to-device copy of a cloud (n doubles, 16-byte aligned buffers): the runtime's blit kernel took 221 µs for the 12 MB of config 2,
// a grid-stride copy with 16-byte accesses runs at HBM speed (~10 µs)
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(256) k_copy_f64(double *dst, const double *src, long long n) {
    const long long n2 = n >> 1, stride = (long long)gridDim.x * blockDim.x;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(dst);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) d2[i] = s2[i];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}
// (a column that starts at an odd multiple of 8 bytes - the weight column of a cloud with n (R-1) odd, a caller's view with an odd offset -
// goes through the runtime's copy: the kernel's 16-byte accesses assume 16-byte alignment)
static inline void launch_copy_f64(double *dst, const double *src, long long n, hipStream_t s) {
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) != 0) { (void)hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s); return; }
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(2048, (n / 2 + 255) / 256));
    k_copy_f64<<<grid, 256, 0, s>>>(dst, src, n);
}
#endif

#ifndef SMCMI_INST_UNIT
static __global__ void k_fill(double *p, long long n, double v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
#endif

}  // namespace smcmi
