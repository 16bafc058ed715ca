// kernels.hpp - HIP kernels of the SMC stage for gfx950 (CDNA4, 64-wide wavefronts).
//
// Cloud layout in HBM: column-major n x R doubles per buffer (two ping-pong buffers), i.e. one contiguous
// array per parameter / metadata column -> every kernel reads and writes fully coalesced 8-byte lanes.
// All stage decisions are taken on the device from DevState, so a stage is a fixed launch sequence.
// Reductions are deterministic: wavefront butterfly -> LDS -> per-block partials -> fixed-order final sum.
#pragma once
#include <hip/hip_runtime.h>

#include "devstate.hpp"
#include "model.hpp"
#include "philox.hpp"

namespace smcmi {

constexpr int TB = 256;  // threads per block for streaming kernels (4 wavefronts)

// ------------------------------------------------------------------------------------------------ reductions
// Butterfly "reduce-scatter" across the 64 lanes of a wavefront: M accumulators per lane go in, and lane l
// comes out holding (in a[0]) the wavefront total of accumulator  l >> (6 - log2 M).  M-1 shuffles instead
// of 6 M for M independent all-reduces.
template <int HALF, int DIST>
struct Butterfly {
    template <int M>
    __device__ static inline void run(double (&a)[M], int lane) {
        const bool upper = (lane & DIST) != 0;
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const double keep = upper ? a[HALF + i] : a[i];
            const double send = upper ? a[i] : a[HALF + i];
            a[i] = keep + __shfl_xor(send, DIST, 64);
        }
        Butterfly<HALF / 2, DIST / 2>::run(a, lane);
    }
};
template <int DIST>
struct Butterfly<0, DIST> {
    template <int M>
    __device__ static inline void run(double (&a)[M], int) {
#pragma unroll
        for (int dist = DIST; dist >= 1; dist >>= 1) a[0] += __shfl_xor(a[0], dist, 64);
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

// Fixed-order sum of per-block partials: partials[b * m + idx], b < nb.  Called by all threads of a 1-block
// kernel; returns the total for idx = threadIdx.x (< m) in every thread with threadIdx.x < m.
// Requires blockDim.x >= m;  uses 4 interleaved slices when blockDim.x >= 4 m.
__device__ inline double final_sum(const double *partials, int nb, int m, double *scratch) {
    const int t = threadIdx.x;
    const int slices = (blockDim.x >= 4 * m) ? 4 : 1;
    const int idx = t % m, s = t / m;
    double acc = 0.0;
    if (s < slices)
        for (int b = s; b < nb; b += slices) acc += partials[(long long)b * m + idx];
    if (slices == 1) return acc;
    if (s < slices) scratch[s * m + idx] = acc;
    __syncthreads();
    double tot = 0.0;
    if (t < m) tot = ((scratch[t] + scratch[m + t]) + scratch[2 * m + t]) + scratch[3 * m + t];
    __syncthreads();
    return tot;
}

// Fixed-order total of nb scalars using the whole block (nb can be ~1e5 mutation blocks).
__device__ inline double final_sum1(const double *partials, int nb, double *scratch /* blockDim.x doubles */) {
    const int t = threadIdx.x, T = blockDim.x;
    double acc = 0.0;
    for (int b = t; b < nb; b += T) acc += partials[b];
    scratch[t] = acc;
    __syncthreads();
    for (int off = T >> 1; off >= 1; off >>= 1) {
        if (t < off) scratch[t] += scratch[t + off];
        __syncthreads();
    }
    const double tot = scratch[0];
    __syncthreads();
    return tot;
}

// chunk of particles owned by block b out of nb (contiguous, multiple of TB except the last)
__device__ inline void block_chunk(long long n, int nb, int b, long long &beg, long long &end) {
    long long per = (n + nb - 1) / nb;
    per = (per + TB - 1) / TB * TB;
    beg = (long long)b * per;
    end = beg + per < n ? beg + per : n;
    if (beg > n) beg = n;
}

struct CloudPtrs {
    double *buf[2];     // two n x R column-major buffers
    long long n;        // local particles (leading dimension)
    int R;
};

__device__ inline double *col(const CloudPtrs &c, int which, int column) { return c.buf[which] + (long long)column * c.n; }

// ------------------------------------------------------------------------------------------------ ESS passes
// One pass over (loglh, old_loglh, weight): for K candidate ϕ accumulate Σ v and Σ v², v = W exp((ϕ_n1-ϕ)old + (ϕ-ϕ_n1)ℓ)
// (src/helpers.jl:173-181, always the prior_weight == 0 formula: quirk Q4).  K = KC for the solver passes.
// FINAL = true is the correction step at the chosen ϕ_n (src/smc_main.jl:401-420): the incremental weight uses the
// prior-weight variant, the unnormalised weight W̃ = W w̃ is written back and w̃ goes to the history column.
template <int K, bool FINAL>
__global__ void __launch_bounds__(TB) k_ess_pass(CloudPtrs cl, DevState *st, double *partials, double *hist_w,
                                                 long long hist_ld) {
    __shared__ double red[(TB / 64) * 2 * K];
    __shared__ double s_c[K];
    if (st->done) return;
    const int mode = st->mode;
    if (FINAL ? (mode != MODE_FINAL) : (mode != MODE_SCAN && mode != MODE_SECTION)) return;
    const int src = st->cur;
    const double phi_prev = st->phi_prev;
    if (threadIdx.x < K) s_c[threadIdx.x] = FINAL ? st->phi_n : st->cand[threadIdx.x < st->n_valid ? threadIdx.x : st->n_valid - 1];
    __syncthreads();
    const int R = cl.R;
    const double *loglh = col(cl, src, R - 5), *old = col(cl, src, R - 3);
    double *w = col(cl, src, R - 1);
    const double pw = st->rp.pw, logp_old = st->rp.logp_old;
    const int stage_col = st->stage - 1;
    const bool hist = FINAL && st->rp.store_history && hist_w != nullptr;
    double acc[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) acc[k] = 0.0;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    for (long long i = beg + threadIdx.x; i < end; i += TB) {
        const double l = loglh[i], o = old[i], wi = w[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double phi = s_c[k];
            double inc;
            if (!FINAL || pw == 0.0) inc = exp((phi_prev - phi) * o + (phi - phi_prev) * l);
            else if (pw == 1.0) inc = exp((phi - phi_prev) * l);
            else inc = exp((phi_prev - phi) * log(exp(o - logp_old + log(1.0 - pw)) + pw) + (phi - phi_prev) * l);
            const double v = wi * inc;
            acc[k] += v;
            acc[K + k] += v * v;
            if (FINAL) {
                w[i] = v;
                if (hist) hist_w[(long long)stage_col * hist_ld + i] = inc;
            }
        }
    }
    const double tot = block_reduce_many<2 * K>(acc, red);
    if (threadIdx.x < 2 * K) partials[(long long)blockIdx.x * (2 * K) + threadIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------ ϕ solver
// Stage begin (src/smc_main.jl:378-396 + src/helpers.jl:14-20): bump the stage index, fold the previous
// mutation's acceptance sums into cloud.accept, flip the cloud buffer after a resample, pick ϕ_n from the fixed
// schedule or arm the adaptive solver with its first candidates.
__global__ void __launch_bounds__(TB) k_stage_begin(DevState *st, const double *sched, const double *acc_partials,
                                                    int acc_nb, Records rec) {
    __shared__ double scratch[TB];
    if (st->done) return;
    // Σ accept over blocks of the previous mutation (update_acceptance_rate!, src/particle.jl:466-468)
    double asum = 0.0;
    if (acc_nb > 0) asum = final_sum1(acc_partials, acc_nb, scratch);
    if (threadIdx.x != 0) return;
    if (acc_nb > 0 && st->stage > 1) {
        st->accept = asum / (double)st->rp.n_parts;
        rec.accept[st->stage - 1] = st->accept;
    }
    if (st->do_resample) { st->cur ^= 1; st->do_resample = 0; }
    if (st->phi_n >= 1.0) { st->done = 1; return; }
    const int i = st->stage + 1;
    if (i > st->rp.max_stages) { st->err = SMCMI_ERR_CAPACITY; st->done = 1; return; }
    st->stage = i;
    st->phi_prev = st->phi_n;
    if (st->rp.use_fixed_schedule) {
        st->phi_n = sched[i - 1];
        st->mode = MODE_FINAL;
        return;
    }
    double ess_now;   // ESS of the current weights = ESS(ϕ_n1)
    if (st->resampled_last) { st->ess_bar = st->rp.tempering_target * (double)st->rp.n_parts; st->resampled_last = 0; ess_now = (double)st->rp.n_parts; }
    else { st->ess_bar = st->rp.tempering_target * st->ess_prev; ess_now = st->ess_prev; }
    st->lo = st->phi_prev;
    st->glo = ess_now - st->ess_bar;
    // scan candidates: current ϕ_prop, then schedule[j], schedule[j+1], ... (1-based j; helpers.jl:29-32)
    const int K = st->rp.n_cand, n_phi = st->rp.n_phi;
    int nv = 0;
    st->cand[nv++] = st->phi_prop;
    for (int jj = st->j; jj <= n_phi && nv < K; ++jj) st->cand[nv++] = sched[jj - 1];
    st->n_valid = nv;
    st->mode = MODE_SCAN;
}

// After a solver pass: reduce the block partials and move the bracket (helpers.jl:29-32 scan, :49 root solve
// restated as K-section to floating-point resolution).
__global__ void __launch_bounds__(TB) k_phi_decide(DevState *st, const double *sched, const double *partials, int nb) {
    __shared__ double scratch[4 * 2 * KC];
    __shared__ double g[KC];
    if (st->done) return;
    const int mode = st->mode;
    if (mode != MODE_SCAN && mode != MODE_SECTION) return;
    const double tot = final_sum(partials, nb, 2 * KC, scratch);
    if (threadIdx.x < 2 * KC) scratch[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x < KC) {
        const double s1 = scratch[threadIdx.x], s2 = scratch[KC + threadIdx.x];
        g[threadIdx.x] = s1 * s1 / s2 - st->ess_bar;   // ESS(ϕ) = (Σv)²/Σv²
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int nv = st->n_valid, K = st->rp.n_cand, n_phi = st->rp.n_phi;
    if (mode == MODE_SCAN) {
        // first candidate with g < 0 ends the reference's while loop; candidate m>0 is schedule[j+m-1]
        int m = -1;
        for (int k = 0; k < nv; ++k) {
            if (!(g[k] >= 0.0)) { m = k; break; }
            if (g[k] >= 0.0 && st->cand[k] > st->lo) { st->lo = st->cand[k]; st->glo = g[k]; }
        }
        if (m >= 0) {
            st->phi_prop = st->cand[m];
            st->j += m;
            if (isnan(g[m])) { st->err = SMCMI_ERR_NAN_ESS; st->done = 1; return; }
            st->hi = st->cand[m]; st->ghi = g[m];
            st->mode = MODE_SECTION;
        } else {
            // all tested candidates keep ESS above target
            const int j_new = st->j + (nv - 1);
            st->phi_prop = st->cand[nv - 1];
            st->j = j_new;
            if (j_new > n_phi) {              // schedule exhausted: ϕ_prop == 1 and g(1) >= 0 -> ϕ_n = 1 (helpers.jl:51-53)
                st->phi_n = st->phi_prop;
                st->mode = MODE_FINAL;
                return;
            }
            int c = 0;                         // continue the scan with the next chunk (current ϕ_prop already known >= 0)
            for (int jj = j_new; jj <= n_phi && c < K; ++jj) st->cand[c++] = sched[jj - 1];
            st->n_valid = c;
            st->j = j_new + 1;                 // cand[0] is schedule[j_new]: keep "cand[m] == schedule[j + m - 1]"
            st->phi_prop = st->cand[0];
            return;
        }
    } else {
        // K-section: candidates are increasing interior points of (lo, hi)
        int m = -1;
        for (int k = 0; k < nv; ++k) {
            if (!(g[k] >= 0.0)) { m = k; break; }
        }
        if (m >= 0) { st->hi = st->cand[m]; st->ghi = g[m]; if (m > 0) { st->lo = st->cand[m - 1]; st->glo = g[m - 1]; } }
        else if (nv > 0) { st->lo = st->cand[nv - 1]; st->glo = g[nv - 1]; }
    }
    // next candidates or convergence
    const double lo = st->lo, hi = st->hi;
    int c = 0;
    double prev = lo;
    for (int k = 1; k <= K; ++k) {
        const double x = lo + (hi - lo) * ((double)k / (double)(K + 1));
        if (x > prev && x < hi) { st->cand[c++] = x; prev = x; }
    }
    if (c == 0) {   // no representable point strictly inside: root at floating-point resolution
        st->phi_n = (fabs(st->glo) <= fabs(st->ghi)) ? lo : hi;
        st->mode = MODE_FINAL;
    } else {
        st->n_valid = c;
        st->mode = MODE_SECTION;
    }
}

// After the correction pass: ESS, log-MDD increment, resample decision, step-size adaptation
// (src/smc_main.jl:427-455, src/particle.jl:362-366).  Also the exclusive prefix of the per-block weight sums
// for the resampling scan.
__global__ void __launch_bounds__(TB) k_post_correct(DevState *st, const double *partials, int nb, double *chunk_off,
                                                     Records rec) {
    __shared__ double scratch[4 * 2];
    if (st->done) return;
    if (st->mode != MODE_FINAL) {     // the fixed number of solver passes did not reach floating-point resolution
        if (threadIdx.x == 0) { st->err = SMCMI_ERR_BRACKET; st->done = 1; }
        return;
    }
    const double tot = final_sum(partials, nb, 2, scratch);
    if (threadIdx.x < 2) scratch[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double s1 = scratch[0], s2 = scratch[1], N = (double)st->rp.n_parts;
    const double ess = s1 * s1 / s2;
    const int i = st->stage;
    st->sumw = s1; st->sumw2 = s2; st->ess = ess; st->ess_prev = ess;
    st->mode = MODE_IDLE;
    rec.phi[i - 1] = st->phi_n;
    rec.ess[i - 1] = ess;
    if (isnan(ess)) { st->err = SMCMI_ERR_NAN_ESS; st->done = 1; return; }     // check_nan_ess, helpers.jl:270-305
    st->logz += log(s1 / N);
    const int rs = ess < st->rp.threshold;
    st->do_resample = rs;
    rec.resampled[i - 1] = rs;
    if (rs) { st->resamples += 1; st->resampled_last = 1; }
    const double a = st->accept, t = st->rp.target;
    st->c = st->c * (0.95 + 0.10 * exp(16.0 * (a - t)) / (1.0 + exp(16.0 * (a - t))));
    rec.c[i - 1] = st->c;
    if (rs && chunk_off) {
        double run = 0.0;
        for (int b = 0; b < nb; ++b) { chunk_off[b] = run; run += partials[2 * (long long)b]; }
    }
}

// ------------------------------------------------------------------------------------------------ resampling
// Inclusive scan of W̃/ΣW̃ (cumsum(weights ./ sum(weights)), src/resample.jl:29,47) in the block chunks of the
// correction pass; chunk offsets come from k_post_correct.
__global__ void __launch_bounds__(TB) k_scan_weights(CloudPtrs cl, const DevState *st, const double *chunk_off,
                                                     double *cum, int force) {
    __shared__ double s_tot[TB];
    if (!force && (st->done || !st->do_resample)) return;
    const double *w = col(cl, st->cur, cl.R - 1);
    const double total = st->sumw;
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    double carry = chunk_off[blockIdx.x];
    constexpr int IPT = 4;
    for (long long base = beg; base < end; base += (long long)TB * IPT) {
        // thread t owns IPT consecutive items so that the running sum follows particle order
        const long long i0 = base + (long long)threadIdx.x * IPT;
        double v[IPT], run = 0.0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) { v[k] = (i0 + k < end) ? w[i0 + k] : 0.0; run += v[k]; v[k] = run; }
        s_tot[threadIdx.x] = run;
        __syncthreads();
        // Hillis-Steele inclusive scan of the 256 thread totals
        for (int off = 1; off < TB; off <<= 1) {
            double add = (threadIdx.x >= off) ? s_tot[threadIdx.x - off] : 0.0;
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

// Ancestor of output slot k: first j with cum[j] > thr  (src/resample.jl:51-70 systematic walk, :33-41 multinomial
// findfirst).  Fall-through (reference returns 0 / nothing; reachable only by round-off) clamps to the last index.
__global__ void __launch_bounds__(TB) k_search_ancestors(const DevState *st, const double *cum, long long n_cum,
                                                         long long slot0, long long n_slots, long long n_parts_total,
                                                         int method, unsigned long long seed, unsigned stage,
                                                         const double *offsets, long long *anc, int force) {
    if (!force && (st->done || !st->do_resample)) return;
    const long long k = (long long)blockIdx.x * TB + threadIdx.x;
    if (k >= n_slots) return;
    const long long slot = slot0 + k;
    if (!force) stage = (unsigned)st->stage;
    double thr;
    if (method == SMCMI_RESAMPLE_MULTINOMIAL) {
        double ua, ub;
        if (offsets) ua = offsets[slot];
        else uniform_pair(seed, (unsigned long long)slot, stage, rng_tag(P_RES, 0, 0), ua, ub);
        thr = ua;
    } else {
        double ua, ub;
        if (offsets) ua = offsets[0];
        else uniform_pair(seed, 0ull, stage, rng_tag(P_RES, 0, 0), ua, ub);
        thr = ((double)slot + ua) / (double)n_parts_total;
    }
    long long lo = 0, hi = n_cum;   // upper_bound: first index with cum > thr
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] > thr) hi = mid; else lo = mid + 1;
    }
    anc[k] = lo < n_cum ? lo : n_cum - 1;
}

// cloud.particles = particles[new_inds, :]; reset_weights! (src/smc_main.jl:440-442).  Column-wise gather:
// coalesced writes, sorted (systematic) or random (multinomial) indexed reads.  grid.y = column.
__global__ void __launch_bounds__(TB) k_gather(CloudPtrs cl, const DevState *st, const long long *anc, int force) {
    if (!force && (st->done || !st->do_resample)) return;
    const long long k = (long long)blockIdx.x * TB + threadIdx.x;
    if (k >= cl.n) return;
    const int c = blockIdx.y, src = st->cur, dst = src ^ 1;
    double *out = col(cl, dst, c);
    if (c == cl.R - 1) { out[k] = 1.0; return; }
    out[k] = col(cl, src, c)[anc[k]];
}

// ------------------------------------------------------------------------------------------------ moments
// One pass over (θ, W̃): normalise the weights (normalize_weights!, src/particle.jl:362-366: W*N then /ΣW; or 1 after a
// resample), write them to the weight column and the W history, and accumulate the augmented second-moment matrix
// Σ w x̃ x̃ᵀ, x̃ = (1, θ - shift), from which weighted_mean / weighted_cov follow (src/particle.jl:481-483, 526-529).
// Particles are staged through LDS in tiles so every (a,b) pair is accumulated from on-chip data.
constexpr int MT = 256;                      // particles per LDS tile
__global__ void __launch_bounds__(TB) k_moments(CloudPtrs cl, DevState *st, double *partials, double *hist_W,
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
    const int src = st->cur ^ resampled;
    double *w = col(cl, src, cl.R - 1);
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
        for (int a = 0; a < d; ++a)
            xs[(a + 1) * ldx + threadIdx.x] = (i < end) ? col(cl, src, a)[i] - st->shift[a] : 0.0;
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

// Fixed-order reduction of the moment partials: totals[p] for the (d+1)(d+2)/2 pairs.  grid = ceil(npairs / 64).
__global__ void __launch_bounds__(TB) k_moments_reduce(const DevState *st, const double *partials, int nb, int npairs,
                                                       double *totals, int standalone) {
    __shared__ double scratch[4 * 64];
    if (!standalone && st->done) return;
    const int p0 = blockIdx.x * 64, m = (npairs - p0) < 64 ? (npairs - p0) : 64;
    // view: partials[b * npairs + p0 + idx]
    const int t = threadIdx.x, idx = t % 64, s = t / 64;
    double acc = 0.0;
    if (idx < m)
        for (int b = s; b < nb; b += 4) acc += partials[(long long)b * npairs + p0 + idx];
    scratch[s * 64 + idx] = acc;
    __syncthreads();
    if (t < m) totals[p0 + t] = ((scratch[t] + scratch[64 + t]) + scratch[128 + t]) + scratch[192 + t];
}

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
__global__ void __launch_bounds__(64) k_finalize_moments(DevState *st, const double *totals, int d) {
    moments_from_totals(st, totals, d, threadIdx.x, 64);
}

// generic fixed-order reduction of block partials into out[0..m) (1 block)
__global__ void __launch_bounds__(TB) k_reduce_partials(const double *partials, int nb, int m, double *out) {
    __shared__ double scratch[TB];
    if (m == 1) {
        const double tot = final_sum1(partials, nb, scratch);
        if (threadIdx.x == 0) out[0] = tot;
        return;
    }
    const double tot = final_sum(partials, nb, m, scratch);
    if (threadIdx.x < m) out[threadIdx.x] = tot;
}

// normalize_weights! as its own pass (src/particle.jl:362-366) for the stand-alone correction call
__global__ void __launch_bounds__(TB) k_normalize_weights(CloudPtrs cl, const DevState *st) {
    const long long i = (long long)blockIdx.x * TB + threadIdx.x;
    if (i >= cl.n) return;
    double *w = col(cl, st->cur, cl.R - 1);
    w[i] = (w[i] * (double)st->rp.n_parts) / st->sumw;
}

// per-chunk weight sums in the layout k_post_correct / k_scan_weights expect (partials[2 b])
__global__ void __launch_bounds__(TB) k_weight_chunk_sums(CloudPtrs cl, const DevState *st, double *partials) {
    __shared__ double red[(TB / 64) * 2];
    const double *w = col(cl, st->cur, cl.R - 1);
    double acc[2] = {0.0, 0.0};
    long long beg, end;
    block_chunk(cl.n, gridDim.x, blockIdx.x, beg, end);
    for (long long i = beg + threadIdx.x; i < end; i += TB) { const double v = w[i]; acc[0] += v; acc[1] += v * v; }
    const double tot = block_reduce_many<2>(acc, red);
    if (threadIdx.x < 2) partials[2 * (long long)blockIdx.x + threadIdx.x] = tot;
}
__global__ void k_chunk_offsets(DevState *st, const double *partials, int nb, double *chunk_off, double base,
                                int set_sum) {
    double run = base;
    for (int b = 0; b < nb; ++b) { chunk_off[b] = run; run += partials[2 * (long long)b]; }
    if (set_sum) st->sumw = run;
}
__global__ void k_flip(DevState *st) { st->cur ^= 1; }

// θ_bar, R from the totals; free subset + symmetrisation (src/smc_main.jl:457-465); random blocks
// (generate_free_blocks/all_blocks, src/helpers.jl:215-260, Fisher-Yates on Philox); then per block the scaled
// covariance c²Σ_b and its Cholesky factor - done ONCE per stage instead of per particle (src/mutation.jl:81,
// src/helpers.jl:90-94,135-155).  One block of 64 threads.
__global__ void __launch_bounds__(64) k_prepare_mutation(DevState *st, const ModelDev *md, const double *totals,
                                                         unsigned long long seed, int from_totals, int gen_blocks,
                                                         int standalone) {
    __shared__ double A[MAXD * MAXD];
    __shared__ double sig_f[MAXD * MAXD];
    __shared__ double mu_f[MAXD];
    __shared__ int s_fail;
    if (!standalone && st->done) return;
    const int d = md->d, nf = md->n_free, t = threadIdx.x;
    if (t == 0) s_fail = 0;
    if (from_totals) moments_from_totals(st, totals, d, t, 64);
    __syncthreads();
    // R_fr = (R[f,f] + R[f,f]')/2, θ_bar_fr
    for (int e = t; e < nf * nf; e += 64) {
        const int a = md->free_inds[e / nf], b = md->free_inds[e % nf];
        sig_f[e] = (st->cov[a * d + b] + st->cov[b * d + a]) / 2.0;
    }
    for (int a = t; a < nf; a += 64) mu_f[a] = st->mean[md->free_inds[a]];
    __syncthreads();
    if (gen_blocks && t == 0) {
        const int nb = st->rp.n_blocks;
        int *bf = st->blocks_free;
        for (int i = 0; i < nf; ++i) bf[i] = i;
        for (int i = nf - 1; i >= 1; --i) {
            double ua, ub;
            uniform_pair(seed, 0ull, (unsigned)st->stage, rng_tag(P_BLK, (unsigned)i, 0), ua, ub);
            int jx = (int)(ua * (double)(i + 1));
            if (jx > i) jx = i;
            const int tmp = bf[i]; bf[i] = bf[jx]; bf[jx] = tmp;
        }
        const int sub = (nf + nb - 1) / nb;
        for (int b = 0; b < nb; ++b) st->block_ptr[b] = b * sub;
        st->block_ptr[nb] = nf;
        st->n_blocks = nb;
    }
    __syncthreads();
    const int nb = st->n_blocks;
    const double c = st->c;
    if (t == 0) {
        int off = 0;
        for (int b = 0; b < nb; ++b) { st->l_off[b] = off; const int db = st->block_ptr[b + 1] - st->block_ptr[b]; off += db * db; }
    }
    for (int i = t; i < nf; i += 64) {
        const int f = st->blocks_free[i];
        st->blocks_all[i] = md->free_inds[f];
        st->mu_b[i] = mu_f[f];
        st->sd_draw[i] = sqrt(c * c * sig_f[f * nf + f]);
        st->sd_dens[i] = sqrt(sig_f[f * nf + f]);
    }
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int p0 = st->block_ptr[b], db = st->block_ptr[b + 1] - p0;
        double *L = st->L + st->l_off[b];
        for (int e = t; e < db * db; e += 64) {
            const int fa = st->blocks_free[p0 + e / db], fb = st->blocks_free[p0 + e % db];
            A[e] = c * c * sig_f[fa * nf + fb];
            L[e] = 0.0;
        }
        __syncthreads();
        // right-looking Cholesky, lane i owns row i; same operation order as the textbook column loop
        for (int jx = 0; jx < db; ++jx) {
            if (t == jx) {
                double s = A[jx * db + jx];
                for (int k = 0; k < jx; ++k) s -= L[jx * db + k] * L[jx * db + k];
                if (!(s > 0.0)) s_fail = 1;
                L[jx * db + jx] = sqrt(s);
            }
            __syncthreads();
            if (s_fail) break;
            if (t > jx && t < db) {
                double s = A[t * db + jx];
                for (int k = 0; k < jx; ++k) s -= L[t * db + k] * L[jx * db + k];
                L[t * db + jx] = s / L[jx * db + jx];
            }
            __syncthreads();
        }
        if (t == 0) {
            double ld = 0.0;
            for (int i = 0; i < db; ++i) ld += log(L[i * db + i]);
            st->logdet[b] = 2.0 * ld;
        }
        __syncthreads();
        if (s_fail) break;
    }
    if (t == 0) {
        if (s_fail) { st->err = SMCMI_ERR_POSDEF; st->done = 1; }   // PosDefException aborts the run (mutation.jl:81)
        if (!standalone) {
            st->mut_c = c; st->mut_alpha = st->rp.alpha; st->mut_phi = st->phi_n; st->mut_steps = st->rp.n_mh_steps;
            st->mut_stage = (unsigned)st->stage;
        }
    }
}

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
    const double *lik_new, *lik_old_new;   // MODE 2
    int *acc_count;            // MODE 1/2: accepted block lengths so far
    int block, step, last;     // MODE 1/2
};

template <int MODE>
__global__ void __launch_bounds__(256) k_mutate(CloudPtrs cl, const DevState *st, const ModelDev *md, MutArgs ma, double *acc_partials,
                         int standalone) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (!standalone && st->done) return;
    const int T = blockDim.x, tid = threadIdx.x;
    const int d = md->d, nf = md->n_free;
    double *th = sm;                       // current θ           [d][T]
    double *tn = th + (long long)d * T;    // proposed θ          [d][T]
    double *y = tn + (long long)d * T;     // z / draw            [d][T]
    double *v = y + (long long)d * T;      // triangular-solve scratch [d][T]
    double *red = v + (long long)d * T;    // [T/64]
    const long long i = (long long)blockIdx.x * T + tid;
    const bool live = i < cl.n;
    const int src = standalone ? st->cur : (st->cur ^ st->do_resample);
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    const unsigned stage = st->mut_stage;
    const double c_alpha = st->mut_alpha, phi_n = st->mut_phi;
    const int nb = st->n_blocks, n_steps = st->mut_steps;
    double like = 0.0, lprior = 0.0, like_prev = 0.0, accept = 0.0;
    if (live) {
        for (int k = 0; k < d; ++k) { const double x = col(cl, src, k)[i]; th[k * T + tid] = x; tn[k * T + tid] = x; }
        like = col(cl, src, d)[i]; lprior = col(cl, src, d + 1)[i]; like_prev = col(cl, src, d + 2)[i];
    }
    auto TN = [&](int k) { return tn[k * T + tid]; };
    if (live) {
        const int s_beg = (MODE == 0) ? 0 : ma.step, s_end = (MODE == 0) ? n_steps : ma.step + 1;
        for (int step = s_beg; step < s_end; ++step) {
            const int b_beg = (MODE == 0) ? 0 : ma.block, b_end = (MODE == 0) ? nb : ma.block + 1;
            for (int b = b_beg; b < b_end; ++b) {
                const int p0 = st->block_ptr[b], db = st->block_ptr[b + 1] - p0;
                const double *L = st->L + st->l_off[b];
                const unsigned t = (unsigned)(step * nb + b);
                // MH uniform for this decision: drawn "before" the proposal (quirk Q3)
                double step_prob, u_dummy;
                if (t == 0) uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, 0xFFFFFu, 0), step_prob, u_dummy);
                else uniform_pair(ma.seed, pid, stage, rng_tag(P_MUT, t - 1, 0), u_dummy, step_prob);
                double q0 = 0.0, q1 = 0.0, prior_new = SMCMI_NEG_INF, like_new = SMCMI_NEG_INF, like_old_data = SMCMI_NEG_INF;
                if (MODE != 2) {
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
                            y[e * T + tid] = th[st->blocks_all[p0 + e] * T + tid] + st->sd_draw[p0 + e] * y[e * T + tid];
                    } else {
                        for (int e = db - 1; e >= 0; --e) {   // in place: row e only needs z_0..z_e
                            double s = 0.0;
                            for (int k = 0; k <= e; ++k) s += L[e * db + k] * y[k * T + tid];
                            const double center = (comp == 0) ? th[st->blocks_all[p0 + e] * T + tid] : st->mu_b[p0 + e];
                            y[e * T + tid] = center + s;
                        }
                    }
                    // ---- compute_proposal_densities
                    const double cst = (double)db * LOG2PI + st->logdet[b];
                    double quad = 0.0;
                    for (int e = 0; e < db; ++e) {            // L⁻¹(θ_b - ϑ_b): forward == reverse density
                        double s = th[st->blocks_all[p0 + e] * T + tid] - y[e * T + tid];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad += ve * ve;
                    }
                    const double lp_sym = -(cst + quad) / 2.0;
                    q0 = c_alpha * exp(lp_sym);
                    q1 = q0;
                    double ind_pdf = 1.0;
                    for (int e = 0; e < db; ++e) {
                        const double sii = st->sd_dens[p0 + e];
                        const double z = (th[st->blocks_all[p0 + e] * T + tid] - y[e * T + tid]) / sii;
                        ind_pdf = ind_pdf / (sii * sqrt(2.0 * M_PI)) * exp(-0.5 * z * z);
                    }
                    q0 += (1.0 - c_alpha) / 2.0 * ind_pdf;
                    q1 += (1.0 - c_alpha) / 2.0 * ind_pdf;
                    double quad_s = 0.0, quad_d = 0.0;
                    for (int e = 0; e < db; ++e) {            // log N(θ_b; θ̄_b, c²Σ)
                        double s = th[st->blocks_all[p0 + e] * T + tid] - st->mu_b[p0 + e];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad_s += ve * ve;
                    }
                    for (int e = 0; e < db; ++e) {            // log N(ϑ_b; θ̄_b, c²Σ)
                        double s = y[e * T + tid] - st->mu_b[p0 + e];
                        for (int k = 0; k < e; ++k) s -= L[e * db + k] * v[k * T + tid];
                        const double ve = s / L[e * db + e];
                        v[e * T + tid] = ve;
                        quad_d += ve * ve;
                    }
                    q0 += (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_s) / 2.0);
                    q1 += (1.0 - c_alpha) / 2.0 * exp(-(cst + quad_d) / 2.0);
                    q0 = log(q0);
                    q1 = log(q1);
                    if (q0 == __builtin_huge_val() && q1 == __builtin_huge_val()) q0 = 0.0;
                    // ---- para_new
                    for (int e = 0; e < db; ++e) tn[st->blocks_all[p0 + e] * T + tid] = y[e * T + tid];
                    const bool inb = in_bounds(*md, TN);
                    if (inb) prior_new = logprior(*md, TN);
                    if (MODE == 1) {
                        for (int k = 0; k < d; ++k) ma.proposals[(long long)k * cl.n + i] = TN(k);
                        ma.prop_logprior[i] = prior_new;
                        ma.prop_qdiff[i] = q0 - q1;
                        continue;
                    }
                    if (inb) {
                        like_new = loglik(md->lik[0], d, TN);
                        if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                        like_old_data = (md->lik[1].family == SMCMI_LIK_NONE) ? 0.0 : loglik(md->lik[1], d, TN);
                    }
                } else {
                    for (int k = 0; k < d; ++k) tn[k * T + tid] = ma.proposals[(long long)k * cl.n + i];
                    prior_new = ma.prop_logprior[i];
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
                if (step_prob < eta) {
                    if (MODE == 2) { for (int k = 0; k < d; ++k) th[k * T + tid] = tn[k * T + tid]; }
                    else for (int e = 0; e < db; ++e) th[st->blocks_all[p0 + e] * T + tid] = y[e * T + tid];
                    like = like_new; lprior = prior_new; like_prev = like_old_data;
                    accept += (double)db;
                } else if (MODE != 2) {
                    for (int e = 0; e < db; ++e) tn[st->blocks_all[p0 + e] * T + tid] = th[st->blocks_all[p0 + e] * T + tid];
                }
            }
        }
    }
    if (MODE == 1) return;
    double acc_val = 0.0;
    if (live) {
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
    double a1[1] = {acc_val};
    Butterfly<0, 32>::run(a1, tid & 63);
    if ((tid & 63) == 0) red[tid >> 6] = a1[0];
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < T / 64; ++w) s += red[w];
        acc_partials[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------ initial draw
// initial_draw! / one_draw (src/initialization.jl:23-119): prior draws with bounds rejection, re-draw until the
// log-likelihood is finite.  Normal / Uniform priors only (others: host draws + upload).
__global__ void __launch_bounds__(TB) k_init_prior(CloudPtrs cl, const DevState *st, const ModelDev *md,
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
            for (unsigned r = 0;; ++r) {
                double ua, ub, x;
                uniform_pair(seed, pid, attempt, rng_tag(P_INIT, r, (unsigned)k), ua, ub);
                if (md->prior_family[k] == SMCMI_PRIOR_NORMAL)
                    x = md->prior_a[k] + md->prior_b[k] * (sqrt(-2.0 * log(ua)) * cos(6.283185307179586476925286766559 * ub));
                else x = md->prior_a[k] + (md->prior_b[k] - md->prior_a[k]) * ua;
                if ((md->lo[k] < x && x < md->hi[k]) || r > 100000u) { thl[k] = x; break; }
            }
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

__global__ void k_fill(double *p, long long n, double v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace smcmi
