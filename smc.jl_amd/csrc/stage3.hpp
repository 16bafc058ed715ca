// stage3.hpp - engine 3: persistent stage segments for small clouds on one handle (n_para <= 10, one 512-particle block per CU).
//
// Engine 2 (stage2.hpp) pays, per tempering stage (src/smc_main.jl:377-508), two launch boundaries, two kernel starts, the reload of
// the cloud from HBM / MALL and, in EVERY block, the serial set-up between the hand-overs (row totals, Newton solve of ϕ_n,
// covariance -> Cholesky): 44 µs per stage at N = 1e5 of which 9 are the bodies.  A run of consecutive stages that neither resample
// nor need the certificate path (95 % of an adaptive run) is ONE launch here:
//
//   * WORKER blocks (one per 512 particles, engine 2's mutation-block geometry): every thread keeps its particle (θ, loglh, logprior,
//     old_loglh, weight) in registers for the whole segment - HBM sees only the history columns and one row per block and phase;
//   * the two chip-wide hand-overs of a stage are neither launches nor barriers nor tickets: a worker publishes its row (the same row,
//     bit for bit, K1 / K2 of engine 2 would store for the same particles) as tagged 8-byte granules {32-bit payload | 32-bit tag} - a
//     reader that sees the tag has the payload, no fence, no flag, no counter (the mailbox's wire format, agent scope here);
//     one GATHERER block per virtual shard (on a CU the workers leave idle, on the shard's own XCD) sweeps its shard's rows, totals
//     them in the canonical order and publishes the shard total the same way; EVERY block - gatherers included - then fetches the V
//     shard totals and derives the stage's decisions itself (decide2, post2, begin2_wave; the workers also proposal2): same inputs,
//     same code, same result everywhere, as in engine 2's kernels; worker 0 records them.  Two store -> load hops per hand-over, no
//     atomics (a first version with per-shard tickets and "last arriver totals" needed nine dependent memory round trips of ~1.5 µs
//     each; one decider block that published the decision as a record cost a third hop: 37.8 against 37.1 µs per stage);
//   * the workers draw the NEXT stage's random numbers while they wait for its begin (they depend on (seed, particle, stage) only).
//
// Rows, totals, decision logic (begin2_wave, decide2, post2, proposal2) and the MH body (k2_mh_steps) are engine 2's own functions:
// a segment leaves the bits engine 2's launches would leave, and the two engines alternate freely inside a run - the segment ends
// (state in memory exactly as between two engine-2 stages: cloud in buffer 0, mutation rows, Post2 / Begin2 in Ctl2) when its last
// stage is done or as soon as a stage needs what it cannot do (selection, a certificate pass, the end of the run); the host then
// runs that stage through engine 2's launches and starts the next segment.  Every block is resident for the whole launch (grid <=
// one block per CU, checked by a residency self-test on first use), so a waiting block can only wait for blocks that are running;
// every wait is bounded (time-out -> SMCMI_ERR_TIMEOUT, never a hung GPU).
#pragma once
#include "stage2.hpp"

namespace smcmi {

#ifndef SMCMI_K3_WAVES
#define SMCMI_K3_WAVES 2          // (launch bound: wavefronts per SIMD the segment kernel is compiled for; 2 = one 512-thread block per CU)
#endif
constexpr int T3 = 512;                      // threads = particles of a block
constexpr int TICK3_STRIDE = 32;             // ints between ticket counters: one 128-byte line each (the arrivals of different shards do not queue behind each other)
constexpr int SEG3_TICKS = (V2_MAXV + 1) * TICK3_STRIDE;      // per kind: one ticket counter per local virtual shard + the top one

// ---- records: tagged granules in device memory (one writer block, every block reads)
struct RecA3 {                               // after the mutation of stage n - 1: how stage n begins
    int act, pad;                            // 0 go on with stage bg.stage; 1 / 5 / 9 / 4: begin2_wave's codes (status written); 7: segment complete
    Begin2 bg;
};
template <int D>
struct Prop3 {                               // the proposal of a stage (what proposal2 leaves in LDS)
    double Lraw[D * D], logdet[D], mub[D], sdd[D], sdn[D];
    int ball[D + (D & 1)], bptr[D + 2 + (D & 1)], loff[D + (D & 1)];
};
template <int D>
struct RecB3 {                               // after the correction of stage n: go (Post2 of n + proposal) or leave the segment
    int act, pad;                            // 0 go; 6 the stage needs the full path (selection / unverified prediction): nothing of it is committed; 9 error
    Post2 po;
    Prop3<D> pr;
};
static_assert(sizeof(RecA3) % 4 == 0 && sizeof(RecB3<10>) % 4 == 0, "records are copied as 32-bit words");
constexpr size_t REC3_WORDS = (sizeof(RecA3) + sizeof(RecB3<10>)) / 4 + 16;       // granules a handle allocates (RecA at 0, RecB behind it)
constexpr int REC3_B_OFF = (int)(sizeof(RecA3) / 4) + 4;

// all threads call; false: a wait timed out (flag words `to` as the mailbox's: to[0] sticky flag, to[1] ticks).
// ONE lane of the block polls (word 0, with a sleep between probes): seventy thousand threads probing the same two dozen cache lines
// would queue every other memory access of the chip - the deciding block's included - behind their probes.  When word 0 is there the
// rest is at most a store queue behind it: every thread then fetches its own words, probing again in the rare case one is not there yet.
__device__ inline bool rec3_wait(const unsigned long long *rec, void *dst_lds, int nwords, unsigned tag, unsigned long long *to, int *s_to) {
    unsigned *w = reinterpret_cast<unsigned *>(dst_lds);
    auto probe = [&](int k, int nap) -> unsigned long long {
        unsigned long long a = __hip_atomic_load(rec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(a >> 32) != tag) {
            const long long t0 = wall_clock64();
            const long long lim = (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            do {
                if (nap) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(1);
                a = __hip_atomic_load(rec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(a >> 32) == tag) break;
                if (wall_clock64() - t0 > lim || __hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_to = 1;
                    break;
                }
            } while (true);
        }
        return a;
    };
    if (threadIdx.x == 0) w[0] = (unsigned)probe(0, 1);
    __syncthreads();
    if (*s_to) return false;
    for (int k = threadIdx.x; k < nwords; k += blockDim.x)
        if (k != 0) w[k] = (unsigned)probe(k, 0);
    __syncthreads();
    return *s_to == 0;
}

// ---- rows and shard totals as granules: a double = two tagged words
__device__ inline void gran_store(unsigned long long *w, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, ((unsigned long long)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same into another handle's table (fine-grained memory, possibly another GPU's over xGMI): system scope, as the mailbox's mb_store
__device__ inline void gran_store_sys(unsigned long long *w, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, ((unsigned long long)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one lane's bounded wait for ONE word to carry `tag` (the sweep proper re-checks every word it uses); sys: the word is written by peers
__device__ inline void gran_poll(const unsigned long long *w, unsigned tag, unsigned long long *to, int *s_to, bool sys = false) {
    auto ld = [&]() { return sys ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    unsigned long long a = ld();
    if ((unsigned)(a >> 32) == tag) return;
    const long long t0 = wall_clock64();
    const long long lim = (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    do {
        __builtin_amdgcn_s_sleep(1);
        a = ld();
        if ((unsigned)(a >> 32) == tag) return;
        if (wall_clock64() - t0 > lim || __hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_to = 1;
            return;
        }
    } while (true);
}
// Gatherer: totals of the nr rows (m doubles = 2 m words each, at tbl) of one virtual shard in the canonical order -> thread t < m
// returns total t.  One lane per row waits for the row's first word; then the WHOLE block fetches the table - consecutive threads
// consecutive granules, every load of a thread in flight at once - checks EVERY tag and leaves the doubles in LDS (`stage`, nr * m
// doubles), where reduce_vshard_f does its additions in the canonical order.  (Until round 5 the reduction's own units loaded the rows:
// 144 threads with 32 loads each, two thirds of them the clamped last row - 4.9 µs per sweep against ~1 here.)  A word that was not
// there yet repeats the sweep (rows arrive within a fraction of a µs of each other).  false: timed out.
// post(idx, total): called by the thread that ends up with total idx (two columns per thread).  nr <= 2 GRP; per canonical group of GRP rows
// reduce_vshard_f's arithmetic (slices 2h, 2h+1 in ascending row order, the tree across the quad, 0 + group) without its hand-over of the
// group totals through LDS and the two barriers around it.
template <int NT, class POST>
__device__ inline bool gather_vshard(const unsigned long long *tbl, int nr, int m, int max_idx, unsigned tag, unsigned long long *to, int *s_to, POST post,
                                     double *stage, long long *pw = nullptr) {
    // (one lane per row waits for the row's first word before the sweep: sweeping as the probe - one round trip less on paper - was
    // measured 4 µs per stage SLOWER: the gatherers' repeated sweeps queue in front of the workers' row stores)
    if ((int)threadIdx.x < nr) gran_poll(tbl + (long long)threadIdx.x * m * 2, tag, to, s_to);
    __syncthreads();
    if (*s_to) return false;
    if (pw && threadIdx.x == 0) { pw[0] = wall_clock64(); pw[1] = 0; }
    const int cnt = nr * m;                                     // granules (16 bytes: {low word | tag}, {high word | tag})
    constexpr int U = 5;                                        // loads per thread and batch: the 31 rows x 72 columns of an 8-shard cloud are ONE batch
    const __amdgpu_buffer_rsrc_t rsrc = rows_rsrc(reinterpret_cast<const double *>(tbl), (long long)cnt * 16);
    const long long t_begin = wall_clock64();
    for (;;) {
        int bad = 0;
        if (pw && threadIdx.x == 0) pw[1] += 1;
        for (int b0 = 0; b0 < cnt; b0 += U * NT) {              // (block-uniform; a second batch only with fewer, longer virtual shards: <= 64 rows)
            u32x4_t gq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = b0 + u * NT + (int)threadIdx.x;
                gq[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (e < cnt ? e : cnt - 1) * 16, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = b0 + u * NT + (int)threadIdx.x;
                if (e < cnt) {
                    bad |= (gq[u].y != tag) | (gq[u].w != tag);
                    stage[e] = __hiloint2double((int)gq[u].z, (int)gq[u].x);
                }
            }
        }
        if (!__syncthreads_or(bad)) break;
        // a word was not there yet: sweep again, bounded like every other wait
        if (threadIdx.x == 0 && (wall_clock64() - t_begin > (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                                 __hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_to = 1;
        }
        __syncthreads();
        if (*s_to) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    const int u = threadIdx.x;
    if (u < (m / 2) * 4) {                                      // (whole quads)
        const int h = u & 3, pr = u >> 2;
        const bool mx0 = 2 * pr == max_idx, mx1 = 2 * pr + 1 == max_idx;
        const double ninf = -__builtin_inf(), id0 = mx0 ? ninf : 0.0, id1 = mx1 ? ninf : 0.0;
        // one canonical group of rows [r_beg, r_end): slices 2h, 2h+1 in ascending row order, the tree across the quad
        auto group = [&](int r_beg, int r_end, double &p0, double &p1) __attribute__((always_inline)) {
            double a0[2] = {id0, id0}, a1[2] = {id1, id1};
#pragma unroll
            for (int j = 0; j < GRP / 8; ++j) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = r_beg + 2 * h + q + 8 * j;
                    const double2 x = reinterpret_cast<const double2 *>(stage + (r < r_end ? r : r_end - 1) * m)[pr];
                    const double x0 = r < r_end ? x.x : id0, x1 = r < r_end ? x.y : id1;
                    a0[q] = mx0 ? fmax(a0[q], x0) : a0[q] + x0;
                    a1[q] = mx1 ? fmax(a1[q], x1) : a1[q] + x1;
                }
            }
            p0 = mx0 ? fmax(a0[0], a0[1]) : a0[0] + a0[1]; p1 = mx1 ? fmax(a1[0], a1[1]) : a1[0] + a1[1];
            const double q0 = fetch_xor<1>(p0), q1 = fetch_xor<1>(p1);
            p0 = mx0 ? fmax(p0, q0) : p0 + q0; p1 = mx1 ? fmax(p1, q1) : p1 + q1;
            const double r0 = fetch_xor<2>(p0), r1 = fetch_xor<2>(p1);
            p0 = mx0 ? fmax(p0, r0) : p0 + r0; p1 = mx1 ? fmax(p1, r1) : p1 + r1;
        };
        double p0, p1;
        group(0, nr < GRP ? nr : GRP, p0, p1);
        double run0 = mx0 ? fmax(ninf, p0) : 0.0 + p0, run1 = mx1 ? fmax(ninf, p1) : 0.0 + p1;
        // (a shard of more than GRP rows - several handles with 32 769 .. 65 536 particles per virtual shard - is two canonical groups, added to
        // 0 in ascending order like reduce_vshard_f's: until round 6 the rows beyond the first group were silently left out.  A branch of its
        // own: as a loop over the groups it cost the one-group sweep of the headline 0.4 µs per stage)
        if (__builtin_expect(nr > GRP, 0)) {
            group(GRP, nr, p0, p1);
            run0 = mx0 ? fmax(run0, p0) : run0 + p0; run1 = mx1 ? fmax(run1, p1) : run1 + p1;
        }
        if (h == 0) {
            post(2 * pr, run0);
            post(2 * pr + 1, run1);
        }
    }
    return true;
}
// Decider: totals over the nvs shard totals (granules at tbl[v * m * 2]) in the order of reduce_rows (0 + x_0 + x_1 + ...; maximum for
// max_idx) -> tot[0, m).  All threads call (a block of at least nvs wavefronts); ends with a barrier.  false: timed out.
// Wavefront v takes shard v: its first lane waits for the shard's first word, the wavefront then fetches the shard's m granules (checking
// EVERY tag; a word not there yet repeats the fetch) into vt[v * m ..] - a shard that is early is in LDS before the last one arrives, and
// no block barrier stands between the wait and the fetch; after the one barrier thread k adds column k over the shards in ascending order.
// sys: the table is this handle's copy in fine-grained memory, posted into by the gatherers of every handle (system-scope loads);
// vt: LDS, V2_MAXV * m doubles; vt_out (one block): the per-shard values as plain doubles [nvs][m] for the launch behind the segment
// RPS = 2: tbl is the workers' ROW table of a cloud whose virtual shards are two blocks each (the reference's default 5 000 particles: up to
// 8 192) - wavefront v fetches shard v's two rows and totals them as gather_vshard would (a row per slice: 0 + x; their sum; the identity for
// the six missing slices; 0 + that), so no worker waits for a gatherer.  (Four rows per shard the same way, inline, cost the stage loop 14
// spilled registers - config 2 8.18 -> 8.41 ms - and a non-inlined fetch of up to eight rows was no faster than the gatherers: measured, dropped.)
template <int RPS = 1>
__device__ inline bool gather_totals(const unsigned long long *tbl, int nvs, int m, int max_idx, unsigned tag, unsigned long long *to, int *s_to, double *tot,
                                     double *vt, bool sys = false, double *vt_out = nullptr) {
    const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (w < nvs) {                                                          // (wave-uniform)
        const unsigned long long *row = tbl + (long long)w * RPS * m * 2;
        // (the paced poll of the first word comes first: fetching the whole shard as the probe - one round trip less on paper - queues 200 blocks' loads in
        // front of the gatherers' posts: config 2 8.28 -> 8.42 ms, round 6)
        if (lane < RPS) gran_poll(row + (long long)lane * m * 2, tag, to, s_to, sys);
        const __amdgpu_buffer_rsrc_t rsrc = rows_rsrc(reinterpret_cast<const double *>(row), (long long)RPS * m * 16);
        const long long t_begin = wall_clock64();
        for (;;) {
            int bad = 0;
            u32x4_t xs[2][RPS];                                             // m <= 128
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = lane + 64 * q;
#pragma unroll
                for (int r = 0; r < RPS; ++r) {
                    const int off = (r * m + (k < m ? k : m - 1)) * 16;
                    xs[q][r] = sys ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, /*sc0 sc1: system scope*/ 17) : __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = lane + 64 * q;
                if (k < m) {
                    if constexpr (RPS == 1) {
                        bad |= (xs[q][0].y != tag) | (xs[q][0].w != tag);
                        vt[w * m + k] = __hiloint2double((int)xs[q][0].z, (int)xs[q][0].x);
                    } else {
                        bad |= (xs[q][0].y != tag) | (xs[q][0].w != tag) | (xs[q][1].y != tag) | (xs[q][1].w != tag);
                        const double x0 = __hiloint2double((int)xs[q][0].z, (int)xs[q][0].x), x1 = __hiloint2double((int)xs[q][1].z, (int)xs[q][1].x);
                        const bool mx = k == max_idx;
                        const double ninf = -__builtin_inf();
                        double p;
                        if (mx) { p = fmax(fmax(ninf, x0), fmax(ninf, x1)); p = fmax(p, ninf); p = fmax(p, ninf); p = fmax(ninf, p); }
                        else { p = (0.0 + x0) + (0.0 + x1); p = p + 0.0; p = p + 0.0; p = 0.0 + p; }
                        vt[w * m + k] = p;
                    }
                }
            }
            if (!__any(bad)) break;
            // a word was not there yet: fetch again, bounded like every other wait (another wavefront's time-out ends this one too)
            if (*(volatile int *)s_to) break;
            if (lane == 0 && (wall_clock64() - t_begin > (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                              __hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_to = 1;
            }
            if (*(volatile int *)s_to) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (*s_to) return false;
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        const bool mx = k == max_idx;
        double t = mx ? -__builtin_inf() : 0.0;
        for (int v = 0; v < nvs; ++v) {
            const double x = vt[v * m + k];
            t = mx ? fmax(t, x) : t + x;
            if (vt_out) vt_out[v * m + k] = x;
        }
        tot[k] = t;
    }
    __syncthreads();
    return true;
}

// A riding stage (k3_rides): the totals of stage n - 1's mutation rows AND of stage n's correction rows in ONE fetch - both went out before the
// block waited for anything, and both are normally there when it looks (the mutation rows a correction and a set of draws ago): wavefront v
// fetches shard v's granules of both tables at once, all loads in flight together, checking every tag; only a fetch that finds a word missing
// falls back to the paced poll of the rows' first words.  One memory round trip where two gather_totals calls take four.  The sums are
// gather_totals' (0 + x_0 + x_1 + ... over the shards; the maximum for the mutation row's RMAX_IDX; RPS = 2: two rows per shard totalled
// as gather_vshard would).  tot_m[0, RMUT), tot_c[0, mc); vt: LDS, V2_MAXV * (72 + RMUT) doubles.  All threads call; false: timed out.
template <int RPS = 1>
__device__ inline bool gather_totals_pair(const unsigned long long *tbl_m, unsigned tag_m, const unsigned long long *tbl_c, int mc, unsigned tag_c, int nvs,
                                          unsigned long long *to, int *s_to, double *tot_m, double *tot_c, double *vt, bool sys = false, double *vt_out = nullptr) {
    const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    double *vt_c = vt, *vt_m = vt + V2_MAXV * 72;
    const double ninf = -__builtin_inf();
    // two rows of a shard as gather_vshard totals them (a row per slice: 0 + x; their sum; the identity for the six missing slices; 0 + that)
    auto two = [&](double x0, double x1, bool mx) {
        double p;
        if (mx) { p = fmax(fmax(ninf, x0), fmax(ninf, x1)); p = fmax(p, ninf); p = fmax(p, ninf); p = fmax(ninf, p); }
        else { p = (0.0 + x0) + (0.0 + x1); p = p + 0.0; p = p + 0.0; p = 0.0 + p; }
        return p;
    };
    if (w < nvs) {                                                          // (wave-uniform)
        const unsigned long long *row_m = tbl_m + (long long)w * RPS * RMUT * 2, *row_c = tbl_c + (long long)w * RPS * mc * 2;
        const __amdgpu_buffer_rsrc_t rs_m = rows_rsrc(reinterpret_cast<const double *>(row_m), (long long)RPS * RMUT * 16);
        const __amdgpu_buffer_rsrc_t rs_c = rows_rsrc(reinterpret_cast<const double *>(row_c), (long long)RPS * mc * 16);
        const long long t_begin = wall_clock64();
        for (int attempt = 0;; ++attempt) {
            int bad = 0;
            u32x4_t xc[2][RPS], xm[RPS];                                    // mc <= 128, RMUT <= 64
#pragma unroll
            for (int r = 0; r < RPS; ++r) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = lane + 64 * q;
                    const int off = (r * mc + (k < mc ? k : mc - 1)) * 16;
                    xc[q][r] = sys ? __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, /*sc0 sc1: system scope*/ 17) : __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 16);
                }
                const int offm = (r * RMUT + (lane < RMUT ? lane : RMUT - 1)) * 16;
                xm[r] = sys ? __builtin_amdgcn_raw_buffer_load_b128(rs_m, offm, 0, 17) : __builtin_amdgcn_raw_buffer_load_b128(rs_m, offm, 0, 16);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = lane + 64 * q;
                if (k < mc) {
                    const double x0 = __hiloint2double((int)xc[q][0].z, (int)xc[q][0].x);
                    bad |= (xc[q][0].y != tag_c) | (xc[q][0].w != tag_c);
                    if constexpr (RPS == 1) vt_c[w * mc + k] = x0;
                    else {
                        bad |= (xc[q][1].y != tag_c) | (xc[q][1].w != tag_c);
                        vt_c[w * mc + k] = two(x0, __hiloint2double((int)xc[q][1].z, (int)xc[q][1].x), false);
                    }
                }
            }
            if (lane < RMUT) {
                const double x0 = __hiloint2double((int)xm[0].z, (int)xm[0].x);
                bad |= (xm[0].y != tag_m) | (xm[0].w != tag_m);
                if constexpr (RPS == 1) vt_m[w * RMUT + lane] = x0;
                else {
                    bad |= (xm[1].y != tag_m) | (xm[1].w != tag_m);
                    vt_m[w * RMUT + lane] = two(x0, __hiloint2double((int)xm[1].z, (int)xm[1].x), lane == RMAX_IDX);
                }
            }
            if (!__any(bad)) break;
            if (*(volatile int *)s_to) break;
            if (attempt == 0) {                                             // a word was not there yet: wait for the rows' first words as gather_totals does
                if (lane < RPS) gran_poll(row_c + (long long)lane * mc * 2, tag_c, to, s_to, sys);
                else if (lane >= 32 && lane < 32 + RPS) gran_poll(row_m + (long long)(lane - 32) * RMUT * 2, tag_m, to, s_to, sys);
            } else {
                if (lane == 0 && (wall_clock64() - t_begin > (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                                  __hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_to = 1;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (*(volatile int *)s_to) break;
        }
    }
    __syncthreads();
    if (*s_to) return false;
    const int t = threadIdx.x;
    if (t < mc) {
        double a = 0.0;
        for (int v = 0; v < nvs; ++v) a = a + vt_c[v * mc + t];
        tot_c[t] = a;
    } else if (t >= 128 && t < 128 + RMUT) {
        const int k = t - 128;
        const bool mx = k == RMAX_IDX;
        double a = mx ? ninf : 0.0;
        for (int v = 0; v < nvs; ++v) { const double x = vt_m[v * RMUT + k]; a = mx ? fmax(a, x) : a + x; if (vt_out) vt_out[v * RMUT + k] = x; }
        tot_m[k] = a;
    }
    __syncthreads();
    return true;
}

// Residency self-test of a handle's segment geometry (first use): `grid` blocks of T3 threads with enough LDS that a CU holds ONE of
// them - the strictest placement the segment kernel can get - take a ticket; the last publishes a granule every block waits for
// (bounded).  ok counts the blocks that saw it: anything but `grid` (or a raised time-out flag) keeps the handle on engine 2.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(T3, 2) k3_census(int *tick, unsigned long long *rec, unsigned tag, unsigned long long *to, int *ok) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_to;
    __shared__ unsigned s_w[2];
    if (threadIdx.x == 0) { s_to = 0; sm[0] = 1.0; }
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(&tick[V2_MAXV * TICK3_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(&tick[V2_MAXV * TICK3_STRIDE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec, ((unsigned long long)tag << 32) | 0x5e6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool good = rec3_wait(rec, s_w, 1, tag, to, &s_to);
    if (threadIdx.x == 0 && good && s_w[0] == 0x5e6u) atomicAdd(ok, 1);
}
#endif

// dynamic LDS a gatherer needs for the rows of its virtual shard (at most 2 GRP = 128 of them, 72 or RMUT columns)
constexpr size_t k3_gather_lds_bytes(int D) {
    const size_t npf = (size_t)(D + 1) * (D + 2) / 2 + 2, mcm = npf + (npf & 1);
    return (size_t)2 * GRP * (mcm > (size_t)RMUT ? mcm : (size_t)RMUT) * sizeof(double);
}
constexpr size_t k3_park_offset(int D) { return (k2_lds_bytes(D) + 15) / 16 * 2; }          // in doubles, 16-byte aligned
// (k2's arrays | the parking area (D + 2) T3 | the particle in transit through an in-place selection: (D + 5) T3, see k3_sel_cols)
constexpr size_t k3_lds_bytes(int D, int sel_cols = 0) {
    return k3_park_offset(D) * sizeof(double) + (size_t)(D + 2) * T3 * sizeof(double) + (size_t)sel_cols * T3 * sizeof(double);
}
// columns of LDS a segment kernel gets for the particle in transit (0: the particle is parked in device memory instead, Sel3Args::transit).
// The mixture variant carries T3 x D doubles of static z columns and the dense mixture block: with them and the D + 5 columns a block
// outgrows a CU's 160 KB beyond n_para 7 (24 KB: a bound on the rest of the kernel's static arrays)
constexpr int k3_sel_cols(int D, bool alpha1) {
    return (alpha1 || 24 * 1024 + ((size_t)T3 * D + 3 * D * D + 3 * D + 2) * sizeof(double) + k3_lds_bytes(D, D + 5) <= 160 * 1024) ? D + 5 : 0;
}
// (MIX = false, α = 1: the mixture-component uniform is never read, so its Philox call is not made at all)
template <int D, bool MIX = true>
__device__ inline void k3_draw_park(double *z_park, unsigned long long seed, unsigned long long pid, unsigned stage, int db, int debug) {
    double step_prob, uc, z[D];
    draw2<D>(seed, pid, stage, 0u, db, debug, step_prob, uc, z);
    double *p = z_park + threadIdx.x;
    p[0] = step_prob;
    if constexpr (MIX) p[T3] = uc;
#pragma unroll
    for (int e = 0; e < D; ++e) p[(2 + e) * T3] = z[e];
}

struct Sel3Args {
    int method, pad;               // resampling method (SMCMI_RESAMPLE_*)
    double *cum;                   // the cum column k2_scan writes ([n])
    long long *anc;                // ancestors (or null)
    unsigned long long *g_sel, *gt_sel;   // "my particle and cum values are written": [blocks][2 * 2] / [V2_MAXV][2 * 2] granules (a hand-over with no payload)
    unsigned long long *g_gm, *gt_gm;     // moment rows of the resampled cloud: [blocks][72 * 2] / [V2_MAXV][72 * 2]
    double *transit;               // [blocks][(D + 5) * T3]: where a worker parks its particle during the selection when the kernel's LDS has no room
                                   // for it (k3_sel_cols = 0: mixture proposals beyond n_para 7); null otherwise
    // several handles (selection inside SHARDED segments): what the handles exchange lives in every handle's mailbox allocation - fine-grained
    // memory every peer has mapped - at the same word offsets (run2.hpp sel3_area): tables of granules the peers POST into (every worker its
    // chunk sum, every gatherer its shard's totals), the whole cloud's cum column (every worker writes its 512 values into every handle's copy),
    // and this handle's particles as the stage found them, which the peers READ their ancestors' rows from
    unsigned long long *const *peers;     // null: one handle
    unsigned long long *mine;      // this handle's allocation
    int world, chunk0;             // chunk0: global index of this handle's first chunk (v0 * nb1)
    long long n_loc;               // particles per handle
    long long off_cs, off_sel, off_gm;    // granules: chunk sums [V nb1], "written" totals [V2_MAXV][2], moment totals [V2_MAXV][72]
    long long off_cum, off_rows;   // doubles: cum [N], rows [(D + 4)][n_loc]
};
struct Seg3Args {
    int n_first, n_last;           // stages this launch may run
    int enter_mut;                 // 1: stage n_first was corrected (and, where needed, resampled) by engine 2's launches - the segment enters at its
                                   // mutation, doing what K2 would (k2_prologue: totals of the correction rows, decision, proposal; Mut2Args::cmrows,
                                   // gmrows, wt, sel_enqueued), so resample and certificate stages cost no mutation launch, no write-back and no
                                   // reload of the cloud
    Rows2 mrows;                   // mutation rows of stage n_first - 1 (direct view: every block totals them for the first begin)
    const double *sched;
    unsigned long long *g_cm, *g_mut;     // the workers' rows as granules: [blocks][MCM * 2] / [blocks][RMUT * 2] words
    unsigned long long *gt_cm, *gt_mut;   // the shard totals as granules: [V2_MAXV][MCM * 2] / [V2_MAXV][RMUT * 2], indexed by GLOBAL virtual shard
    // (a RIDING launch keeps two copies of these four tables - the rows' k3_copy_words(blocks) words apart, the totals' too on one handle and
    // MB_SEG_COPY_WORDS apart in the mailbox allocation of several handles - stage n's rows and totals in copy
    // n & 1: a block that publishes stage n + 1's correction row BEFORE it has read stage n's mutation totals overwrites nothing a slower block
    // may still be waiting for - a copy is rewritten two stages later, which every block can only reach through hand-overs that need the slow
    // block's next row.  The kernel forms the offset itself: it has no scalar registers for two more arguments)
    // several handles (one per GPU; stage2.hpp peer mailbox): a gatherer posts its shard's totals into EVERY handle's tables - gt_cm / gt_mut
    // are this handle's copies inside its fine-grained mailbox allocation, peers[r] + off_cm / off_mut the same tables of handle r - and
    // every block reads its own handle's copy: the segment spans the GPUs with the hand-overs it has on one (two store -> load hops, the
    // second one over xGMI), no collective call, no launch.  Worker 0 leaves the V x RMUT mutation totals of the last completed stage as
    // plain doubles (vt_mut_out) for the launches behind the segment.
    unsigned long long *const *peers;     // null: one handle
    int world;
    long long off_cm, off_mut;
    double *vt_mut_out;
    unsigned long long *rec;       // REC3_WORDS granules
    unsigned tag_base;             // launch sequence << 16 (never reused inside a handle's life without clearing the tables)
    unsigned long long *to;        // time-out flag words
    double *hist_w;
    long long hist_ld;
    int *done_out;                 // non-null: number of stages this launch completed (profiling)
    // selection inside the segment (one handle, stage3.hpp "SELECTION"): a stage that must resample does so without leaving.  What only that
    // path needs lives in device memory (kernel arguments sit in scalar registers for the whole launch: the segment kernel has none to spare)
    const struct Sel3Args *sel;    // null: the segment leaves at a stage that must resample (status code 6)
    int clear_status;              // enter_mut: this launch resumes the stage a segment left (status code 6) - block 0 clears the status once it is in
    int *note;                     // non-null (one handle): host-mapped words - block 0 leaves a copy of Ctl2 at note + 16 and then this launch's
    int note_seq;                  // sequence number at note[0] when it is done with Ctl2 (k3_leave_note)
    long long *prof;               // development only (SMCMI_PROF2=<stage>): stamps of that stage
    long long *gprof;              // ... and every block's hand-over stamps (K3_WALL)
    int prof_stage;
};
constexpr size_t k3_copy_words(int blocks) { return (size_t)blocks * (72 + RMUT + 2 + 72) * 2 + (size_t)V2_MAXV * (72 + RMUT + 2 + 72) * 2; }
constexpr size_t k3_table_words(int blocks) { return 2 * k3_copy_words(blocks); }      // (two copies: riding launches, Seg3Args; the selection's tables use the first only)

// Block 0 of a segment launch, when it has written everything it writes into Ctl2 (all its threads call, at a block-uniform point): Ctl2 as
// the launch leaves it goes into host-mapped memory, then the launch's sequence number - the host that finds the number there has the state
// the copy-and-sync at the end of a batch would hand it, tens of microseconds earlier (run2.hpp read_ctl).
__device__ inline void k3_leave_note(const Seg3Args &sa, const Ctl2 *ctl) {
    if (blockIdx.x != 0 || sa.note == nullptr) return;
    __syncthreads();                                            // (the block's own stores into Ctl2 are in the L2 behind this barrier)
    constexpr int NW = sizeof(Ctl2) / sizeof(double);
    volatile double *dst = reinterpret_cast<volatile double *>(sa.note + 16);
    for (int k = threadIdx.x; k < NW; k += blockDim.x) dst[k] = __hip_atomic_load(reinterpret_cast<const double *>(ctl) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sa.note, sa.note_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define K3_STAMP(prof, slot)                                                                                                   \
    do {                                                                                                                      \
        if ((prof) != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && n == sa.prof_stage) {                                                        \
            unsigned long long tt_;                                                                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");       \
            (prof)[(slot)] = (long long)tt_;                                                                                   \
        }                                                                                                                     \
    } while (0)

// grid = g.Vl * g.nb2 blocks of T3 threads, every one resident; block b owns block (b / Vl) of local virtual shard (b % Vl): with the
// hardware's round-robin of consecutive blocks over the 8 XCDs a virtual shard's blocks share an XCD (its rows are totalled out of
// that die's L2 / MALL path) - placement is speed only, never correctness.
// (the shader clocks of different XCDs have different origins: a helper block's stamps compare with block 0's on XCD 0 only)
#define K3_STAMP_D(prof, slot)                                                                                                 \
    do {                                                                                                                      \
        if ((prof) != nullptr && threadIdx.x == 0 && n == sa.prof_stage) {                                                     \
            unsigned long long tt_;                                                                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");       \
            (prof)[(slot)] = (long long)tt_;                                                                                   \
        }                                                                                                                     \
    } while (0)

// (development, SMCMI_PROF2: the hand-overs of the profiled stage on the 100 MHz wall clock, which all dies share - worker b's slots at
// PROF2_WORK + 4 b, gatherer v's at PROF2_GATH + 6 v (stage2.hpp: the buffer's layout); prof2.hpp prints the spread)
#define K3_WALL(prof, idx)                                                                                                    \
    do {                                                                                                                      \
        if ((prof) != nullptr && threadIdx.x == 0 && n == sa.prof_stage) (prof)[(idx)] = wall_clock64();                       \
    } while (0)

// SELECTION inside the segment (src/smc_main.jl:435-446, src/resample.jl:23-72): what k2_scan + k2_gather do between two launches, by the
// workers on the particles they hold - the same functions on the same 512-particle blocks, hence the same cum column, ancestors and moment
// rows.  Two more hand-overs: "every particle and cum value is written" (no payload), and the moment rows of the resampled cloud.
// Not inlined, and the particle comes and goes through LDS (stx: [θ_1..θ_D][T3], sto: [W̃ | loglh | logprior | old_loglh | accept][T3]): the
// stage loop keeps its registers.  sc: the workers' parking area as scratch.  Returns 1 when a wait timed out.
template <int D>
__device__ __attribute__((noinline)) int k3_select_inside(const Sel3Args *selp, double *buf0, long long cl_n, int cl_R, long long Ng, int nchunks, int V, int rowi, long long i,
                                                          long long beg, long long end, unsigned tag, int n, unsigned long long seed, long long gid0,
                                                          const unsigned long long *g_cm, unsigned long long *to, int *s_to, double *s_tot, double *s_vt, double *s_sw,
                                                          double *red, double *sc, double *stx, double *sto, const double *shift, long long *pf, bool rows_direct, bool rows_two) {
#define K3S(k) do { if (pf && threadIdx.x == 0) pf[k] = wall_clock64(); } while (0)
    constexpr int NPm = Mut2Lds<D>::NP, MGM = pad2(NPm), DAm = D + 1, NPF = Mut2Lds<D>::NPF, MCM = pad2(NPF);
    const Sel3Args sl = *selp;
    const int tid = threadIdx.x;
    const bool live = i < end;
    const bool sys = sl.peers != nullptr;                       // several handles: see Sel3Args
    // scratch: the tile scan's and the search's few words in front; the chunk sums and offsets (phases 2-3) under the chunk ends and the
    // staged cum values (phases 5-6).  Up to 1 024 chunks / chunk ends (several handles; one handle has at most 248)
    double *s_w = sc, *s_cs = sc + 16, *s_scr = s_cs + nchunks, *s_off = s_scr + 256, *s_ce = sc + 16;
    long long *s_r = reinterpret_cast<long long *>(sc + 8);
    const int ncg = (int)((Ng + SEL_GCH - 1) / SEL_GCH);
    double *s_cw = s_ce + ncg;
    const int cap_w = (D + 2) * T3 - 16 - ncg;
    double *rows = sys ? reinterpret_cast<double *>(sl.mine + sl.off_rows) : buf0;
    const long long rows_n = sys ? sl.n_loc : cl_n;
    double *cum = sys ? reinterpret_cast<double *>(sl.mine + sl.off_cum) : sl.cum;
    const __amdgpu_buffer_rsrc_t cl_rsrc = rows_rsrc(rows, (long long)(sys ? D + 4 : cl_R) * rows_n * 8);
    const __amdgpu_buffer_rsrc_t cum_rsrc = rows_rsrc(cum, Ng * 8);
    __syncthreads();
    // (1) my particle as stage n - 1 left it -> buffer 0 (several handles: my rows of the exchange area), other dies / handles read it below
    if (live) {
#pragma unroll
        for (int k = 0; k < D + 4; ++k) {
            const double val = k < D ? stx[k * T3 + tid] : sto[(1 + k - D) * T3 + tid];
            if (sys) __hip_atomic_store(rows + (long long)k * rows_n + i, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else row_store(rows + (long long)k * rows_n + i, val, true);
        }
    }
    K3S(0);
    // (2) the chunk sums = entry 0 of every block's correction row (published under this stage's tag) -> chunk offsets.  Several handles:
    // every worker posts its own into every handle's table first
    if (sys && tid < sl.world) {
        const unsigned long long *wd = g_cm + (long long)rowi * MCM * 2;
        const unsigned long long w0 = __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w1 = __hip_atomic_load(wd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gran_store_sys(sl.peers[tid] + sl.off_cs + (long long)(sl.chunk0 + rowi) * 2, __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0), tag);
    }
    for (int c = tid; c < nchunks; c += T3) {
        const unsigned long long *wd = sys ? sl.mine + sl.off_cs + (long long)c * 2 : g_cm + (long long)c * MCM * 2;
        gran_poll(wd, tag, to, s_to, sys);
        gran_poll(wd + 1, tag, to, s_to, sys);
        const unsigned long long w0 = sys ? __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = sys ? __hip_atomic_load(wd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(wd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cs[c] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
    }
    __syncthreads();
    if (*s_to) return 1;
    sel_chunk_offsets([&](int b) { return s_cs[b]; }, nchunks, s_scr, s_off);
    K3S(1);
    // (3) the cum values of my chunk (k2_scan's arithmetic on W̃); several handles: into every handle's copy of the column
    {
        double tt;
        const double incl = sel_tile_scan(live ? sto[tid] : 0.0, s_w, &tt);
        const double cv = (s_off[(sys ? sl.chunk0 : 0) + rowi] + incl) / s_tot[0];
        if (live) {
            if (sys) {
                for (int pr = 0; pr < sl.world; ++pr)
                    __hip_atomic_store(reinterpret_cast<double *>(sl.peers[pr] + sl.off_cum) + gid0 + i, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else row_store(cum + i, cv, true);
        }
    }
    K3S(2);
    // (4) hand-over: every store above is acknowledged before this block says so
    // (several handles: the rows and cum values went out as system-scope stores into fine-grained memory - written through, acknowledged by the
    // memory that holds them; a system-scope FENCE here would write back the die's whole L2 first: measured 20 µs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    K3S(3);
    if (tid < 2) gran_store(sl.g_sel + ((long long)rowi * 2 + tid) * 2, 0.0, tag);
    if (!(rows_two ? gather_totals<2>(sl.g_sel, V, 2, -1, tag, to, s_to, s_sw, s_vt)
                   : gather_totals(sys ? sl.mine + sl.off_sel : (rows_direct ? sl.g_sel : sl.gt_sel), V, 2, -1, tag, to, s_to, s_sw, s_vt, sys))) return 1;
    K3S(4);
    // (5) ancestors of my output slots
    double u_sys = 0.0, ub_;
    if (sl.method != SMCMI_RESAMPLE_MULTINOMIAL) uniform_pair(seed, 0ull, (unsigned)n, rng_tag(P_RES, 0, 0), u_sys, ub_);
    auto ldcum = [&](long long j) { return load_f64_sc1(cum_rsrc, (unsigned)j * 8u); };
    const bool staged = sl.method != SMCMI_RESAMPLE_MULTINOMIAL && ncg <= 1024 && cap_w >= 2 * SEL_GCH;
    if (staged) sel_chunk_ends(ldcum, 0, Ng, ncg, s_ce);
    K3S(5);
    const long long slot = gid0 + (live ? i : (end > beg ? end - 1 : 0));
    const double ua = sel_threshold(sl.method, seed, slot, n, u_sys, Ng);
    const long long anc_i = (end > beg) ? sel_search_tile(ua, staged, ncg, s_ce, s_cw, s_r, ldcum, 0, Ng, cap_w) : 0;
    K3S(6);
    // (6) the ancestor's row becomes my particle (several handles: out of the rows of the handle that holds it - one buffer descriptor per
    // handle some lane of the wavefront reads from; a tile's ancestors are consecutive rows, so that is one handle, two at a boundary)
    double xx[DAm];
    xx[0] = 1.0;
#pragma unroll
    for (int q = 0; q < D; ++q) xx[q + 1] = 0.0;
    {
        if (live && sl.anc) sl.anc[i] = anc_i;
        double row[D + 4];
#pragma unroll
        for (int q = 0; q < D + 4; ++q) row[q] = 0.0;
        if (!sys) {
            if (live) {
#pragma unroll
                for (int q = 0; q < D + 4; ++q) row[q] = load_f64_sc1(cl_rsrc, (unsigned)(((long long)q * cl_n + anc_i) * 8));
            }
        } else {
            const int ar = live ? (int)(anc_i / sl.n_loc) : -1;
            const long long al = anc_i - (long long)ar * sl.n_loc;
            for (int pr = 0; pr < sl.world; ++pr) {
                if (!__any(ar == pr)) continue;                                   // (wave-uniform)
                const __amdgpu_buffer_rsrc_t pr_rsrc = rows_rsrc(reinterpret_cast<const double *>(sl.peers[pr] + sl.off_rows), (long long)(D + 4) * sl.n_loc * 8);
                if (ar == pr) {
#pragma unroll
                    for (int q = 0; q < D + 4; ++q) {
                        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(pr_rsrc, (int)(((long long)q * sl.n_loc + al) * 8), 0, /*sc0 sc1: system scope*/ 17);
                        row[q] = __hiloint2double((int)v.y, (int)v.x);
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < D + 4; ++q) { if (q < D) stx[q * T3 + tid] = row[q]; else sto[(1 + q - D) * T3 + tid] = row[q]; }
#pragma unroll
            for (int q = 0; q < D; ++q) xx[q + 1] = row[q] - shift[q];
        }
    }
    K3S(7);
    // (7) the moment row of my block (pair sums x̃_a x̃_b about the shift, all weights 1: 0 + x̃_a x̃_b is the accumulator k2_gather holds for a
    // block's only tile), published; (8) the totals of the resampled cloud's moments replace the correction's
    unsigned long long *my_gm = sl.g_gm + (long long)rowi * MGM * 2;
    cm_row_chunks<NPm, T3 / 64>(red, [&](auto C, double (&a)[CMW]) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        static_for<CMW>([&](auto Q) __attribute__((always_inline)) {
            constexpr int idx = c * CMW + decltype(Q)::value, q = decltype(Q)::value;
            double val = 0.0;
            if constexpr (idx < NPm) {
                constexpr int pa = cm_pair_a(DAm, idx + 2), pb = cm_pair_b(DAm, idx + 2);
                val = 0.0 + xx[pa] * xx[pb];
            }
            a[q] = live ? val : 0.0;
        });
    }, [&](int idx, double val) { gran_store(my_gm + idx * 2, val, tag); });
    if (tid >= NPm && tid < MGM) gran_store(my_gm + tid * 2, 0.0, tag);
    K3S(8);
    if (!(rows_two ? gather_totals<2>(sl.g_gm, V, MGM, -1, tag, to, s_to, s_tot + 2, s_vt)
                   : gather_totals(sys ? sl.mine + sl.off_gm : (rows_direct ? sl.g_gm : sl.gt_gm), V, MGM, -1, tag, to, s_to, s_tot + 2, s_vt, sys))) return 1;
    K3S(9);
    return 0;
#undef K3S
}

#ifdef SMCMI_K3_CH2
// The same for a worker of TWO chunks (k3_segment<D, true, RIDE, 2>; one handle): both chunks' particles and cum values are written before the first
// hand-over, both chunks' ancestors are found and fetched behind it, both moment rows go out before the second - two hand-overs as for one chunk.
// A chunk: stx [θ_1..θ_D][T3]; sto columns 1..4 = loglh | logprior | old_loglh | accept; wv = W̃ (the chunk in registers travels through the
// block's slice of Sel3Args::transit in k3_select_inside's layout, the parked one stays where it is: stx = its LDS columns, sto = column D - 1 on,
// wv = its column D + 5).  has = false: the worker's second chunk lies beyond its virtual shard - no rows, nothing posted.
struct Sel2Chunk { double *stx, *sto; const double *wv; int rowi, has; long long beg, end; };
struct Sel2Geo { int rowi, has; long long beg, end; };             // chunk c of a worker (block-uniform, left in LDS once per launch)
template <int D>
__device__ __attribute__((noinline)) int k3_select_two(const Sel3Args *selp, double *buf0, long long cl_n, int cl_R, long long Ng, int nchunks, int V, const Sel2Geo *geo, int cur, double *tr, double *st2,
                                                       unsigned tag, int n, unsigned long long seed, long long gid0, const unsigned long long *g_cm, unsigned long long *to, int *s_to,
                                                       double *s_tot, double *s_vt, double *s_sw, double *red, double *sc, const double *shift) {
    constexpr int NPm = Mut2Lds<D>::NP, MGM = pad2(NPm), DAm = D + 1, NPF = Mut2Lds<D>::NPF, MCM = pad2(NPF);
    const Sel3Args sl = *selp;
    const int tid = threadIdx.x;
    // [0]: the chunk that was in registers (chunk `cur`, now in the transit slice), [1]: the parked one
    const Sel2Geo g0 = geo[cur], g1 = geo[cur ^ 1];
    const Sel2Chunk ch[2] = {Sel2Chunk{tr + 5 * T3, tr, tr, g0.rowi, g0.has, g0.beg, g0.end},
                             Sel2Chunk{st2, st2 + (D - 1) * T3, st2 + (D + 5) * T3, g1.rowi, g1.has, g1.beg, g1.end}};
    const long long ci[2] = {ch[0].beg + tid, ch[1].beg + tid};
    double *s_w = sc, *s_cs = sc + 16, *s_scr = s_cs + nchunks, *s_off = s_scr + 256, *s_ce = sc + 16;
    long long *s_r = reinterpret_cast<long long *>(sc + 8);
    const int ncg = (int)((Ng + SEL_GCH - 1) / SEL_GCH);
    double *s_cw = s_ce + ncg;
    const int cap_w = (D + 2) * T3 - 16 - ncg;
    const __amdgpu_buffer_rsrc_t cl_rsrc = rows_rsrc(buf0, (long long)cl_R * cl_n * 8);
    const __amdgpu_buffer_rsrc_t cum_rsrc = rows_rsrc(sl.cum, Ng * 8);
    __syncthreads();
    // (1) the particles as stage n - 1 left them -> buffer 0
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (ch[q].has && ci[q] < ch[q].end) {
#pragma unroll
            for (int k = 0; k < D + 4; ++k) row_store(buf0 + (long long)k * cl_n + ci[q], k < D ? ch[q].stx[k * T3 + tid] : ch[q].sto[(1 + k - D) * T3 + tid], true);
        }
    }
    // (2) the chunk sums = entry 0 of every correction row -> chunk offsets
    for (int c = tid; c < nchunks; c += T3) {
        const unsigned long long *wd = g_cm + (long long)c * MCM * 2;
        gran_poll(wd, tag, to, s_to);
        gran_poll(wd + 1, tag, to, s_to);
        const unsigned long long w0 = __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w1 = __hip_atomic_load(wd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cs[c] = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
    }
    __syncthreads();
    if (*s_to) return 1;
    sel_chunk_offsets([&](int b) { return s_cs[b]; }, nchunks, s_scr, s_off);
    // (3) the cum values of both chunks (k2_scan's arithmetic on W̃)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!ch[q].has) continue;                                   // (block-uniform)
        const bool live = ci[q] < ch[q].end;
        double tt;
        const double incl = sel_tile_scan(live ? ch[q].wv[tid] : 0.0, s_w, &tt);
        if (live) row_store(sl.cum + ci[q], (s_off[ch[q].rowi] + incl) / s_tot[0], true);
    }
    // (4) hand-over: every store above is acknowledged before this block says so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (ch[q].has && tid < 2) gran_store(sl.g_sel + ((long long)ch[q].rowi * 2 + tid) * 2, 0.0, tag);
    if (!gather_totals(sl.gt_sel, V, 2, -1, tag, to, s_to, s_sw, s_vt, false)) return 1;
    // (5) ancestors of the output slots, (6) their rows become the particles
    double u_sys = 0.0, ub_;
    if (sl.method != SMCMI_RESAMPLE_MULTINOMIAL) uniform_pair(seed, 0ull, (unsigned)n, rng_tag(P_RES, 0, 0), u_sys, ub_);
    auto ldcum = [&](long long j) { return load_f64_sc1(cum_rsrc, (unsigned)j * 8u); };
    const bool staged = sl.method != SMCMI_RESAMPLE_MULTINOMIAL && ncg <= 1024 && cap_w >= 2 * SEL_GCH;
    if (staged) sel_chunk_ends(ldcum, 0, Ng, ncg, s_ce);
    double xx[2][DAm];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        xx[q][0] = 1.0;
#pragma unroll
        for (int a = 0; a < D; ++a) xx[q][a + 1] = 0.0;
        if (!ch[q].has) continue;
        const long long beg = ch[q].beg, end = ch[q].end, i = ci[q];
        const bool live = i < end;
        const long long slot = gid0 + (live ? i : (end > beg ? end - 1 : 0));
        const double ua = sel_threshold(sl.method, seed, slot, n, u_sys, Ng);
        __syncthreads();                                            // (the previous chunk's search is done with s_r / s_cw)
        const long long anc_i = (end > beg) ? sel_search_tile(ua, staged, ncg, s_ce, s_cw, s_r, ldcum, 0, Ng, cap_w) : 0;
        if (live) {
            if (sl.anc) sl.anc[i] = anc_i;
            double row[D + 4];
#pragma unroll
            for (int k = 0; k < D + 4; ++k) row[k] = load_f64_sc1(cl_rsrc, (unsigned)(((long long)k * cl_n + anc_i) * 8));
#pragma unroll
            for (int k = 0; k < D + 4; ++k) { if (k < D) ch[q].stx[k * T3 + tid] = row[k]; else ch[q].sto[(1 + k - D) * T3 + tid] = row[k]; }
#pragma unroll
            for (int k = 0; k < D; ++k) xx[q][k + 1] = row[k] - shift[k];
        }
    }
    // (7) the moment rows of both blocks, published; (8) the totals of the resampled cloud's moments replace the correction's
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!ch[q].has) continue;
        const bool live = ci[q] < ch[q].end;
        unsigned long long *my_gm = sl.g_gm + (long long)ch[q].rowi * MGM * 2;
        cm_row_chunks<NPm, T3 / 64>(red, [&](auto C, double (&a)[CMW]) __attribute__((always_inline)) {
            constexpr int c = decltype(C)::value;
            static_for<CMW>([&](auto Q) __attribute__((always_inline)) {
                constexpr int idx = c * CMW + decltype(Q)::value, qq = decltype(Q)::value;
                double val = 0.0;
                if constexpr (idx < NPm) {
                    constexpr int pa = cm_pair_a(DAm, idx + 2), pb = cm_pair_b(DAm, idx + 2);
                    val = 0.0 + xx[q][pa] * xx[q][pb];
                }
                a[qq] = live ? val : 0.0;
            });
        }, [&](int idx, double val) { gran_store(my_gm + idx * 2, val, tag); });
        if (tid >= NPm && tid < MGM) gran_store(my_gm + tid * 2, 0.0, tag);
    }
    if (!gather_totals(sl.gt_gm, V, MGM, -1, tag, to, s_to, s_tot + 2, s_vt, false)) return 1;
    return 0;
}
#endif

// ONE hand-over per stage (fixed schedules, RunParams::shift_lag; the reference's default: use_fixed_schedule = true, src/smc_main.jl:139,386-387):
// ϕ_{n+1} is the schedule's next entry and the energy shift of stage n + 1's incremental weights is the maximum the begin of stage n learnt
// (Begin2::e_seen), so stage n + 1's correction row needs nothing of stage n's mutation totals - a worker forms and publishes it right behind
// its mutation row, and the totals of both arrive together: the begin of stage n + 1 (acceptance rate -> c, the records) runs on the way to the
// decision instead of behind a hand-over of its own.  True when stage n + 1's correction rides stage n's mutation rows: every block evaluates
// it from the same values (po = Post2 of stage n); the stage must be sure to begin (begin2_wave's exits: ϕ = 1 reached, pause, capacity).
// A kernel of its own (k3_segment<D, α1, RIDE = true>, launched for fixed schedules under shift_lag on one handle): the order of the stage
// loop's three leading steps is then a compile-time matter in both instantiations - chosen at run time it cost every stage of an ADAPTIVE run
// 0.8 µs and the mixture variant 14 spilled registers.  Inside a riding launch a stage that does not ride never begins (the conditions above
// are begin2_wave's exits; Post2::e_seen is finite from the first begin on), so the riding order needs no second site for the correction.
__device__ inline bool k3_rides(const RunParams &rp, const Seg3Args &sa, const Post2 &po, int n, bool sys) {
    (void)sys;
    return rp.use_fixed_schedule && rp.shift_lag && n < sa.n_last && po.phi_n < 1.0 && !(rp.stop_stage > 0 && n >= rp.stop_stage) && n + 1 <= rp.max_stages;
}

// DRAW and BEGIN of the worker's stage loop (see there) as text: each instantiation of the kernel expands them at ONE place (the other is
// discarded by `if constexpr`), and - unlike lambdas, whose by-reference captures put a dozen loop variables into scratch - they cost nothing.
#define K3_DO_DRAW(ns)                                                                                                                          \
    do {                                                                                                                                        \
        if ((ns) <= sa.n_last) k3_draw_park<D, !ALPHA1>(z_park, ma.seed, pid_park, (unsigned)(ns), db0, ma.debug);                               \
        K3_STAMP(sa.prof, 7);                                                                                                                   \
    } while (0)
// po_p: Post2 of stage ns - 1; pair: stage ns's correction row is out as well (riding) - both tables' totals in one fetch, s_tot filled for the
// decision; the stage that ends a riding launch has no correction row: its begin takes the mutation totals alone.  ACT <- the begin's code, -1: timed out
#define K3_DO_BEGIN(ns, po_p, pair, ACT)                                                                                                         \
    do {                                                                                                                                        \
        const unsigned tag_p_ = sa.tag_base | (unsigned)((ns) - 1);                                                                             \
        bool ok_;                                                                                                                               \
        if (RIDE && (pair)) {                                                                                                                   \
            const unsigned tag_c_ = sa.tag_base | (unsigned)(ns);                                                                               \
            ok_ = rows_two ? gather_totals_pair<2>(sa.g_mut + K3_RPAR((ns) - 1), tag_p_, sa.g_cm + K3_RPAR(ns), MCM, tag_c_, g.V, sa.to, &s_to, s_totm, s_tot, s_vt)  \
                           : gather_totals_pair(rows_direct ? sa.g_mut + K3_RPAR((ns) - 1) : sa.gt_mut + K3_TPAR((ns) - 1), tag_p_,             \
                                                rows_direct ? sa.g_cm + K3_RPAR(ns) : sa.gt_cm + K3_TPAR(ns), MCM, tag_c_, g.V, sa.to, &s_to, s_totm, s_tot, s_vt, sys,  \
                                                (writer && sys) ? sa.vt_mut_out : nullptr);                                                     \
        } else {                                                                                                                                \
            ok_ = rows_two ? gather_totals<2>(sa.g_mut + K3_RPAR((ns) - 1), g.V, RMUT, RMAX_IDX, tag_p_, sa.to, &s_to, s_tot, s_vt)            \
                           : gather_totals(rows_direct ? sa.g_mut + K3_RPAR((ns) - 1) : sa.gt_mut + K3_TPAR((ns) - 1), g.V, RMUT, RMAX_IDX, tag_p_, sa.to, &s_to, s_tot, s_vt, sys,  \
                                           (writer && sys) ? sa.vt_mut_out : nullptr);                                                          \
        }                                                                                                                                       \
        if (!ok_) { timed_out = true; ACT = -1; break; }                                                                                        \
        K3_STAMP(sa.prof, 8);                                                                                                                   \
        K3_WALL(sa.gprof, PROF2_WORK + 4 * blockIdx.x + 3);                                                                                            \
        ACT = begin_stage((ns), (po_p), (RIDE && (pair)) ? s_totm : s_tot);                                                                     \
        constexpr int NWB_ = sizeof(Begin2) / sizeof(double);                                                                                   \
        if (ACT == 0 && writer && tid < NWB_) reinterpret_cast<double *>(&ctl->bg)[tid] = reinterpret_cast<const double *>(&s_a.bg)[tid];       \
        K3_STAMP(sa.prof, 9);                                                                                                                   \
    } while (0)
#define K3_RPAR(stage) ((RIDE && ((stage) & 1)) ? (long long)k3_copy_words(g.Vl * g.nb2) : 0)
#define K3_TPAR(stage) ((RIDE && ((stage) & 1)) ? (sys ? (long long)MB_SEG_COPY_WORDS : (long long)k3_copy_words(g.Vl * g.nb2)) : 0)
// TWO CHUNKS PER WORKER (k3_segment<D, true, RIDE, 2>) is compiled from this same kernel body in translation units of its own, which define
// SMCMI_K3_CH2 (inst3.hip with -DSMCMI_INST3_C=2).  Everywhere else the few places where the variants differ expand to the one-chunk text of
// round 5 - not to `if constexpr (CH == 1)` equivalents: the segment kernel's code generation is that touchy (the same kernel with the
// variants' differences as dead template branches ran the headline's stage 0.5 µs - 1.6 % - slower: profiles/r06_codegen_ab.txt).
#ifdef SMCMI_K3_CH2
// CH = 2: exchange the chunk in registers with the parked one (thread-private LDS slots: no barrier)
#define K3_SWAP_CHUNKS()                                                                                                                        \
    do {                                                                                                                                        \
        if constexpr (CH > 1) {                                                                                                                 \
            double *s2_ = st2 + tid;                                                                                                            \
            _Pragma("unroll") for (int k_ = 0; k_ < D; ++k_) { const double t_ = s2_[k_ * T3]; s2_[k_ * T3] = x[k_]; x[k_] = t_; }              \
            { double t_; t_ = s2_[D * T3]; s2_[D * T3] = like; like = t_; t_ = s2_[(D + 1) * T3]; s2_[(D + 1) * T3] = lprior; lprior = t_;      \
              t_ = s2_[(D + 2) * T3]; s2_[(D + 2) * T3] = like_prev; like_prev = t_; t_ = s2_[(D + 3) * T3]; s2_[(D + 3) * T3] = acc_val; acc_val = t_; \
              t_ = s2_[(D + 4) * T3]; s2_[(D + 4) * T3] = Wt; Wt = t_; t_ = s2_[(D + 5) * T3]; s2_[(D + 5) * T3] = v; v = t_; }                \
            cur ^= 1;                                                                                                                           \
            beg = beg_c[cur]; end = end_c[cur]; i = beg + tid; live = i < end; has = has_c[cur]; rowi = rowi_c[cur];                            \
            pid = (unsigned long long)(ma.gid0 + i);                                                                                            \
        }                                                                                                                                       \
    } while (0)
#define K3_HAS has                                                       /* this chunk exists (the last worker of a shard with an odd block count has one) */
#define K3_CHUNKS_BEGIN for (int it = 0; it < CH; ++it) {                /* the chunk in registers, then the parked one */
#define K3_CHUNKS_END if (it + 1 < CH) K3_SWAP_CHUNKS(); }
#define K3_PARKED_DRAWS (cur == 0)                                       /* only chunk 0's first proposal is drawn ahead */
#define K3_V_RESET() v = 0.0
#else
#define K3_HAS true
#define K3_CHUNKS_BEGIN {
#define K3_CHUNKS_END }
#define K3_PARKED_DRAWS true
#define K3_V_RESET() (void)0
#endif

// grid = W + g.Vl blocks of T3 threads, every one resident (W = g.Vl * g.nb2 workers, then one gatherer per local virtual shard; W blocks
// where the workers take the rows themselves: one handle with one or two blocks per virtual shard).
// Worker b owns block (b / Vl) of local virtual shard (b % Vl): with the hardware's round-robin of consecutive blocks over the 8 XCDs
// a virtual shard's workers and (W a multiple of 8) its gatherer share an XCD - placement is speed only, never correctness.
// Every block - gatherers included - derives the stage's decisions itself from the V shard totals (decide2, post2, begin2_wave: same
// inputs, same code, same result everywhere, as in engine 2's kernels), the workers also the proposal; worker 0 records them.
// CH = 2 (α = 1 only; one handle of 126 977 .. 253 952 particles with a cheap likelihood, run2.hpp seg3_ready): a worker owns TWO consecutive
// 512-particle chunks of its virtual shard - one in registers, one parked in LDS ((D + 6) columns behind the parking area), exchanged between
// the per-particle phases; it publishes two rows per hand-over and pays the serial phases and the hand-overs once.  A stage that must resample
// does so in place (k3_select_two).
template <int D, bool ALPHA1, bool RIDE, int CH = 1, bool SYS = false>
__global__ void __launch_bounds__(T3, SMCMI_K3_WAVES) k3_segment(CloudPtrs cl, DevState *st, Ctl2 *ctl, const ModelDev *md, Geo2 g, Mut2Args ma, Seg3Args sa, int nb, int nf) {
    constexpr int NPF = Mut2Lds<D>::NPF, MCM = pad2(NPF);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ RecA3 s_a;
    __shared__ RecB3<D> s_b[2];                                 // [n & 1]: Post2 of stage n lives on as "Post2 of n - 1" during stage n + 1
    __shared__ double s_vt[V2_MAXV * RMUT * 4], s_tot[pad2(NPF) > RMUT ? pad2(NPF) : RMUT], s_sw[64];
    __shared__ double s_totm[RIDE ? RMUT : 2];                  // riding stages: the mutation totals arrive WITH the correction totals (gather_totals_pair)
    static_assert(V2_MAXV * RMUT * 4 >= V2_MAXV * (72 + RMUT), "gather_totals_pair stages both tables in s_vt");
    __shared__ double red[(T3 / 64) * (cm_row_ld(NPF) > 64 ? cm_row_ld(NPF) : 64)];
    __shared__ int s_act, s_to, s_fail;
    __shared__ double s_cfac;
#ifdef SMCMI_K3_CH2
    __shared__ Sel2Geo s_ch2[2];                                // the worker's two chunks, for an in-place selection (k3_select_two)
#endif
    __shared__ RunParams s_rp;                                  // (by value in registers it costs ~40 SGPRs for the whole launch)
    __shared__ double mixbuf[ALPHA1 ? 1 : MixDense<D>::DOUBLES + D * D];
    __shared__ int mixpos[ALPHA1 ? 1 : D];
    __shared__ double mixzt[ALPHA1 ? 1 : T3 * D];
    Mut2Lds<D> L(sm);
    const int tid = threadIdx.x;
#ifdef SMCMI_K3_CH2
    const int W = g.Vl * ((g.nb2 + CH - 1) / CH);               // (workers per virtual shard: every one owns CH blocks of it)
#else
    static_assert(CH == 1, "two chunks per worker: a translation unit that defines SMCMI_K3_CH2 (inst3.hip -DSMCMI_INST3_C=2)");
    const int W = g.Vl * g.nb2;
#endif
    const bool worker = (int)blockIdx.x < W, writer = blockIdx.x == 0;
    if (tid == 0) { s_rp = st->rp; s_to = 0; }
    if (tid < nf) L.fi[tid] = md->free_inds[tid];
    __syncthreads();
    const RunParams &rp = s_rp;
    int n = sa.n_first;
    int rs0 = 0;                                                // the entered stage resampled (enter_mut)
    if (!sa.enter_mut) {
        // ---- the first stage's begin: as K1's prologue, by every block from the rows the previous launch left (block 0 records it)
        const int act = begin2_block<T3>(n, st, ctl, sa.mrows, 1, sa.sched, ma.rec, &s_b[(n - 1) & 1].po, &s_a.bg, s_vt, s_tot, s_sw, &s_act);
        if (act != 0) { k3_leave_note(sa, ctl); return; }       // nothing was touched: the cloud in memory is current
        if (tid == 0) {
            const double a = s_a.bg.accept, tg = rp.target;
            s_cfac = 0.95 + 0.10 * exp(16.0 * (a - tg)) / (1.0 + exp(16.0 * (a - tg)));
            s_a.bg.cfac = s_cfac;
        }
        __syncthreads();
    } else {
        // ---- entering at the mutation of stage n: K2's prologue (k2_prologue), by every block
        constexpr int NWB = sizeof(Begin2) / sizeof(double), NWP = sizeof(Post2) / sizeof(double), NPm = Mut2Lds<D>::NP;
        if (tid < NWB) reinterpret_cast<double *>(&s_a.bg)[tid] = reinterpret_cast<const double *>(&ctl->bg)[tid];
        if (tid < NWP) reinterpret_cast<double *>(&s_b[(n - 1) & 1].po)[tid] = reinterpret_cast<const double *>(&ctl->ps[(n - 1) & 1])[tid];
        if (ma.cmrows.mb) {                                       // mailbox: a launch that will not run must not wait for rows nobody posts (begin2_block)
            __syncthreads();
            if (s_a.bg.stage != n || !s_a.bg.final || s_b[(n - 1) & 1].po.stage != n - 1) { k3_leave_note(sa, ctl); return; }
        }
        reduce_rows<MCM, 1, T3>(ma.cmrows, s_vt, s_tot);          // (its barriers also publish the LDS copies above)
        if (s_a.bg.stage != n || !s_a.bg.final || s_b[(n - 1) & 1].po.stage != n - 1) { k3_leave_note(sa, ctl); return; }       // the state this launch was enqueued for is not there: no-op
        double ess;
        const int dec = decide2(s_a.bg, rp.threshold, rp.phi_rtol, s_tot[0], s_tot[1], &ess);
        if (dec == 4 || dec < 0 || (dec == 1 && !ma.sel_enqueued)) {                             // the stalls K2 reports (the host resumes the stage through the launches)
            if (writer && tid == 0) {
                if (dec < 0) { ma.rec.phi[n - 1] = s_a.bg.phi_n; ma.rec.ess[n - 1] = ess; ctl->status.err = dec; }
                ctl->status.stage = n;
                ctl->status.code = dec == 4 ? 4 : (dec < 0 ? 9 : 3);
            }
            k3_leave_note(sa, ctl);
            return;
        }
        rs0 = dec == 1 ? 1 : 0;
        if (sa.clear_status && writer && tid == 0) { ctl->status.code = 0; ctl->status.stage = 0; }        // (instead of a fill launch by the host)
        if (rs0) reduce_rows<pad2(NPm), 1, T3>(ma.gmrows, s_vt, s_tot + 2);          // moments of the resampled cloud (k2_gather's rows) replace the correction's
    }
    const double inv_pre = INV_FACTORIAL[tid & 31];
    // Stage nb's begin from the totals of stage nb - 1's mutation rows (in s_tot), by every block alike: 0 go on, 7 segment complete,
    // else begin2_wave's code (finished / paused / error / no usable prediction: the writer has set the status).  Ends with a barrier.
    auto begin_stage = [&](int nb_, const Post2 &po_n, const double *tm) -> int {
        if (tid == 0) s_act = 7;
        if (tid < 64) {
            const int jj = po_n.j - 1 + tid;                    // the window of the proposed schedule the begin walks
            s_sw[tid] = (!rp.use_fixed_schedule && jj >= 0 && jj < rp.n_phi) ? sa.sched[jj] : 2.0;
        }
        if (nb_ <= sa.n_last) {
            if (tid < 64) {
                const int act = begin2_wave(nb_, po_n, rp, tm, tm[RMAX_IDX], true, 1, sa.sched, s_sw, &s_a.bg, &st->sol[0], writer, ma.rec,
                                            &ctl->status, inv_pre);
                if (tid == 0) s_act = act;
            } else if (tid == 64) {
                // the step-size multiplier of stage nb (smc_main.jl:453-455) from the acceptance rate begin2_wave folds
                const double a = tm[EACC] / (double)rp.n_parts, tg = rp.target;
                s_cfac = 0.95 + 0.10 * exp(16.0 * (a - tg)) / (1.0 + exp(16.0 * (a - tg)));
            }
        }
        __syncthreads();
        if (tid == 0 && s_act == 0) s_a.bg.cfac = s_cfac;
        __syncthreads();
        return s_act;
    };
    // (SYS: several handles, Seg3Args::peers - a compile-time matter: with the two kinds of hand-over behind run-time branches the one-handle
    // kernel carried 30 more spilled scalar registers and ran the headline's stage 0.5 µs slower)
    constexpr bool sys = SYS;
    auto post_total = [&](unsigned long long *mine, long long off, long long w, double val, unsigned tg) {
        if (sys) { for (int pr = 0; pr < sa.world; ++pr) gran_store_sys(sa.peers[pr] + off + w, val, tg); }
        else gran_store(mine + w, val, tg);
    };
    if (!worker) {
        // ================================================================ GATHERER of local virtual shard vg
        const int vg = (int)blockIdx.x - W;
        // (the gatherer stages a shard's rows - at most 2 GRP of them - in the dynamic LDS: launch2.hpp launch_k3_seg sizes it, k3_gather_lds_bytes)
        double *g_stage = sm;                                    // (a gatherer uses none of the workers' dynamic LDS: model constants, proposal, parked draws)
        // (one hand-over per stage, see the workers: a stage whose correction rows ride the mutation rows in front of it has them swept and posted
        // BEFORE this block takes the mutation totals and runs the begin - the workers wait for the correction totals, nothing else)
        bool cm_posted = false;
        auto sweep_cm = [&](int ns) __attribute__((always_inline)) -> bool {
            const unsigned tg = sa.tag_base | (unsigned)ns;
            const long long rp_ = (RIDE && (ns & 1)) ? (long long)k3_copy_words(g.Vl * g.nb2) : 0, tp_ = K3_TPAR(ns);
            return gather_vshard<T3>(sa.g_cm + rp_ + (long long)vg * g.nb2 * MCM * 2, g.nb2, MCM, -1, tg, sa.to, &s_to,
                                     [&](int idx, double val) { post_total(sa.gt_cm + tp_, sa.off_cm + tp_, ((long long)(g.v0 + vg) * MCM + idx) * 2, val, tg); }, g_stage,
                                     (sa.gprof && ns == sa.prof_stage) ? sa.gprof + PROF2_GPOLL + 4 * vg : nullptr);
        };
        for (;; ++n) {
            const unsigned tag = sa.tag_base | (unsigned)n;
            const long long rpar = (RIDE && (n & 1)) ? (long long)k3_copy_words(g.Vl * g.nb2) : 0, tpar = K3_TPAR(n);      // stage n's copy of the tables (riding launches)
            const bool entered = sa.enter_mut && n == sa.n_first;        // (its correction totals and decision are there: the entry block above)
            if (!entered) {
                K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 0);
                if (!cm_posted && !sweep_cm(n)) break;
                K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 1);
                K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 2);
                // the decision every worker takes from the V totals (a stage that does not go on mutates nothing: no rows to wait for)
                if (!gather_totals(sa.gt_cm + tpar, g.V, MCM, -1, tag, sa.to, &s_to, s_tot, s_vt, sys)) break;
            }
            const double ess = s_tot[0] * s_tot[0] / s_tot[1];
            int rs_g = entered ? rs0 : 0;
            if (!entered) {
                double e2;
                const int dec = decide2(s_a.bg, rp.threshold, rp.phi_rtol, s_tot[0], s_tot[1], &e2);
                if (__builtin_expect(dec == 1 && sa.sel != nullptr, 0)) {
                    // SELECTION inside the segment (the workers' side is below): "everything is written", then the moment rows of the resampled cloud
                    constexpr int MGM = pad2(Mut2Lds<D>::NP);
                    const Sel3Args sl = *sa.sel;
                    // (several handles: the totals go into every handle's tables, Sel3Args)
                    auto post_sel = [&](unsigned long long *mine_t, long long off, long long w, double val) {
                        if (sl.peers) { for (int pr = 0; pr < sl.world; ++pr) gran_store_sys(sl.peers[pr] + off + w, val, tag); }
                        else gran_store(mine_t + w, val, tag);
                    };
                    const bool ssys = sl.peers != nullptr;
                    if (!gather_vshard<T3>(sl.g_sel + (long long)vg * g.nb2 * 2 * 2, g.nb2, 2, -1, tag, sa.to, &s_to,
                                           [&](int idx, double val) { post_sel(sl.gt_sel, sl.off_sel, ((long long)(g.v0 + vg) * 2 + idx) * 2, val); }, g_stage)) break;
                    if (!gather_totals(ssys ? sl.mine + sl.off_sel : sl.gt_sel, g.V, 2, -1, tag, sa.to, &s_to, s_sw, s_vt, ssys)) break;
                    if (!gather_vshard<T3>(sl.g_gm + (long long)vg * g.nb2 * MGM * 2, g.nb2, MGM, -1, tag, sa.to, &s_to,
                                           [&](int idx, double val) { post_sel(sl.gt_gm, sl.off_gm, ((long long)(g.v0 + vg) * MGM + idx) * 2, val); }, g_stage)) break;
                    if (!gather_totals(ssys ? sl.mine + sl.off_gm : sl.gt_gm, g.V, MGM, -1, tag, sa.to, &s_to, s_tot + 2, s_vt, ssys)) break;
                    rs_g = 1;
                } else if (dec != 0) break;
            }
            if (tid == 0) post2(n, s_a.bg, s_b[(n - 1) & 1].po, rp, s_tot[0], s_tot[1], ess, rs_g, &s_b[n & 1].po);
            __syncthreads();
            K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 3);
            if (!gather_vshard<T3>(sa.g_mut + rpar + (long long)vg * g.nb2 * RMUT * 2, g.nb2, RMUT, RMAX_IDX, tag, sa.to, &s_to,
                                   [&](int idx, double val) { post_total(sa.gt_mut + tpar, sa.off_mut + tpar, ((long long)(g.v0 + vg) * RMUT + idx) * 2, val, tag); }, g_stage,
                                   (sa.gprof && n == sa.prof_stage) ? sa.gprof + PROF2_GPOLL + 4 * vg + 2 : nullptr)) break;
            K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 4);
            cm_posted = false;
            if (RIDE && k3_rides(rp, sa, s_b[n & 1].po, n, sys)) { if (!sweep_cm(n + 1)) break; cm_posted = true; }
            K3_WALL(sa.gprof, PROF2_GATH + 6 * vg + 5);
            if (!gather_totals(sa.gt_mut + tpar, g.V, RMUT, RMAX_IDX, tag, sa.to, &s_to, s_tot, s_vt, sys)) break;
            if (begin_stage(n + 1, s_b[n & 1].po, s_tot) != 0) break;
        }
        return;
    }
    // ==================================================================== WORKER
    // the first proposal's random numbers of the NEXT stage, drawn while the block waits for that stage's begin (they depend on (seed,
    // particle, stage) only) and parked here, slot-major: z_park[slot * T3 + tid], slots = MH uniform, mixture uniform, D normals
    double *z_park = sm + k3_park_offset(D);
#ifndef SMCMI_K3_CH2
    const int vl = (int)blockIdx.x % g.Vl, r = (int)blockIdx.x / g.Vl, rowi = vl * g.nb2 + r;       // row index = engine 2's block index
    const double pw = rp.pw, logp_old = rp.logp_old, nrm_N = ma.n_parts;
    const bool hist = rp.store_history && sa.hist_w != nullptr;
    long long beg, end;
    vchunk(g, vl, r, T3, beg, end);
    const long long i = beg + tid;
    const bool live = i < end;
    const long long il = live ? i : (end > beg ? end - 1 : 0);
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    const unsigned long long pid_park = pid;                    // (the two-chunk variant parks chunk 0's draws: stage3.hpp under SMCMI_K3_CH2)
    double x[D], like, lprior, like_prev, Wt, acc_val;
    // (entered at the mutation of a stage that resampled: the gathered cloud is in buffer 1 - k2_gather - as K2 reads it)
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = col(cl, rs0, k)[il];
    like = col(cl, rs0, D)[il]; lprior = col(cl, rs0, D + 1)[il]; like_prev = col(cl, rs0, D + 2)[il];
    acc_val = col(cl, 0, D + 3)[il]; Wt = col(cl, 0, D + 4)[il];
    double v_entered = sa.enter_mut ? ma.wt[il] : 0.0;          // the unnormalised weight K1 left for the entered stage
    if (!live) {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = 0.0;
        like = lprior = like_prev = 0.0;
    }
#else
    const int vl = (int)blockIdx.x % g.Vl, wr = (int)blockIdx.x / g.Vl;
    const double pw = rp.pw, logp_old = rp.logp_old, nrm_N = ma.n_parts;
    const bool hist = rp.store_history && sa.hist_w != nullptr;
    // chunk c of this worker = block (wr CH + c) of the virtual shard (row index = engine 2's block index); a chunk beyond the shard's last
    // block holds nothing and publishes nothing.  (CH = 1: nothing below ever changes `i`, `live`, `rowi`, `pid` - constants to the compiler)
    long long beg_c[CH], end_c[CH];
    int rowi_c[CH];
    bool has_c[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int rc = wr * CH + c;
        has_c[c] = CH == 1 || rc < g.nb2;                  // (CH = 1: every worker has its block - a constant, no predicate around the row stores)
        rowi_c[c] = vl * g.nb2 + (has_c[c] ? rc : g.nb2 - 1);
        vchunk(g, vl, rc, T3, beg_c[c], end_c[c]);
        if (!has_c[c]) beg_c[c] = end_c[c];
    }
    int cur = 0;                                                // the chunk in registers
    if (tid < CH) s_ch2[tid] = Sel2Geo{rowi_c[tid ? CH - 1 : 0], has_c[tid ? CH - 1 : 0] ? 1 : 0, beg_c[tid ? CH - 1 : 0], end_c[tid ? CH - 1 : 0]};     // (read behind the stage loop's barriers)
    long long beg = beg_c[0], end = end_c[0], i = beg + tid;
    bool live = i < end, has = has_c[0];
    int rowi = rowi_c[0];
    unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    const unsigned long long pid_park = pid;                    // (the draws parked ahead are chunk 0's)
    double *st2 = z_park + (D + 2) * T3;                        // CH = 2: the parked chunk [θ_1..θ_D | loglh | logprior | old_loglh | accept | W | W̃][T3]
    double x[D], like, lprior, like_prev, Wt, acc_val;
    double v_entered;                                           // the unnormalised weight K1 left for the entered stage
    // (entered at the mutation of a stage that resampled: the gathered cloud is in buffer 1 - k2_gather - as K2 reads it)
    {
        const long long il = live ? i : (end > beg ? end - 1 : 0);
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = col(cl, rs0, k)[il];
        like = col(cl, rs0, D)[il]; lprior = col(cl, rs0, D + 1)[il]; like_prev = col(cl, rs0, D + 2)[il];
        acc_val = col(cl, 0, D + 3)[il]; Wt = col(cl, 0, D + 4)[il];
        v_entered = sa.enter_mut ? ma.wt[il] : 0.0;
        if (!live) {
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = 0.0;
            like = lprior = like_prev = 0.0;
        }
    }
    if constexpr (CH > 1) {
        const long long i1 = beg_c[1] + tid;
        const bool live1 = i1 < end_c[1];
        const long long il = live1 ? i1 : (end_c[1] > beg_c[1] ? end_c[1] - 1 : 0);
        double *s2 = st2 + tid;
#pragma unroll
        for (int k = 0; k < D + 3; ++k) s2[k * T3] = live1 ? col(cl, rs0, k)[il] : 0.0;
        s2[(D + 3) * T3] = col(cl, 0, D + 3)[il]; s2[(D + 4) * T3] = col(cl, 0, D + 4)[il];
        s2[(D + 5) * T3] = sa.enter_mut ? ma.wt[il] : 0.0;
    }
#endif
    for (int k = tid; k < D; k += T3) {
        L.m_lo[k] = md->lo[k]; L.m_hi[k] = md->hi[k]; L.m_a[k] = md->prior_a[k]; L.m_b[k] = md->prior_b[k]; L.m_k[k] = md->prior_k[k];
        L.m_fix[k] = md->fixed[k]; L.m_fam[k] = md->prior_family[k];
    }
    for (int k = tid; k < 2 * LIK_PAR_MAX; k += T3) L.l_par[k] = md->lik[k / LIK_PAR_MAX].par[k % LIK_PAR_MAX];
    ModelView mv{D, L.m_fix, L.m_fam, L.m_lo, L.m_hi, L.m_a, L.m_b, L.m_k};
    LikView lv[2];
    k2_stage_lik<T3>(ma.lik[0], ma.lik[1], L.l_par, L.l_dat, lv);     // once per segment (the mutation rows' scratch is `red`, not this area)
    const int db0 = nb == 1 ? nf : (nf + nb - 1) / nb;          // entries of the first random block
    k3_draw_park<D, !ALPHA1>(z_park, ma.seed, pid_park, (unsigned)n, db0, ma.debug);      // (later stages: under the wait for their begin)
    __syncthreads();
    int done = 0;
    bool timed_out = false;
    // one handle whose virtual shards are ONE block each (clouds of up to 4 096 particles): a shard's total is its only row (the canonical
    // sum of one row adds zeros to it), so every worker takes the V rows themselves - one hop per hand-over instead of two; no gatherer is
    // launched for such a cloud (launch2.hpp launch_k3_seg)
    const bool rows_direct = g.nb2 == 1 && !sys;
    const bool rows_two = g.nb2 == 2 && !sys;                    // ... two blocks each (up to 8 192 particles: the reference's default 5 000): gather_totals<2>
    // rides: this stage's correction row was formed in front of its begin, behind the previous stage's mutation row (k3_rides)
    bool rides = false;
#ifdef SMCMI_K3_CH2
    double v = v_entered;                                       // the particle's unnormalised weight W̃ of the stage: travels with its chunk
#endif
    for (;; ++n) {
        K3_STAMP(sa.prof, 1);
        RecB3<D> &B = s_b[n & 1];
        const Post2 &po = s_b[(n - 1) & 1].po;                  // stage n - 1 as completed
        const unsigned tag = sa.tag_base | (unsigned)n;
        // (stage n's copy of the tables: formed where they are used - the stage loop has no scalar registers to carry addresses across its phases)
        // (K3_RPAR / K3_TPAR: only a riding launch needs the second copy - Seg3Args: with two hand-overs per stage nobody is ever a table ahead)
        // the proposal arrays of THIS stage
        L.Lraw = B.pr.Lraw; L.logdet_s = B.pr.logdet; L.mub_raw = B.pr.mub; L.sdd_raw = B.pr.sdd; L.sdn_raw = B.pr.sdn;
        L.ball_raw = B.pr.ball; L.bptr_s = B.pr.bptr; L.loff_s = B.pr.loff;
        const bool first = n == sa.n_first;                     // (its begin ran in front of the loop, its draws are parked)
        const bool entered = sa.enter_mut && first;             // this stage's correction (and selection) ran as launches: totals in s_tot
        int rs = entered ? rs0 : 0;
#ifndef SMCMI_K3_CH2
        double v = entered ? v_entered : 0.0;                   // the particle's unnormalised weight W̃ of stage n (the entered stage: what K1 left)
#endif
        // Three steps lead up to a stage's correction totals (ONE site of code each per instantiation: the stage loop has neither registers nor
        // instruction cache for a second copy):
        //   CORR   correction at ϕ_n (src/smc_main.jl:401-420) + moments: one row per block, published
        //   DRAW   the first proposal's random numbers of the stage (functions of (seed, particle, stage) only), under whatever hand-over is pending
        //   BEGIN  the V shard totals of the previous stage's mutation rows -> the stage's begin (smc_main.jl:378-396, helpers.jl:9-56)
        // RIDE = false (adaptive schedules, several handles): CORR here, DRAW and BEGIN of the NEXT stage at the end of the loop body - the begin
        // decides ϕ_n.  RIDE = true (k3_rides): CORR, DRAW, BEGIN here, in that order - ϕ_n and the energy shift are known beforehand, both rows of
        // a block are out before it waits for anything.  (The first stage of a launch has its begin and its draws from the launch's prologue.)
        int act = 0;
        if (!entered && !(RIDE && !first && !rides)) {          // (riding: the stage that ends the launch forms no row - its begin, next, says so)
            // (riding: what begin2_wave will put into Begin2 for a fixed schedule under shift_lag - the same values, before the begin has run)
            const double phi = rides ? (n <= rp.n_phi ? sa.sched[n - 1] : 1.0) : s_a.bg.phi_n, phi_prev = rides ? po.phi_n : s_a.bg.phi_prev;
            const double esh = pw == 0.0 ? (rides ? po.e_seen - (rp.shift_lag == n ? 1e6 : 0.0) : s_a.bg.e_shift) : 0.0;
            K3_CHUNKS_BEGIN                                     // (two chunks: the one in registers, then the parked one - which stays in registers for the MH step)
            unsigned long long *my_cm = sa.g_cm + K3_RPAR(n) + (long long)rowi * MCM * 2;
            K3_V_RESET();
            if constexpr (ALPHA1) {
                // (one particle per thread: the row's sums are formed where the butterflies need them - no accumulator array alive)
                double xx[D + 1];
#pragma unroll
                for (int a = 0; a <= D; ++a) xx[a] = 0.0;
                if (live) {
                    double inc;
                    v = k2_cm_weight<D>([&](int a) { return x[a]; }, po.shift, like, like_prev, Wt, esh, phi, phi_prev, pw, logp_old, xx, &inc);
                    if (hist) {
                        const double unshift = exp((phi - phi_prev) * esh);
                        sa.hist_w[(long long)(n - 1) * sa.hist_ld + i] = inc * unshift;
                    }
                }
                k2_cm_row_one<D, T3 / 64>(v, xx, live, red, [&](int idx, double val) { if (K3_HAS) gran_store(my_cm + idx * 2, val, tag); });
            } else {
                // (the mixture kernel: the same sums, same bits, through the accumulator form - its register allocation takes that better:
                // 115 against 164 scratch reloads in the stage loop, 48.7 against 59.8 µs per stage)
                constexpr int NCH = (NPF + 63) / 64;
                double acc[NCH * 64];
#pragma unroll
                for (int q = 0; q < NCH * 64; ++q) acc[q] = 0.0;
                if (live) {
                    double inc;
                    v = k2_cm_particle<D>(acc, [&](int a) { return x[a]; }, po.shift, like, like_prev, Wt, esh, phi, phi_prev, pw, logp_old, &inc);
                    if (hist) {
                        const double unshift = exp((phi - phi_prev) * esh);
                        sa.hist_w[(long long)(n - 1) * sa.hist_ld + i] = inc * unshift;
                    }
                }
                k2_cm_row_f<D>(acc, red, [&](int idx, double val) { if (K3_HAS) gran_store(my_cm + idx * 2, val, tag); });
            }
            if (K3_HAS && tid >= NPF && tid < MCM) gran_store(my_cm + tid * 2, 0.0, tag);         // (the pad columns of the even row width)
            K3_CHUNKS_END
        }
        K3_STAMP(sa.prof, 2);
        K3_WALL(sa.gprof, PROF2_WORK + 4 * blockIdx.x + 0);
        if constexpr (RIDE) {
            if (!first) {
                K3_DO_DRAW(n);
                K3_DO_BEGIN(n, po, rides, act);
                if (act == 0 && !rides) {                       // (cannot happen, see k3_rides: this stage's correction row was due in front of its begin)
                    if (writer && tid == 0) { ctl->status.err = SMCMI_ERR_STATE; ctl->status.stage = n; ctl->status.code = 9; __hip_atomic_store(sa.to, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    act = -1;
                }
            }
        }
        if (act != 0) break;
        const double e_center = s_a.bg.e_center;
        const int jx_pre = (tid >= 64 && tid < 128) ? shuffle_partner(ma.seed, (unsigned)n, tid - 64, nf) : 0;     // (before the totals exist)
        // ---- the V shard totals -> decision (smc_main.jl:427-455) -> proposal (smc_main.jl:457-465, helpers.jl:215-260, mutation.jl:81)
        if (!entered && !(RIDE && rides) && !(rows_two ? gather_totals<2>(sa.g_cm + K3_RPAR(n), g.V, MCM, -1, tag, sa.to, &s_to, s_tot, s_vt)
                                   : gather_totals(rows_direct ? sa.g_cm + K3_RPAR(n) : sa.gt_cm + K3_TPAR(n), g.V, MCM, -1, tag, sa.to, &s_to, s_tot, s_vt, sys))) { timed_out = true; break; }
        K3_STAMP(sa.prof, 3);
        K3_WALL(sa.gprof, PROF2_WORK + 4 * blockIdx.x + 1);
        double ess = s_tot[0] * s_tot[0] / s_tot[1];
        int dec = entered ? 0 : decide2(s_a.bg, rp.threshold, rp.phi_rtol, s_tot[0], s_tot[1], &ess);
#ifndef SMCMI_K3_CH2
        if (__builtin_expect(dec == 1 && sa.sel != nullptr, 0)) {
            // ================= SELECTION inside the segment (k3_select_inside above): the particle goes through LDS - a call that took it in registers
            // would cost the stage loop 26 registers and 38 spills for a path one stage in twenty takes.  (A kernel whose static arrays leave no room
            // for the D + 5 columns - mixture proposals beyond n_para 7 - parks it in the block's slice of a scratch buffer in device memory instead:
            // the same thread writes and reads back every word, 60 KB per block that stay in the die's L2)
            double *sto = z_park + (D + 2) * T3;
            if constexpr (k3_sel_cols(D, ALPHA1) == 0) sto = sa.sel->transit + (long long)blockIdx.x * (D + 5) * T3;
            double *stx = sto + 5 * T3;
            __syncthreads();
            sto[tid] = v;
#pragma unroll
            for (int k = 0; k < D; ++k) stx[k * T3 + tid] = x[k];
            sto[T3 + tid] = like; sto[2 * T3 + tid] = lprior; sto[3 * T3 + tid] = like_prev; sto[4 * T3 + tid] = acc_val;
            const int bad = k3_select_inside<D>(sa.sel, cl.buf[0], cl.n, cl.R, g.N, g.V * g.nb1, g.V, rowi, i, beg, end, tag, n, ma.seed, ma.gid0, sa.g_cm + K3_RPAR(n), sa.to, &s_to, s_tot, s_vt, s_sw,
                                                red, z_park, stx, sto, po.shift, (sa.prof && writer && n == sa.prof_stage) ? sa.gprof + PROF2_SEL : nullptr, rows_direct, rows_two);
            if (bad) { timed_out = true; break; }
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = stx[k * T3 + tid];
            like = sto[T3 + tid]; lprior = sto[2 * T3 + tid]; like_prev = sto[3 * T3 + tid]; acc_val = sto[4 * T3 + tid];
            // this stage's draws again (the parking area was the selection's scratch; they are functions of (seed, particle, stage))
            k3_draw_park<D, !ALPHA1>(z_park, ma.seed, pid, (unsigned)n, db0, ma.debug);
            __syncthreads();
            rs = 1; dec = 0;
        }
#else
        if (__builtin_expect(dec == 1 && sa.sel != nullptr, 0)) {
            // ================= SELECTION inside a two-chunk segment (k3_select_two): the chunk in registers travels through the block's slice of
            // Sel3Args::transit, the parked one stays in its LDS columns
            double *tr = sa.sel->transit + (long long)blockIdx.x * (D + 5) * T3;
            __syncthreads();
            tr[tid] = v;
#pragma unroll
            for (int k = 0; k < D; ++k) tr[(5 + k) * T3 + tid] = x[k];
            tr[T3 + tid] = like; tr[2 * T3 + tid] = lprior; tr[3 * T3 + tid] = like_prev; tr[4 * T3 + tid] = acc_val;
            const int bad = k3_select_two<D>(sa.sel, cl.buf[0], cl.n, cl.R, g.N, g.V * g.nb1, g.V, s_ch2, cur, tr, st2, tag, n, ma.seed, ma.gid0, sa.g_cm + K3_RPAR(n), sa.to, &s_to, s_tot, s_vt, s_sw,
                                             red, z_park, po.shift);
            if (bad) { timed_out = true; break; }
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = tr[(5 + k) * T3 + tid];
            like = tr[T3 + tid]; lprior = tr[2 * T3 + tid]; like_prev = tr[3 * T3 + tid]; acc_val = tr[4 * T3 + tid];
            // chunk 0's draws of this stage again (the parking area was the selection's scratch)
            k3_draw_park<D, !ALPHA1>(z_park, ma.seed, pid_park, (unsigned)n, db0, ma.debug);
            __syncthreads();
            rs = 1; dec = 0;
        }
#endif
        if (dec != 0) {                                         // leave: nothing of the stage is committed, registers hold the cloud after stage n - 1
            if (writer && tid == 0) {
                if (dec < 0) { ma.rec.phi[n - 1] = s_a.bg.phi_n; ma.rec.ess[n - 1] = ess; ctl->status.err = dec; }
                ctl->status.stage = n;
                ctl->status.code = dec < 0 ? 9 : (dec == 4 ? 4 : 6);
            }
            break;
        }
        if (tid == T3 - 64) post2(n, s_a.bg, po, rp, s_tot[0], s_tot[1], ess, rs, &B.po);       // (the last wavefront: its logarithm runs beside the covariance and the shuffle of wavefronts 0 and 1)
        {
            Prop2 P{L.covl, L.Aw, L.mean_s, L.bfree, L.bptr_s, L.fi, L.Lraw, L.logdet_s, L.mub_raw, L.sdd_raw, L.sdn_raw, L.ball_raw, L.loff_s};
            if (!proposal2(s_tot + 2, po.shift, D, nf, nb, po.c * s_a.bg.cfac, ma.seed, (unsigned)n, P, &s_fail, T3, jx_pre, nullptr,
                           (sa.prof && writer && n == sa.prof_stage) ? sa.prof + 30 : nullptr)) {
                // PosDefException aborts the run (mutation.jl:81).  The gatherers build no proposal and would wait for mutation rows that
                // never come: the writer raises the waits' abort word (every bounded wait polls it), so they leave at once
                if (writer && tid == 0) {
                    ctl->status.err = SMCMI_ERR_POSDEF; ctl->status.stage = n; ctl->status.code = 9;
                    __hip_atomic_store(sa.to, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
        }
        if (tid < D) B.po.shift[tid] = L.mean_s[tid];
        __syncthreads();
        if (writer) {                                           // bookkeeping nobody in this launch waits for: Ctl2 / records / diagnostics (k2_bookkeeping)
            if (tid == 0) { ma.rec.phi[n - 1] = B.po.phi_n; ma.rec.ess[n - 1] = B.po.ess; ma.rec.resampled[n - 1] = rs; ma.rec.c[n - 1] = B.po.c; }
            if (tid < D) st->mean[tid] = L.mean_s[tid];
            for (int e = tid; e < D * D; e += T3) st->cov[e] = L.covl[e];
            constexpr int NWP = sizeof(Post2) / sizeof(double);
            if (tid < NWP) reinterpret_cast<double *>(&ctl->ps[n & 1])[tid] = reinterpret_cast<const double *>(&B.po)[tid];
        }
        K3_STAMP(sa.prof, 4);
        // ================= mutation (src/mutation.jl:56-138): normalize_weights!, the MH steps, one row per block
        const double phi_n = B.po.phi_n, nrm_sumw = B.po.sumw;
        K3_CHUNKS_BEGIN                                         // (two chunks: the parked chunk's turn first - it is in registers since the correction)
        double accept = 0.0;
        double step_prob, uc, z[D];
        if (K3_PARKED_DRAWS) {
            const double *p = z_park + tid;                     // (written by this thread)
            step_prob = p[0];
            uc = p[T3];
#pragma unroll
            for (int e = 0; e < D; ++e) z[e] = p[(2 + e) * T3];
        } else draw2<D>(ma.seed, pid, (unsigned)n, 0u, db0, ma.debug, step_prob, uc, z);       // (only chunk 0's first proposal is drawn ahead)
        if (live) {
            Wt = rs ? 1.0 : (v * nrm_N) / nrm_sumw;             // W·N then /ΣW̃, two roundings like the reference (particle.jl:362-366); 1 after a resample
            if (ma.hist_W && ma.store_history) ma.hist_W[(long long)(n - 1) * ma.hist_ld + i] = Wt;
        }
        k2_mh_steps<D, ALPHA1, T3, true>(L, mixbuf, mixpos, mixzt, ma, g.n, lv, mv, nb, nf, live, i, pid, (unsigned)n, phi_n, x, like, lprior, like_prev, accept,
                                         step_prob, uc, z);
        if (live) acc_val = accept / (double)nf;                // quirk Q2: normalised by n_free only
        K3_STAMP(sa.prof, 5);
        {
            double *plain = ma.rows_mut + (long long)rowi * RMUT;      // (the launch after this one totals the last stage's rows from here)
            unsigned long long *my_mut = sa.g_mut + K3_RPAR(n) + (long long)rowi * RMUT * 2;
            k2_mut_row_f<T3>(ma.adaptive != 0, like, like_prev, live ? Wt : 0.0, live ? acc_val : 0.0, e_center, live, rs != 0, red, L.red,
                             [&](int idx, double val) { if (K3_HAS) { gran_store(my_mut + idx * 2, val, tag); plain[idx] = val; } });
            if (K3_HAS && tid == RMUT - 1) gran_store(my_mut + tid * 2, 0.0, tag);                  // (column 33 is unused)
        }
        K3_CHUNKS_END
        ++done;
        K3_STAMP(sa.prof, 6);
        K3_WALL(sa.gprof, PROF2_WORK + 4 * blockIdx.x + 2);
        if constexpr (RIDE) rides = k3_rides(rp, sa, B.po, n, sys);             // stage n + 1's correction row goes out right behind this row?
        else {
            K3_DO_DRAW(n + 1);                                  // stage n + 1's draws, under the hand-over
            K3_DO_BEGIN(n + 1, B.po, false, act);               // ---- the V shard totals -> stage n + 1's begin
            if (act != 0) break;                                // leave: registers hold the cloud after stage n
        }
    }
#undef K3_RPAR
#undef K3_TPAR
#undef K3_DO_DRAW
#undef K3_DO_BEGIN
#undef K3_SWAP_CHUNKS
#undef K3_HAS
#undef K3_CHUNKS_BEGIN
#undef K3_CHUNKS_END
#undef K3_PARKED_DRAWS
#undef K3_V_RESET
    // ---- the cloud goes back to buffer 0 as the last completed stage left it
    if (live && !timed_out) {
#pragma unroll
        for (int k = 0; k < D; ++k) col(cl, 0, k)[i] = x[k];
        col(cl, 0, D)[i] = like; col(cl, 0, D + 1)[i] = lprior; col(cl, 0, D + 2)[i] = like_prev;
        col(cl, 0, D + 3)[i] = acc_val; col(cl, 0, D + 4)[i] = Wt;
    }
#ifdef SMCMI_K3_CH2
    {
        const long long io = beg_c[cur ^ 1] + tid;
        if (io < end_c[cur ^ 1] && !timed_out) {
            const double *s2 = st2 + tid;
#pragma unroll
            for (int k = 0; k < D + 5; ++k) col(cl, 0, k)[io] = s2[k * T3];
        }
    }
#endif
    if (writer && tid == 0) {
        if (sa.done_out) *sa.done_out = done;
        if (timed_out) { ctl->status.err = SMCMI_ERR_TIMEOUT; ctl->status.stage = n; ctl->status.code = 9; }
    }
    k3_leave_note(sa, ctl);
}

}  // namespace smcmi
