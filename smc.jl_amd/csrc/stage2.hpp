// stage2.hpp - the two-launch stage ("engine 2") for models with n_para <= 10.
//
// A tempering stage of the reference (src/smc_main.jl:377-508) needs two chip-wide hand-overs: the correction's weight sums and
// moments must be complete before the proposal can be built (smc_main.jl:427-469), and the mutated cloud's energies must be
// complete before the next ϕ can be chosen (helpers.jl:9-56).  Engine 1 (kernels.hpp) spends a single-block kernel on each of
// them (k_prepare_mutation, k_stage_begin: 44 % of the stage at N = 1e5).  Here both hand-overs are *launch-boundary reduces*:
//
//   k2_correct<D>  (K1)  prologue: every block totals the mutation's per-block rows (energy power sums, Σ accept, energy max) in a
//                        fixed order and runs the stage-begin logic itself (acceptance fold, energy shift, ϕ predictor / fixed
//                        schedule); block 0 records it.  Body: correction + moments at ϕ_n -> one row of sums per block.
//   k2_mutate<D,α1> (K2) prologue: every block totals the correction rows, takes the post-correction decision (ESS, verification
//                        of a predicted ϕ_n, resample?, c, log-MDD), builds the proposal (covariance, random blocks, Cholesky)
//                        in LDS; block 0 records it.  Body: the register-resident mutation -> one row of energy sums per block.
//
// No single-block launches, no in-kernel fences or atomics, no DevState round trips between the kernels of a stage; the kernel
// boundary is the only synchronisation.  Resample stages add k2_scan + k2_gather<D> (selection + moments of the resampled
// cloud) between the two; stages without a usable prediction run certificate passes k2_begin -> P x k2_pass -> k2_finish first.
//
// State discipline: a kernel never reads a word another block of the SAME launch may write.  K1 reads Post2[(n-1)&1] and rows,
// writes Begin2; K2 reads Begin2, Post2[(n-1)&1] and rows, writes Post2[n&1].  Every kernel carries its stage index n as a launch
// argument and does nothing unless the state it reads belongs to that stage - so a stage that stalls (prediction not verified,
// selection needed but not enqueued, solver out of passes) turns everything enqueued behind it into no-ops without a flag.
//
// Shard-count invariance: the N particles are cut into V = 8 *virtual shards* (fixed global ranges); every per-block quantity is
// defined per virtual shard, rows are totalled per virtual shard in a canonical order (groups of 64 rows; inside a group 8
// interleaved slices combined as a fixed tree; groups, then virtual shards, in ascending order) - so 1, 2, 4 or 8 GPUs holding
// 8, 4, 2 or 1 virtual shards each produce bit-identical sums, hence identical ϕ schedules, ancestors and clouds.
#pragma once
#include "kernels.hpp"

namespace smcmi {

constexpr int V2_MAXV = 8;        // virtual shards
constexpr int RMUT = 34;          // mutation row: ES = 32 sums (energy power sums | Σ accept), [32] = energy maximum, [33] unused
constexpr int RMAX_IDX = 32;
constexpr int T1 = 512;           // threads of a correction block
constexpr int GRP = 64;           // rows per canonical reduction group
constexpr int pad2(int m) { return (m + 1) & ~1; }   // row widths are even: rows are totalled with 16-byte loads (pad column = 0)

// development aid (SMCMI_PROF2=<stage>): ONE layout of the handle's stamp buffer (Eng2::d_prof, long long), so that no two users share a slot
// whatever mix of launches and segments the profiled stage runs through:
//   [PROF2_K1, +64)  K1's stamps: block 0 at +0.., the middle block at +32..   [PROF2_K2, +64)  K2's likewise (K2_STAMP below)
//   [PROF2_CENSUS, + 3 PROF2_BLOCKS)   wall-clock start / end and CU of every block of a large-shard mutation launch (stage2b.hpp)
//   [PROF2_SEG0, +64)   worker 0 of a segment: shader-clock stamps 1..9, the proposal's at +30..33 (K3_STAMP)
//   [PROF2_SEL, +16)    ... its in-place selection (wall clock)
//   [PROF2_GATH + 6 v, +6)  gatherer v's hand-over stamps, [PROF2_GPOLL + 4 v, +4) its polls    [PROF2_WORK + 4 b, +4)  worker b's hand-over stamps (wall clock)
constexpr int PROF2_BLOCKS = 4096;
constexpr int PROF2_K1 = 0, PROF2_K2 = 64, PROF2_CENSUS = 128, PROF2_SEG0 = PROF2_CENSUS + 3 * PROF2_BLOCKS, PROF2_SEL = PROF2_SEG0 + 64,
              PROF2_GATH = PROF2_SEL + 16, PROF2_GPOLL = PROF2_GATH + 64, PROF2_WORK = PROF2_GPOLL + 64, PROF2_WORDS = PROF2_WORK + 4 * 256;
// shader-clock stamps of thread 0 of block 0 (slots 0..31) and of the middle block (32..63)
#define K2_STAMP(prof, slot)                                                                                                   \
    do {                                                                                                                      \
        if ((prof) != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {                       \
            unsigned long long tt_;                                                                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");       \
            (prof)[(blockIdx.x == 0 ? 0 : 32) + (slot)] = (long long)tt_;                                                      \
        }                                                                                                                     \
    } while (0)

#define K3_STAMP_ANY(prof, slot)                                                                                               \
    do {                                                                                                                      \
        if ((prof) != nullptr && threadIdx.x == 0) {                                                                           \
            unsigned long long tt_;                                                                                           \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");       \
            (prof)[(slot)] = (long long)tt_;                                                                                   \
        }                                                                                                                     \
    } while (0)

struct Geo2 {
    long long N, n, nv;           // global / local / per-virtual-shard particles
    int V, Vl, v0;                // virtual shards in total / held by this handle / global index of the first local one
    int nb1, nb2, nbg;            // blocks per virtual shard: correction (and scan chunks) / mutation / gather
    long long per1, perg;         // particles per correction / gather block
    int t2;                       // threads (= particles) of a mutation block: 512 (the register kernels: one block per CU with the prologue
                                  // in every block, two per CU on large shards - stage2b.hpp); 256 / 64 the wide kernels
    int direct;                   // consumers total the per-block rows themselves (one handle, <= GRP rows per virtual shard)
    int inker;                    // every block runs the begin / decision / proposal logic in its prologue (<= one 512-thread block per CU):
                                  // direct, or several handles with small shards (the rows then are the all-gathered V x m totals)
    int wide;                     // n_para > 10 (k2w_mutate): lanes per particle of the mutation kernel (1 or 4), t2 = particles of a mutation
                                  // block (256 or 64); a mutation row is one block's, never paired; 0: the register kernels (n_para <= 10)
};

struct Rows2 {                    // rows[(v * nr + r) * ld + idx], v < nvs, r < nr
    const double *p;
    int nvs, nr, ld;
    const unsigned long long *mb; // non-null: the V x m totals arrive in this handle's mailbox table (Mailbox2 below) under `tag`
    unsigned tag;
    unsigned long long *to;       // the mailbox's time-out flag words (mb_load)
};

// ------------------------------------------------------------------------------------------------ peer mailbox (several GPUs)
// The two hand-overs of a stage (correction sums K1 -> K2, mutation sums K2 -> K1) between handles without a collective call: the
// block that totals a virtual shard (Tail2) writes the m totals straight into every peer's mailbox table - over xGMI into the
// peer's fine-grained memory when the handles sit on different GPUs - and every consumer block polls its own GPU's table.  The
// transport is the low-latency protocol collectives libraries use for small messages: every 64-bit word carries 32 bits of payload
// and a 32-bit tag (a double = two words), a word is stored and loaded atomically, so a reader that sees the expected tag in a
// word has that word's payload - no fence, no ordering assumption between words; a word not yet there is polled again.
// Table of one (kind, parity): [V2_MAXV][MB_LD] doubles = 2 words each.  Tags are unique per hand-over (run epoch | counter) and
// consecutive hand-overs of a kind alternate parity: a handle can only post hand-over c + 2 after every handle has consumed c.
constexpr int MB_LD = 72;                                      // row capacity in doubles (m <= 72)
constexpr int MB_TABLE_WORDS = V2_MAXV * MB_LD * 2;            // words of one (kind, parity) table
constexpr int MB_KINDS = 2;                                    // 0: correction rows, 1: mutation rows
constexpr int MB_WORDS = MB_KINDS * 2 * MB_TABLE_WORDS;        // a handle's whole mailbox
constexpr long long MB_TIMEOUT_TICKS_DEFAULT = 1000000000ll;    // ~10 s of the 100 MHz wall clock (SMCMI_MAILBOX_TIMEOUT_MS): a peer that never posts
                                                               // poisons the totals with NaN and raises the handle's time-out flag, which the
                                                               // driver turns into SMCMI_ERR_TIMEOUT (never into a "no particles" message)
constexpr int MB_FLAG_WORDS = 2;                               // behind the tables: word MB_WORDS = sticky time-out flag, MB_WORDS + 1 = time-out in ticks
// behind the flag words: the shard-total tables of persistent stage segments that span several handles (stage3.hpp: a gatherer posts its
// virtual shard's totals into every handle's copy, every block reads its own handle's) - [V2_MAXV][MB_LD] granule pairs per kind
constexpr int MB_SEG_OFF = MB_WORDS + MB_FLAG_WORDS;
constexpr int MB_SEG_KIND_WORDS = V2_MAXV * MB_LD * 2;
constexpr int MB_SEG_COPY_WORDS = 2 * MB_SEG_KIND_WORDS;        // both kinds; a riding launch uses two copies by stage parity (stage3.hpp Seg3Args)
constexpr int MB_ALLOC_WORDS = MB_SEG_OFF + 2 * MB_SEG_COPY_WORDS;
// behind them, in handles small enough for segments (mailbox.hpp mbox_sel_words): what a selection INSIDE sharded segments exchanges (stage3.hpp
// Sel3Args) - three tables of granules, then the whole cloud's cum column [N] and this handle's rows [(n_para + 4)][n] as plain doubles
constexpr long long MB_SEL_CS_WORDS = 1024 * 2, MB_SEL_T_WORDS = V2_MAXV * 2 * 2, MB_SEL_GM_WORDS = V2_MAXV * 72 * 2;
constexpr long long MB_SEL_OFF = MB_ALLOC_WORDS, MB_SEL_TABLE_WORDS = MB_SEL_CS_WORDS + MB_SEL_T_WORDS + MB_SEL_GM_WORDS;
__device__ inline void mb_store(unsigned long long *w, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, ((unsigned long long)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// to: the handle's flag words (MB_FLAG_WORDS).  to[0] is sticky: after the first time-out every wait gives up at once
__device__ inline double mb_load(const unsigned long long *w, unsigned tag, unsigned long long *to) {
    unsigned long long a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(a >> 32) != tag || (unsigned)(b >> 32) != tag) {
        const long long t0 = wall_clock64();
        do {
            if (__hip_atomic_load(to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return __builtin_nan("");
            __builtin_amdgcn_s_sleep(2);
            a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (wall_clock64() - t0 > (long long)__hip_atomic_load(to + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(to, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return __builtin_nan("");
            }
        } while ((unsigned)(a >> 32) != tag || (unsigned)(b >> 32) != tag);
    }
    return __longlong_as_double((long long)((b << 32) | (a & 0xffffffffull)));
}
// totals of columns [0, M) over the R.nvs virtual shards from the mailbox: the order of reduce_rows on a one-row-per-shard table
// (0 + x_0 + x_1 + ...), so the result is the one the all-gathered table gives.  All T threads call; ends with a barrier.
template <int M, int T>
__device__ inline void mbox_totals(const Rows2 &R, double *vt, double *tot, int max_idx) {
    for (int idx = threadIdx.x; idx < R.nvs * M; idx += T) {
        const int v = idx / M, k = idx % M;
        vt[idx] = mb_load(R.mb + ((long long)v * MB_LD + k) * 2, R.tag, R.to);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < M; k += T) {
        const bool mx = k == max_idx;
        double t = mx ? -__builtin_inf() : 0.0;
        for (int v = 0; v < R.nvs; ++v) { const double x = vt[v * M + k]; t = mx ? fmax(t, x) : t + x; }
        tot[k] = t;
    }
    __syncthreads();
}

struct Begin2 {                   // stage n as decided at its begin (K1 / k2_begin / k2_finish, block 0)
    int stage, final, spec, j;
    double phi_prev, phi_n, phi_prop, ess_bar, gprime, pred_delta;
    double accept;                // cloud.accept: acceptance rate of stage n-1's mutation (particle.jl:466-468)
    double e_shift, e_center;
    double cfac;                  // step-size multiplier 0.95 + 0.10 e^{16(a-t)} / (1 + e^{16(a-t)}) (smc_main.jl:453-455), stored by K1's block 0
    double e_seen;                // the largest energy of the cloud as this begin learnt it (after stage n - 1's mutation).  Adaptive schedules shift
                                  // stage n's incremental weights by it (e_shift = e_seen); fixed schedules under RunParams::shift_lag by the one the
                                  // begin BEFORE learnt (e_shift = Post2::e_seen of stage n - 1), so that stage n's correction needs nothing of stage
                                  // n - 1's mutation rows: one hand-over per stage (stage3.hpp)
};
struct Post2 {                    // stage n after its correction (K2, block 0); two copies, indexed by n & 1
    int stage, j, resampled_last, do_resample, resamples, fold_valid;
    double phi_n, phi_prop, ess, sumw, sumw2, logz, c, accept, e_center, e_shift;
    double e_seen;                // Begin2::e_seen of the stage (NaN: none yet - a fresh run, a continuation from saved scalars)
    double shift[MAXD >= 16 ? 16 : MAXD];
};
struct Status2 {
    int code;                     // 0 running, 1 finished (ϕ = 1 reached, last acceptance rate folded), 2 solver out of passes,
                                  // 3 selection needed but not enqueued, 4 predicted ϕ_n unusable / not verified, 5 paused, 9 error
    int stage, err, pad;
    long long solver_passes;
    double accept;                // cloud.accept at the end of the run / at the pause
};
struct Ctl2 {
    Status2 status;
    Begin2 bg;
    Post2 ps[2];
};
static_assert(sizeof(Begin2) % 8 == 0 && sizeof(Post2) % 8 == 0, "state structs are copied as doubles");

// ------------------------------------------------------------------------------------------------ canonical row totals
template <int SL>
__device__ inline double slice_tree(const double (&a)[SL]) {
    if constexpr (SL == 8) return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    else if constexpr (SL == 4) return (a[0] + a[1]) + (a[2] + a[3]);
    else if constexpr (SL == 2) return a[0] + a[1];
    else return a[0];
}
template <int SL>
__device__ inline double slice_tree_max(const double (&a)[SL]) {
    double m = a[0];
#pragma unroll
    for (int q = 1; q < SL; ++q) m = fmax(m, a[q]);
    return m;
}

// Totals of M (even) columns over R.nvs (<= V2_MAXV) virtual shards x R.nr (<= NRMAX <= GRP) rows in the canonical order, by the
// whole block of T threads.  A unit = (virtual shard, pair of adjacent columns, h): H threads share a (shard, pair), each owning
// 8 / H of the 8 slices (slice s = rows s, s + 8, ... in ascending order); the result does not depend on H, T or NRMAX.  Column
// max_idx (if >= 0) is a maximum instead of a sum.  Loads are 16 bytes wide and unconditional (rows beyond nr re-read the last
// row and contribute the identity): a load under a run-time condition makes the compiler wait for each one separately, i.e. one
// memory round trip per row instead of one per call; with few rows all of a thread's loads are in flight at once.
// vt: LDS scratch of V2_MAXV * M * H doubles; tot: LDS, M doubles.  All threads must call; ends with a barrier.
template <int M, int H, int NRMAX, int T>
__device__ inline void reduce_rows_ct(const Rows2 &R, double *vt, double *tot, int max_idx = -1) {
    static_assert(M % 2 == 0, "columns are totalled in pairs (16-byte loads)");
    static_assert(NRMAX % 8 == 0 && NRMAX <= GRP, "NRMAX: multiple of 8, at most one group");
    constexpr int MP = M / 2, SL = 8 / H, NJ = NRMAX / 8, UPT = (V2_MAXV * MP * H + T - 1) / T;
    constexpr bool FULL = UPT * NJ * SL <= 16;           // all units' loads in one batch while the load targets stay <= 64 VGPRs
    const int units = R.nvs * MP * H, nr = R.nr;
    const double ninf = -__builtin_inf();
    auto unit = [&](int k) {
        const int u = (int)threadIdx.x + k * T;
        if ((u & ~63) >= units) return;                  // the whole wavefront has no unit in this batch
        const int uc = u < units ? u : 0;
        const int h = uc % H, pr = (uc / H) % MP, v = uc / (H * MP);
        const bool mx0 = 2 * pr == max_idx, mx1 = 2 * pr + 1 == max_idx;
        const double id0 = mx0 ? ninf : 0.0, id1 = mx1 ? ninf : 0.0;
        const double2 *base = reinterpret_cast<const double2 *>(R.p + ((long long)v * nr) * R.ld + 2 * pr);
        const long long ldp = R.ld / 2;
        double a0[SL], a1[SL];
#pragma unroll
        for (int q = 0; q < SL; ++q) { a0[q] = id0; a1[q] = id1; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q = 0; q < SL; ++q) {
                const int r = h * SL + q + 8 * j;
                const int rc = r < nr ? r : nr - 1;
                const double2 x = base[(long long)rc * ldp];
                const double x0 = r < nr ? x.x : id0, x1 = r < nr ? x.y : id1;
                a0[q] = mx0 ? fmax(a0[q], x0) : a0[q] + x0;
                a1[q] = mx1 ? fmax(a1[q], x1) : a1[q] + x1;
            }
        }
        if (u < units) {
            vt[(v * M + 2 * pr) * H + h] = mx0 ? slice_tree_max<SL>(a0) : slice_tree<SL>(a0);
            vt[(v * M + 2 * pr + 1) * H + h] = mx1 ? slice_tree_max<SL>(a1) : slice_tree<SL>(a1);
        }
    };
    if constexpr (FULL) {
#pragma unroll
        for (int k = 0; k < UPT; ++k) unit(k);
    } else {
#pragma unroll 1
        for (int k = 0; k < UPT; ++k) {
            if (k * T >= units) break;                     // block-uniform
            unit(k);
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < M; idx += T) {
        const bool mx = idx == max_idx;
        double t = mx ? ninf : 0.0;
        for (int v = 0; v < R.nvs; ++v) {
            double p[H];
#pragma unroll
            for (int h = 0; h < H; ++h) p[h] = vt[(v * M + idx) * H + h];
            const double vs = mx ? slice_tree_max<H>(p) : slice_tree<H>(p);
            t = mx ? fmax(t, vs) : t + vs;
        }
        tot[idx] = t;
    }
    __syncthreads();
}
// the same with the row capacity picked at run time (8 / 16 / 32 / 64)
template <int M, int H, int T>
__device__ inline void reduce_rows(const Rows2 &R, double *vt, double *tot, int max_idx = -1) {
    if (R.mb) { mbox_totals<M, T>(R, vt, tot, max_idx); return; }
    if (R.nr <= 8) reduce_rows_ct<M, H, 8, T>(R, vt, tot, max_idx);
    else if (R.nr <= 16) reduce_rows_ct<M, H, 16, T>(R, vt, tot, max_idx);
    else if (R.nr <= 32) reduce_rows_ct<M, H, 32, T>(R, vt, tot, max_idx);
    else reduce_rows_ct<M, H, 64, T>(R, vt, tot, max_idx);
}

// Virtual-shard totals of a row set with any number of rows (sharded runs and large clouds): block v totals the nr rows of its
// virtual shard in the canonical order (groups of GRP rows by slice tree, groups in ascending order) -> out[v][m].  Four adjacent
// lanes share a (group, column pair): each owns two of the eight slices (sixteen unconditional 16-byte loads in flight), the
// slice tree is finished with two DPP exchanges inside the quad.  m even, <= 72.
constexpr int RT = 1024;
// pair = 1 (mutation rows written by 256-thread blocks): a logical row is raw row 2r + raw row 2r+1 - the row a 512-thread
// block over the same particles writes (block_reduce_es2).
// By a whole block of NT threads (k2_reduce: its own launch; or the last block of a virtual shard to finish, Tail2 below): the
// result does not depend on NT (groups are totalled in ascending order however many fit a batch).
// COH: the rows were written by other blocks of the SAME launch (Tail2): read them with agent-scope loads that bypass this die's L2.
__device__ inline void row_store(double *p, double v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ inline double2 load_pair(const double2 *p, bool coh) {
    if (!coh) return *p;
    const double *q = reinterpret_cast<const double *>(p);
    double2 x;
    x.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return x;
}
// The same 16 bytes with agent-scope cache policy (sc1: served from memory / MALL, not from this die's L2) as a load the compiler
// counts and pipelines like any other: sixteen of them are in flight at once, where sixteen atomic loads are sixteen round trips
// (3.6 -> ~1 µs for the totals of a virtual shard).  rsrc: buffer descriptor over the rows (wave-uniform), off: byte offset.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ inline double2 load_pair_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, /*aux = sc1*/ 16);
    double2 x;
    x.x = __hiloint2double((int)v.y, (int)v.x);
    x.y = __hiloint2double((int)v.w, (int)v.z);
    return x;
}
__device__ inline double load_f64_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, /*aux = sc1*/ 16);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ inline __amdgpu_buffer_rsrc_t rows_rsrc(const double *base, long long bytes) {
    // (the base is the same in every lane; readfirstlane tells the compiler so)
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
// LD: double2 ldrow(int row, int pr) - columns 2 pr, 2 pr + 1 of raw row `row` of the virtual shard.  Returns, in thread t < m, total t.
// MMAX: capacity in columns (72: every row of the register kernels; 160: the correction rows of n_para up to 16)
template <int NT, class LD, int MMAX = 72>
__device__ inline double reduce_vshard_f(LD ldrow, int nr_raw, int m, int max_idx, int pair) {
    constexpr int GB = NT / (MMAX / 2 * 4) > 0 ? NT / (MMAX / 2 * 4) : 1;      // groups per batch: GB * (m / 2) * 4 <= NT for m <= MMAX
    static_assert(NT >= MMAX / 2 * 4, "a group needs (m / 2) * 4 threads");
    __shared__ double gs[GB * MMAX];
    const int mp = m / 2;
    const int nr = pair ? (nr_raw + 1) / 2 : nr_raw;
    const int ng = (nr + GRP - 1) / GRP;
    const double ninf = -__builtin_inf();
    double run = (int)threadIdx.x == max_idx ? ninf : 0.0;
    for (int g0 = 0; g0 < ng; g0 += GB) {
        const int gb = (ng - g0) < GB ? (ng - g0) : GB;
        const int units = gb * mp * 4, u = threadIdx.x;
        if ((u & ~63) < units) {                        // wave-uniform: idle wavefronts skip the loads
            const int uc = u < units ? u : 0;
            const int h = uc & 3, pr = (uc >> 2) % mp, gl = (uc >> 2) / mp, g = g0 + gl;
            const bool mx0 = 2 * pr == max_idx, mx1 = 2 * pr + 1 == max_idx;
            const double id0 = mx0 ? ninf : 0.0, id1 = mx1 ? ninf : 0.0;
            const int r_beg = g * GRP, r_end = (r_beg + GRP < nr) ? r_beg + GRP : nr;
            double a0[2] = {id0, id0}, a1[2] = {id1, id1};
#pragma unroll
            for (int j = 0; j < GRP / 8; ++j) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = r_beg + 2 * h + q + 8 * j;
                    const int rc = r < r_end ? r : r_end - 1;
                    double2 x;
                    if (pair) {                         // (uniform condition; both loads unconditional)
                        const int ra = 2 * rc, rb = 2 * rc + 1 < nr_raw ? 2 * rc + 1 : ra;
                        const double2 xa = ldrow(ra, pr), xb = ldrow(rb, pr);
                        const bool hb = 2 * rc + 1 < nr_raw;
                        x.x = mx0 ? fmax(xa.x, hb ? xb.x : id0) : xa.x + (hb ? xb.x : 0.0);
                        x.y = mx1 ? fmax(xa.y, hb ? xb.y : id1) : xa.y + (hb ? xb.y : 0.0);
                    } else x = ldrow(rc, pr);
                    const double x0 = r < r_end ? x.x : id0, x1 = r < r_end ? x.y : id1;
                    a0[q] = mx0 ? fmax(a0[q], x0) : a0[q] + x0;
                    a1[q] = mx1 ? fmax(a1[q], x1) : a1[q] + x1;
                }
            }
            // slices 2h, 2h+1 -> ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)) across the quad
            double p0 = mx0 ? fmax(a0[0], a0[1]) : a0[0] + a0[1], p1 = mx1 ? fmax(a1[0], a1[1]) : a1[0] + a1[1];
            const double q0 = fetch_xor<1>(p0), q1 = fetch_xor<1>(p1);
            p0 = mx0 ? fmax(p0, q0) : p0 + q0; p1 = mx1 ? fmax(p1, q1) : p1 + q1;
            const double r0 = fetch_xor<2>(p0), r1 = fetch_xor<2>(p1);
            p0 = mx0 ? fmax(p0, r0) : p0 + r0; p1 = mx1 ? fmax(p1, r1) : p1 + r1;
            if (u < units && h == 0) { gs[gl * MMAX + 2 * pr] = p0; gs[gl * MMAX + 2 * pr + 1] = p1; }
        }
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const bool mx = (int)threadIdx.x == max_idx;
            for (int g = 0; g < gb; ++g) run = mx ? fmax(run, gs[g * MMAX + threadIdx.x]) : run + gs[g * MMAX + threadIdx.x];
        }
        __syncthreads();
    }
    return run;
}
template <int NT, bool COH = false, int MMAX = 72>
__device__ inline void reduce_vshard(const double *base0, int nr_raw, int m, int max_idx, double *out_v, int pair) {
    const __amdgpu_buffer_rsrc_t rsrc = rows_rsrc(base0, (long long)nr_raw * m * 8);
    const unsigned brow = (unsigned)m * 8u;
    auto ldrow = [&](int row, int pr) {
        if constexpr (COH) return load_pair_sc1(rsrc, (unsigned)(2 * pr) * 8u + (unsigned)row * brow);
        else return reinterpret_cast<const double2 *>(base0 + 2 * pr)[(long long)row * (m / 2)];
    };
    // (round 5: staging the rows through LDS with the whole block's loads - what made engine 3's gatherers 2.5x faster, stage3.hpp
    // gather_vshard - was measured here too: K2b's tail indifferent, K1's 1.5 / 4 µs SLOWER at 250 000 / 500 000 particles per handle: one
    // 16-byte load per unit and row is already one round trip, and 35 KB more LDS per block are not free)
    const double run = reduce_vshard_f<NT, decltype(ldrow), MMAX>(ldrow, nr_raw, m, max_idx, pair);
    if ((int)threadIdx.x < m) row_store(out_v + threadIdx.x, run, COH);     // (COH: a block of the same launch may read the totals)
}
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(RT) k2_reduce(const double *rows, int nr_raw, int m, int max_idx, double *out, int pair) {
    const int v = blockIdx.x;
    if (m <= 72) reduce_vshard<RT>(rows + (long long)v * nr_raw * m, nr_raw, m, max_idx, out + (long long)v * m, pair);
    else reduce_vshard<RT, false, 160>(rows + (long long)v * nr_raw * m, nr_raw, m, max_idx, out + (long long)v * m, pair);
}
#endif

// The same totals without a launch of their own (sharded runs, large clouds): every block of a row-producing kernel takes a
// ticket of its virtual shard once its row is stored; the block that draws the last ticket totals the shard's rows - in the
// canonical order, so the result is the one k2_reduce gives - and re-arms the counter.  No fence: an agent-scope release would
// write back everything the kernel has left dirty in its die's L2 (the particle columns) once per block - measured 23 -> 58 µs for
// the mutation kernel; instead the rows themselves are stored and read with agent-scope accesses (write-through / L2 bypass,
// `row_store`, `load_pair`), ordered by the wait for their completion, the block barrier and the relaxed ticket.  Blocks that
// leave early (a stage that does not run) take no ticket.  All threads of the block call, after the row is stored.
constexpr int TICK2_STRIDE = 32;   // ints between a handle's ticket counters: one 128-byte line each.  Agent-scope atomics execute at the memory side; eight
                                   // counters on ONE line serialised all 984 arrivals of a 250 000-particle mutation launch (~15 of its 37 µs)
struct Tail2 {
    int *tick;                 // [Vl * TICK2_STRIDE] counters (counter v at tick[v * TICK2_STRIDE]), zero between launches; null: no tail (direct geometry: consumers read the rows)
    double *vt;                // this handle's slice of the V x m table: vt[v * m]
    unsigned long long *const *peers;   // non-null: post the totals into every handle's mailbox (peers[r], r < world) instead of an all-gather
    int world, gv0;            // handles; global index of this handle's first virtual shard
    long long table;           // word offset of the (kind, parity) table inside a mailbox
    unsigned tag;
};
template <int NT, int MMAX = 72>
__device__ inline void tail_reduce(const Tail2 &t, const double *rows, int v, int nr_raw, int m, int max_idx, int pair) {
    if (!t.tick) return;
    __shared__ int s_last;
    // every wavefront waits for its own row stores to be acknowledged before the barrier that precedes the ticket (a workgroup-scope
    // release fence does not: outside tgsplit mode the compiler omits the vmcnt wait there, and the ticket could overtake a row)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&t.tick[v * TICK2_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nr_raw - 1;
    __syncthreads();
    if (!s_last) return;
    reduce_vshard<NT, true, MMAX>(rows + (long long)v * nr_raw * m, nr_raw, m, max_idx, t.vt + (long long)v * m, pair);
    if (t.peers && (int)threadIdx.x < m) {             // (thread k < m stored total k just above)
        const double x = t.vt[(long long)v * m + threadIdx.x];
        const long long w = t.table + ((long long)(t.gv0 + v) * MB_LD + threadIdx.x) * 2;
        for (int r = 0; r < t.world; ++r) mb_store(t.peers[r] + w, x, t.tag);
    }
    if (threadIdx.x == 0) __hip_atomic_store(&t.tick[v * TICK2_STRIDE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block-wide fixed-order reduction of M accumulators per thread for a block of NW wavefronts; thread t < M gets total t.
// red: NW * M doubles of LDS.
template <int M, int NW>
__device__ inline double block_reduce_nw(double (&a)[M], double *red) {
    static_assert(M >= 1 && M <= 64 && (M & (M - 1)) == 0, "M must be a power of two <= 64");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Butterfly<M / 2, 32>::run(a, lane);
    constexpr int SH = 6 - ilog2(M);
    if ((lane & ((1 << SH) - 1)) == 0) red[wave * M + (lane >> SH)] = a[0];
    __syncthreads();
    double tot = 0.0;
    if ((int)threadIdx.x < M) {
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w * M + threadIdx.x];
    }
    __syncthreads();
    return tot;
}

// Block-wide fixed-order reduction of the ES = 32 epilogue sums for a mutation block of NW = 4 or 8 wavefronts.  A row stands
// for 256 particles = four wavefronts summed in wavefront order; an 8-wavefront block adds its two halves, which is exactly how
// the rows of two 4-wavefront blocks over the same particles are paired by k2_reduce - the canonical order does not depend on
// the block size.  red: NW * ES doubles of LDS; thread t < ES gets total t.
template <int NW>
__device__ inline double block_reduce_es2(double (&a)[ES], double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Butterfly<ES / 2, 32>::run(a, lane);
    constexpr int SH = 6 - ilog2(ES);
    __syncthreads();
    if ((lane & ((1 << SH) - 1)) == 0) red[wave * ES + (lane >> SH)] = a[0];
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < ES) {
        const int t = threadIdx.x;
        tot = ((red[t] + red[ES + t]) + red[2 * ES + t]) + red[3 * ES + t];
        if constexpr (NW == 8) tot += ((red[4 * ES + t] + red[5 * ES + t]) + red[6 * ES + t]) + red[7 * ES + t];
    }
    __syncthreads();
    return tot;
}

// particle range of block r of local virtual shard vl (per = particles per block)
__device__ inline void vchunk(const Geo2 &g, int vl, int r, long long per, long long &beg, long long &end) {
    // (one handle with virtual shards of ceil(n / V) particles - run2.hpp make_geo2_uneven: the last shard ends at n)
    const long long v_beg0 = (long long)vl * g.nv, v_beg = v_beg0 < g.n ? v_beg0 : g.n, v_end = v_beg + g.nv < g.n ? v_beg + g.nv : g.n;
    beg = v_beg + (long long)r * per;
    end = beg + per < v_end ? beg + per : v_end;
    if (beg > v_end) beg = v_end;
}

// ------------------------------------------------------------------------------------------------ state import / export
// Engine 2 keeps its loop state in Ctl2; the C ABI's stand-alone calls, pause / continue and the result read DevState.  One
// thread copies one into the other at the start / end of a run.
// fresh: a new run's first records (ϕ_1 = 0, ESS, c, target acceptance: smc_main.jl:337-352) are written here, not by four host copies
__device__ inline void k2_export_state(DevState *st, const Ctl2 *ctl) {
    const Post2 &p = ctl->ps[0].stage >= ctl->ps[1].stage ? ctl->ps[0] : ctl->ps[1];
    const Status2 &s = ctl->status;
    st->stage = p.stage; st->j = p.j; st->resampled_last = p.resampled_last; st->do_resample = p.do_resample; st->resamples = p.resamples;
    st->phi_n = p.phi_n; st->phi_prop = p.phi_prop; st->ess_prev = p.ess; st->ess = p.ess; st->sumw = p.sumw; st->sumw2 = p.sumw2;
    st->logz = p.logz; st->c = p.c; st->accept = (s.code == 1 || s.code == 5) ? s.accept : p.accept; st->e_center = p.e_center;
    st->e_seen = p.e_seen;
    if (ctl->bg.stage == p.stage) st->phi_prev = ctl->bg.phi_prev;
    for (int a = 0; a < (int)(sizeof(p.shift) / sizeof(double)); ++a) { st->shift[a] = p.shift[a]; st->mean[a] = p.shift[a]; }
    st->solver_passes = s.solver_passes;
    st->err = s.err;
    st->done = s.code == 1 ? 1 : (s.code == 5 ? 5 : (s.code == 9 ? 1 : (s.code ? s.code : 0)));
    st->skip_fold = 0;
}
// to_ctl: DevState -> Ctl2 (a run starts); else Ctl2 -> DevState (it ends).  One kernel: the two directions are never needed in one launch
#ifndef SMCMI_INST_UNIT
static __global__ void k2_state(DevState *st, Ctl2 *ctl, int to_ctl, Records rec = Records{}, int fresh = 0, double ess0 = 0.0, double c0 = 0.0, double acc0 = 0.0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!to_ctl) { k2_export_state(st, ctl); return; }
    if (fresh) { rec.phi[0] = 0.0; rec.ess[0] = ess0; rec.c[0] = c0; rec.accept[0] = acc0; }
    Ctl2 c;
    memset(&c, 0, sizeof(c));
    Post2 &p = c.ps[st->stage & 1];
    p.stage = st->stage; p.j = st->j; p.resampled_last = st->resampled_last; p.do_resample = 0; p.resamples = st->resamples;
    p.fold_valid = 0;                                  // no mutation rows of this chain exist yet (fresh or continued run)
    p.phi_n = st->phi_n; p.phi_prop = st->phi_prop; p.ess = st->ess_prev; p.sumw = st->sumw; p.sumw2 = st->sumw2;
    p.logz = st->logz; p.c = st->c; p.accept = st->accept; p.e_center = st->e_center; p.e_shift = 0.0;
    p.e_seen = st->e_seen;                             // (NaN on a fresh run; what the export left when a paused run goes on)
    for (int a = 0; a < (int)(sizeof(p.shift) / sizeof(double)); ++a) p.shift[a] = st->shift[a];
    c.ps[(st->stage & 1) ^ 1].stage = -1;
    c.bg.stage = -1;
    c.status.solver_passes = st->solver_passes;
    *ctl = c;
}
#endif

// ------------------------------------------------------------------------------------------------ stage begin
// src/smc_main.jl:378-396 + src/helpers.jl:9-56 (see k_stage_begin in kernels.hpp for the predictor and the candidate set): run
// by wavefront 0 of EVERY block of the stage's first kernel from the same inputs; `writer` (block 0) alone stores to global
// memory.  s_es: totals of the previous mutation's rows (valid iff rows_valid); s_sw: window of the proposed schedule,
// s_sw[q] = schedule[j + q] (1-based j), 2.0 beyond the end.  Result in *bg (LDS, written by lane 0).
// Returns: 0 ϕ_n decided (bg->final), 1 run finished, 5 paused, 9 capacity error, 4 no usable prediction (spec_expected),
//          6 solver armed in *arm (certificate passes follow).
__device__ inline int begin2_wave(int n, const Post2 &po, const RunParams &rp, const double *s_es, double emax_tot, bool rows_valid,
                                  int spec_expected, const double *sched, const double *s_sw, Begin2 *bg, Solver *arm, bool writer,
                                  const Records &rec, Status2 *status, double inv_pre = -1.0) {
    const int lane = threadIdx.x & 63;
    const int stage0 = n - 1, n_phi = rp.n_phi, fixed = rp.use_fixed_schedule, j = po.j;
    const double N = (double)rp.n_parts, phi_n = po.phi_n, phi_prop = po.phi_prop, target = rp.tempering_target;
    const bool fold = rows_valid && po.fold_valid && stage0 > 1;
    const bool have_es = fold && !fixed;
    const double accept = fold ? s_es[EACC] / N : po.accept;
    if (writer && lane == 0 && fold) rec.accept[stage0 - 1] = accept;
    if (phi_n >= 1.0) { if (writer && lane == 0) { status->accept = accept; status->stage = stage0; status->code = 1; } return 1; }
    if (rp.stop_stage > 0 && stage0 >= rp.stop_stage) { if (writer && lane == 0) { status->accept = accept; status->stage = stage0; status->code = 5; } return 5; }
    if (n > rp.max_stages) { if (writer && lane == 0) { status->err = SMCMI_ERR_CAPACITY; status->stage = stage0; status->code = 9; } return 9; }
    Begin2 b;
    b.stage = n; b.final = 0; b.spec = 0; b.j = j;
    b.phi_prev = phi_n; b.phi_n = phi_n; b.phi_prop = phi_prop; b.ess_bar = 0.0;
    b.gprime = __longlong_as_double(0x7ff8000000000000ll); b.pred_delta = b.gprime;
    b.accept = accept;
    b.e_shift = fabs(emax_tot) < 1e300 ? emax_tot : po.e_shift;     // no live particle with a finite energy: keep the previous shift
    b.e_seen = fabs(emax_tot) < 1e300 ? emax_tot : (fabs(po.e_seen) < 1e300 ? po.e_seen : po.e_shift);
    // fixed schedules, RunParams::shift_lag: the shift is the maximum the PREVIOUS begin learnt (any common shift leaves W, ESS and log-MDD
    // what they are up to rounding; this one is known before stage n - 1's mutation rows are: stage n's correction can ride them)
    // (development, SMCMI_SHIFT_LAG=<k >= 3>: stage k's lagged shift is lowered by 1e6 - mathematically neutral, but its sums overflow: the
    // fallback to exact shifts, run2.hpp, under test)
    if (fixed && rp.shift_lag && fabs(po.e_seen) < 1e300) b.e_shift = po.e_seen - (rp.shift_lag == n ? 1e6 : 0.0);
    b.e_center = po.e_center;
    if (fixed) {
        b.phi_n = (n <= n_phi) ? sched[n - 1] : 1.0;
        b.final = 1;
        if (lane == 0) *bg = b;
        return 0;
    }
    const int rl = po.resampled_last;
    const double ess_bar = target * (rl ? N : po.ess), ess_now = rl ? N : po.ess;      // helpers.jl:14-20
    b.ess_bar = ess_bar;
    double pd = b.gprime, gp = b.gprime;
    if (have_es) {
        pd = predict_delta_wave(s_es, po.do_resample != 0, ess_bar, &gp, inv_pre, false);
        const double ec = po.e_center + s_es[1] / s_es[0];         // weighted mean energy: centre for the next epilogue
        if (fabs(ec) < 1e300) b.e_center = ec;
    }
    b.pred_delta = pd; b.gprime = gp;
    const int q_end = n_phi - j + 1;                         // last existing walk step
    const double ph = phi_n + pd;
    bool use_pred = pd > 0.0 && ph < 1.0;
    const double wl = lane == 0 ? phi_prop : s_sw[lane - 1];
    const unsigned long long above = __ballot(lane <= 62 && lane <= q_end && wl > ph);
    int qstar = above ? (__ffsll((long long)above) - 1) : (q_end <= 62 ? q_end : -1);
    if (qstar < 0) use_pred = false;
    if (spec_expected) {
        const bool none_above = !above;
        const bool beyond = pd > 0.0 && ph >= 1.0 && qstar >= 0 && none_above;
        if ((use_pred || beyond) && gp < 0.0) {
            const double step_q = __shfl(wl, qstar, 64);
            b.final = 1; b.spec = 1;
            b.phi_prop = step_q; b.j = j + qstar;
            b.phi_n = none_above ? step_q : ph;
            if (lane == 0) *bg = b;
            return 0;
        }
        if (writer && lane == 0) { status->stage = n; status->code = 4; }
        return 4;
    }
    // certificate path: arm the solver with the walk steps (and the rings around a prediction) - block 0 only
    if (lane == 0) *bg = b;
    if (!writer) return 6;
    Solver &S = *arm;
    if (lane == 0) {
        S.unconverged = 0; S.ess_bar = ess_bar; S.lo = phi_n; S.phi0 = phi_n; S.glo = ess_now - ess_bar; S.hi = phi_prop; S.ghi = 0.0;
        S.j = j; S.phi_prop = phi_prop; S.spec = 0; S.gprime = gp;
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
    if (!use_pred) {
        cq = lane; keep = lane < KC && lane <= q_end;
        x = keep ? wl : 0.0;
    } else {
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
    return 6;
}

// Shared front of K1 and k2_begin: load Post2[(n-1)&1], total the mutation rows, run begin2_wave.  Returns the action code to all
// threads; s_bg holds stage n's Begin2 when the action is 0 or 6.  LDS: s_po, s_bg, s_vt (V2_MAXV * RMUT * 4 doubles), s_tot (RMUT),
// s_sw (64), s_act.
// writer_ov: -1 = block 0 stores the stage's state (every block of a launch runs this); 0 / 1 = this block does not / does (a helper block
// behind the blocks of another kernel, stage2b.hpp).  MBONLY: the rows arrive through the mailbox - nothing of reduce_rows_ct is instantiated
// (a kernel whose other blocks live on a small register budget)
template <int T, bool MBONLY = false>
__device__ inline int begin2_block(int n, DevState *st, Ctl2 *ctl, const Rows2 &mrows, int spec_expected, const double *sched,
                                   const Records &rec, Post2 *s_po, Begin2 *s_bg, double *s_vt, double *s_tot, double *s_sw, int *s_act,
                                   long long *prof = nullptr, bool coh_bg = false, int writer_ov = -1) {
    const int t = threadIdx.x;
    const bool writer = writer_ov >= 0 ? writer_ov != 0 : blockIdx.x == 0;
    constexpr int NWP = sizeof(Post2) / sizeof(double);
    if (t < NWP) reinterpret_cast<double *>(s_po)[t] = reinterpret_cast<const double *>(&ctl->ps[(n - 1) & 1])[t];
    if (t == 0) *s_act = -1;
    // every load of the prologue is issued before the first wait: the state copy above, the schedule window (index from a direct
    // read of j instead of the LDS copy: one round trip less) and the rows
    const int j_direct = ctl->ps[(n - 1) & 1].j;
    const double inv_pre = INV_FACTORIAL[t & 31];        // 1 / k! of the predictor's lane (a dependent global load if taken there)
    K2_STAMP(prof, 1);
    const bool rows_valid = mrows.p != nullptr;
    double swv = 2.0;
    if (t < 64) {
        const int jj = j_direct - 1 + t;                 // 0-based index of walk step t + 1
        if (!st->rp.use_fixed_schedule && jj >= 0 && jj < st->rp.n_phi) swv = sched[jj];
    }
    if (mrows.mb) {                                      // mailbox: a launch that will not run must not wait for rows nobody posts
        __syncthreads();
        if (s_po->stage != n - 1) return -1;
    }
    if (rows_valid) {
        if constexpr (MBONLY) mbox_totals<RMUT, T>(mrows, s_vt, s_tot, RMAX_IDX);
        else reduce_rows<RMUT, 2, T>(mrows, s_vt, s_tot, RMAX_IDX);
    } else {
        if (t < RMUT) s_tot[t] = t == RMAX_IDX ? -__builtin_inf() : 0.0;
        __syncthreads();
    }
    if (s_po->stage != n - 1) return -1;                 // the state this launch was enqueued for is not there: no-op
    if (t < 64) s_sw[t] = swv;                           // (read by wavefront 0 only, which wrote it)
    K2_STAMP(prof, 2);
    if (t < 64) {
        const int act = begin2_wave(n, *s_po, st->rp, s_tot, s_tot[RMAX_IDX], rows_valid && s_po->fold_valid, spec_expected, sched, s_sw, s_bg,
                                    &st->sol[0], writer, rec, &ctl->status, inv_pre);
        if (t == 0) *s_act = act;
    }
    __syncthreads();
    K2_STAMP(prof, 3);
    const int act = *s_act;
    constexpr int NWB = sizeof(Begin2) / sizeof(double);
    // (coh_bg: the launch rewrites Ctl2::bg from other blocks later on - stage3.hpp - so no copy of it may stay dirty in this die's L2)
    if ((act == 0 || act == 6) && writer && t < NWB) row_store(reinterpret_cast<double *>(&ctl->bg) + t, reinterpret_cast<const double *>(s_bg)[t], coh_bg);
    return act;
}

// stage begin as its own launch (certificate path: the solver is armed in DevState::sol[0]); 1 block
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(T1) k2_begin(DevState *st, Ctl2 *ctl, int n, Rows2 mrows, const double *sched, Records rec, int spec_expected = 0) {
    __shared__ Post2 s_po;
    __shared__ Begin2 s_bg;
    __shared__ double s_vt[V2_MAXV * RMUT * 4], s_tot[RMUT], s_sw[64];
    __shared__ int s_act;
    begin2_block<T1>(n, st, ctl, mrows, spec_expected, sched, rec, &s_po, &s_bg, s_vt, s_tot, s_sw, &s_act);
}
#endif

// ------------------------------------------------------------------------------------------------ certificate passes
// One pass over (loglh, old_loglh, weight) for the <= KC candidates of solver copy (p-1)&1 (see k_pass in kernels.hpp); the
// decision of pass p-1 is taken in the prologue of pass p by every block from the canonically totalled rows.
// s_flag: 0 go on, 2 out of passes (stall), < 0 error.
__device__ inline void solver_prologue2(DevState *st, const double *sched, const Rows2 &prev, int p, Solver *S, double *vt, double *tot,
                                        double *srt, int force_final, int *s_flag, int T) {
    const int t = threadIdx.x;
    constexpr int NW = sizeof(Solver) / sizeof(double);
    const Solver *src = &st->sol[p == 0 ? 0 : ((p - 1) & 1)];
    if (t < NW) reinterpret_cast<double *>(S)[t] = reinterpret_cast<const double *>(src)[t];
    if (t == 0) *s_flag = 0;
    __syncthreads();
    if (p == 0) return;
    const int mode = S->mode;
    if (mode == MODE_SCAN || mode == MODE_SECTION) {
        if (T == T1) reduce_rows<2 * KC, 2, T1>(prev, vt, tot); else reduce_rows<2 * KC, 2, TB>(prev, vt, tot);
        __shared__ int s_err;
        if (t == 0) s_err = 0;
        __syncthreads();
        if (t < 64) solver_decide_wave(*S, tot, sched, st->rp.n_phi, st->rp.phi_rtol, &s_err, srt);
        __syncthreads();
        if (t == 0) {
            if (s_err) { *s_flag = s_err; S->mode = MODE_IDLE; }
            else if (force_final && S->mode != MODE_FINAL) *s_flag = 2;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && t < NW && *s_flag != 2) reinterpret_cast<double *>(&st->sol[p & 1])[t] = reinterpret_cast<const double *>(S)[t];
}

#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(T1) k2_pass(CloudPtrs cl, DevState *st, Ctl2 *ctl, Geo2 g, int n, int p, Rows2 prev, const double *sched,
                                              double *rows_out) {
    constexpr int K = KC;
    __shared__ double red[(T1 / 64) * 2 * K];
    __shared__ double s_vt[V2_MAXV * 2 * K * 2], s_tot[2 * K], s_srt[K];
    __shared__ Solver S;
    __shared__ int s_flag;
    const int bstage = ctl->bg.stage, bfinal = ctl->bg.final;
    const double phi_prev = ctl->bg.phi_prev, esh = st->rp.pw == 0.0 ? ctl->bg.e_shift : 0.0;
    if (bstage != n || bfinal) return;
    solver_prologue2(st, sched, prev, p, &S, s_vt, s_tot, s_srt, 0, &s_flag, T1);
    if (s_flag) {                                         // an error (the objective is NaN at the walk's end: fzero's bracket error, helpers.jl:49-50)
        // reported HERE: the solver copy is idle from now on, so neither the passes behind this one nor k2_finish see the error again -
        // without the status the host would feed an idle search more passes for ever
        if (blockIdx.x == 0 && threadIdx.x == 0 && s_flag < 0) { ctl->status.err = s_flag; ctl->status.stage = n; ctl->status.code = 9; }
        return;
    }
    const int mode = S.mode;
    if (mode != MODE_SCAN && mode != MODE_SECTION) return;
    const int nv = S.n_valid, R = cl.R;
    const double *loglh = col(cl, 0, R - 5), *old = col(cl, 0, R - 3), *w = col(cl, 0, R - 1);
    double acc[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) acc[k] = 0.0;
    long long beg, end;
    vchunk(g, blockIdx.x / g.nb1, blockIdx.x % g.nb1, g.per1, beg, end);
    for (long long i = beg + threadIdx.x; i < end; i += T1) {
        const double l = loglh[i] - esh, o = old[i], wi = w[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k >= nv) continue;
            const double phi = S.cand[k];
            const double v = wi * exp((phi_prev - phi) * o + (phi - phi_prev) * l);      // always the prior_weight = 0 formula (quirk Q4)
            acc[k] += v;
            acc[K + k] += v * v;
        }
    }
    const double total = block_reduce_nw<2 * K, T1 / 64>(acc, red);
    if (threadIdx.x < 2 * K) rows_out[(long long)blockIdx.x * (2 * K) + threadIdx.x] = total;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->status.solver_passes += 1;
}
#endif

// decision of the last enqueued pass: ϕ_n certified -> Begin2 becomes final; otherwise the stage stalls (code 2) until the host
// enqueues more passes (which continue the same search from solver copy (P-1)&1 and its rows).  1 block.
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(T1) k2_finish(DevState *st, Ctl2 *ctl, int n, int P, Rows2 prev, const double *sched) {
    __shared__ double s_vt[V2_MAXV * 2 * KC * 2], s_tot[2 * KC], s_srt[KC];
    __shared__ Solver S;
    __shared__ int s_flag;
    if (ctl->bg.stage != n || ctl->bg.final) return;
    if (ctl->status.code == 9 && ctl->status.stage == n) return;       // a pass of this stage has reported an error
    solver_prologue2(st, sched, prev, P, &S, s_vt, s_tot, s_srt, 1, &s_flag, T1);
    if (threadIdx.x != 0) return;
    if (s_flag < 0) { ctl->status.err = s_flag; ctl->status.stage = n; ctl->status.code = 9; return; }
    if (s_flag == 2 || S.mode != MODE_FINAL) { ctl->status.stage = n; ctl->status.code = 2; return; }
    ctl->bg.phi_n = S.phi_n; ctl->bg.j = S.j; ctl->bg.phi_prop = S.phi_prop; ctl->bg.spec = 0;
    __threadfence();
    ctl->bg.final = 1;
}
#endif

// Random numbers of proposal t (mh_step * n_blocks + block) of one particle (src/mutation.jl:66,133, helpers.jl:87-100; RNG
// contract in DESIGN.md): the MH uniform of this decision, the mixture-component uniform and the block's normals.  Box-Muller is
// written stage by stage over the pairs so the independent log / sqrt / sincospi chains interleave.
template <int D>
__device__ inline void draw2(unsigned long long seed, unsigned long long pid, unsigned stage, unsigned t, int db, int debug, double &step_prob,
                             double &uc, double (&z)[D]) {
SMCMI_FP_CONTRACT
    double u_dummy, unext;
    if (t == 0) uniform_pair(seed, pid, stage, rng_tag(P_MUT, 0xFFFFFu, 0), step_prob, u_dummy);       // quirk Q3: drawn before the proposal
    else uniform_pair(seed, pid, stage, rng_tag(P_MUT, t - 1, 0), u_dummy, step_prob);
    uniform_pair(seed, pid, stage, rng_tag(P_MUT, t, 0), uc, unext);
    constexpr int NP2 = (D + 1) / 2;
    constexpr int GRPB = 3;                     // pairs interleaved at a time (more raises register pressure)
    double ua[NP2], ub[NP2], rr[NP2], sn[NP2], cs[NP2];
#pragma unroll
    for (int q = 0; q < NP2; ++q) {
        ua[q] = 0.5; ub[q] = 0.0;
        if (2 * q < db) uniform_pair(seed, pid, stage, rng_tag(P_MUT, t, 1 + q), ua[q], ub[q]);
    }
#pragma unroll
    for (int g0 = 0; g0 < NP2; g0 += GRPB) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = g0; q < g0 + GRPB && q < NP2; ++q) rr[q] = bx_neg2log(ua[q]);
#pragma unroll
        for (int q = g0; q < g0 + GRPB && q < NP2; ++q) rr[q] = bx_sqrt(rr[q]);
#pragma unroll
        for (int q = g0; q < g0 + GRPB && q < NP2; ++q) bx_sincos2pi(ub[q], &sn[q], &cs[q]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NP2; ++q) {
        z[2 * q] = (2 * q < db) ? rr[q] * cs[q] : 0.0;
        if (2 * q + 1 < D) z[2 * q + 1] = (2 * q + 1 < db) ? rr[q] * sn[q] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < D; ++e) asm volatile("" : "+v"(z[e]));   // materialise the normals here
}

// ------------------------------------------------------------------------------------------------ random numbers drawn ahead
// The mutation's draws depend only on (seed, particle id, stage, proposal index).  K1 keeps <= 64 of the 256 CUs busy while the
// cloud is small, so the same launch carries extra blocks (behind the correction blocks in dispatch order) that draw the stage's
// random numbers into zbuf on the idle CUs; K2 then loads D + 2 coalesced values per proposal instead of running Philox +
// Box-Muller (~40 % of its arithmetic).  Layout: zbuf[(t ZS + slot) n + i], t = mh_step n_blocks + block, ZS = D + 2 slots:
// MH uniform, mixture uniform, D normals.  Same expressions as draw2 -> same bits.
struct Rng2 {
    double *zbuf;            // null: disabled
    int t_lim;               // > 0: only the first t_lim proposals (mh_step * n_blocks + block) of every particle are drawn
    int n_steps, nb, nf;
    unsigned long long seed;
    long long gid0;
};
// Extra block `block` of `nblocks` draws the mutation-block chunks block, block + nblocks, ...: the host sizes nblocks so that
// correction blocks + extra blocks fit the 256 CUs at one block each (this kernel's register budget admits no more).
template <int D>
__device__ inline void rng2_block(const Geo2 &g, const Rng2 &ra, int n, int block, int nblocks, int debug) {
    const int sub = (ra.nf + ra.nb - 1) / ra.nb;
    auto particle = [&](long long i) {
        const unsigned long long pid = (unsigned long long)(ra.gid0 + i);
        for (int step = 0; step < ra.n_steps; ++step)
            for (int b = 0; b < ra.nb; ++b) {
                const unsigned t = (unsigned)(step * ra.nb + b);
                if (ra.t_lim > 0 && (int)t >= ra.t_lim) continue;
                const int db = (b < ra.nb - 1) ? sub : ra.nf - sub * (ra.nb - 1);
                double step_prob, uc, z[D];
                draw2<D>(ra.seed, pid, (unsigned)n, t, db, debug, step_prob, uc, z);
                double *zt = ra.zbuf + (long long)t * (D + 2) * g.n + i;
                zt[0] = step_prob;
                zt[g.n] = uc;
#pragma unroll
                for (int e = 0; e < D; ++e) zt[(long long)(2 + e) * g.n] = z[e];
            }
    };
    if (!g.inker) {           // large shards: the local particles dealt out block-width by block-width
        for (long long i = (long long)block * blockDim.x + threadIdx.x; i < g.n; i += (long long)nblocks * blockDim.x) particle(i);
        return;
    }
    for (int cb = block; cb < g.Vl * g.nb2; cb += nblocks) {
        long long beg, end;
        vchunk(g, cb / g.nb2, cb % g.nb2, g.t2, beg, end);
        for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) particle(i);
    }
}

// One correction block's row (ΣW̃, ΣW̃², the (D+1)(D+2)/2 augmented pair sums) from the threads' accumulators; csum (may be null)
// receives ΣW̃ as well (chunk sums of the selection scan).  red: T1 / 64 * 64 doubles of LDS.  All T1 threads call.
// A correction block's row: the NPF = (D+1)(D+2)/2 + 2 sums over the block's particles, in the canonical order - every sum is a 64-lane
// butterfly per wavefront (16 sums at a time: Butterfly<8, 32>, lane 4k ends with sum k of the chunk), then the wavefronts in order.
// Sixteen at a time, not sixty-four: the wide butterfly holds 128 registers on top of the accumulators, which a kernel that keeps its
// particle in registers across stages (stage3.hpp) does not have - 60 scratch reloads inside its stage loop came from there.  All
// chunks go to LDS before ONE barrier (the 64-wide form took two barriers per chunk).  red: NW * cm_row_ld(NPF) doubles.
constexpr int CMW = 16;
constexpr int cm_row_ld(int npf) { return (npf + CMW - 1) / CMW * CMW; }
template <int NPF, int NW, class GET, class ST>
__device__ inline void cm_row_chunks(double *red, GET get, ST store) {
    constexpr int NC = (NPF + CMW - 1) / CMW, LD = NC * CMW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    static_for<NC>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        double a[CMW];
        get(C, a);
        Butterfly<CMW / 2, 32>::run(a, lane);
        if ((lane & 3) == 0) red[wave * LD + c * CMW + (lane >> 2)] = a[0];
    });
    __syncthreads();
    double tot = 0.0;
    if ((int)threadIdx.x < NPF) {
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w * LD + threadIdx.x];
    }
    __syncthreads();
    // (the stores come AFTER the barrier: a barrier waits for the vector-memory operations in front of it, and these are write-through)
    if ((int)threadIdx.x < NPF) store((int)threadIdx.x, tot);
}
template <int D, class ST>
__device__ inline void k2_cm_row_f(double (&acc)[((D + 1) * (D + 2) / 2 + 2 + 63) / 64 * 64], double *red, ST store) {
    constexpr int NPF = (D + 1) * (D + 2) / 2 + 2;
    cm_row_chunks<NPF, T1 / 64>(red, [&](auto C, double (&a)[CMW]) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
#pragma unroll
        for (int q = 0; q < CMW; ++q) a[q] = (c * CMW + q < NPF) ? acc[c * CMW + q] : 0.0;
    }, store);
}
template <int D>
__device__ inline void k2_cm_row(double (&acc)[((D + 1) * (D + 2) / 2 + 2 + 63) / 64 * 64], double *red, double *out, double *csum, bool coh) {
    k2_cm_row_f<D>(acc, red, [&](int idx, double v) {
        row_store(out + idx, v, coh);
        if (idx == 0 && csum) *csum = v;
    });
}
// One particle's incremental weight (src/smc_main.jl:401-409), W̃ = W w̃, and x̃ = (1, θ - sh).  Returns W̃; *inc_out = w̃.
template <int D, class XF>
__device__ inline double k2_cm_weight(XF xf, const double *sh, double loglh, double old, double wi, double esh, double phi, double phi_prev, double pw,
                                      double logp_old, double (&xx)[D + 1], double *inc_out) {
    const double l = loglh - esh, o = old;
    xx[0] = 1.0;
#pragma unroll
    for (int a = 0; a < D; ++a) xx[a + 1] = xf(a) - sh[a];
    double inc;
    if (pw == 0.0) inc = exp((phi_prev - phi) * o + (phi - phi_prev) * l);
    else if (pw == 1.0) inc = exp((phi - phi_prev) * l);
    else inc = exp((phi_prev - phi) * log(exp(o - logp_old + log(1.0 - pw)) + pw) + (phi - phi_prev) * l);
    *inc_out = inc;
    return wi * inc;
}
// ... and its contribution to a correction block's accumulators: the sums of weighted_mean / weighted_cov about `sh`
// (particle.jl:481-483, 526-529).
template <int D, class XF>
__device__ inline double k2_cm_particle(double (&acc)[((D + 1) * (D + 2) / 2 + 2 + 63) / 64 * 64], XF xf, const double *sh, double loglh, double old, double wi,
                                        double esh, double phi, double phi_prev, double pw, double logp_old, double *inc_out) {
    constexpr int DA = D + 1;
    double xx[DA];
    const double v = k2_cm_weight<D>(xf, sh, loglh, old, wi, esh, phi, phi_prev, pw, logp_old, xx, inc_out);
    acc[0] += v;
    acc[1] += v * v;
    int q = 2;
#pragma unroll
    for (int a = 0; a < DA; ++a) {
        const double wx = v * xx[a];
#pragma unroll
        for (int b = a; b < DA; ++b) { acc[q] += wx * xx[b]; ++q; }
    }
    return v;
}
// The row of a block whose threads hold ONE particle each (stage3.hpp): the sixteen sums of a chunk are formed where the butterfly
// needs them - 0 + W̃ x̃_a x̃_b is the accumulator k2_cm_particle would hold, bit for bit - so no accumulator array is alive at all.
__host__ __device__ constexpr int cm_pair_a(int da, int idx) {      // sum idx >= 2 is the pair (a, b), a <= b, row-major
    int a = 0, k = idx - 2;
    while (k >= da - a) { k -= da - a; ++a; }
    return a;
}
__host__ __device__ constexpr int cm_pair_b(int da, int idx) {
    int a = 0, k = idx - 2;
    while (k >= da - a) { k -= da - a; ++a; }
    return a + k;
}
template <int D, int NW, class ST>
__device__ inline void k2_cm_row_one(double v, const double (&xx)[D + 1], bool live, double *red, ST store) {
    constexpr int DA = D + 1, NPF = DA * (DA + 1) / 2 + 2;
    cm_row_chunks<NPF, NW>(red, [&](auto C, double (&a)[CMW]) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        static_for<CMW>([&](auto Q) __attribute__((always_inline)) {
            constexpr int idx = c * CMW + decltype(Q)::value, q = decltype(Q)::value;
            double val = 0.0;
            if constexpr (idx == 0) val = v;
            else if constexpr (idx == 1) val = v * v;
            else if constexpr (idx < NPF) {
                constexpr int pa = cm_pair_a(DA, idx), pb = cm_pair_b(DA, idx);      // (constant-evaluated: register indices)
                const double wx = v * xx[pa];
                val = wx * xx[pb];
            }
            a[q] = live ? val : 0.0;
        });
    }, store);
}

// ------------------------------------------------------------------------------------------------ K1: correction + moments
// src/smc_main.jl:401-420 (incremental weights, update_weights!) fused with the sums of weighted_mean / weighted_cov
// (particle.jl:481-483, 526-529) of the corrected cloud, as k_correct_moments - but the unnormalised weights W̃ always go to the
// scratch column `wt` (the cloud's weight column is rewritten by K2 / the gather once the stage is decided), the block's ΣW̃
// also goes to csum[b] (chunk sums of the selection scan), and the stage-begin logic runs in the prologue (begin_done = 0).
// TAIL: the geometry hands its rows over through Tail2 (sharded runs, large clouds); the direct geometry's instantiation carries none of it
// Large shards (stage2b.hpp): ONE extra block behind the correction blocks - the only block of the launch that waits for anything - takes the
// hand-over of this launch's own rows (the V x m totals arrive in the handle's mailbox), decides, builds the proposal and leaves it in
// Prop2Glob for the mutation launch: k2_prepare without its launch, its kernel start and its launch boundary.
struct Prop2Glob;
struct Prep2Args {
    int enable;                   // the launch carries the helper block (block Vl * nb1; dynamic LDS = k2_lds_bytes)
    int nb, nf;                   // random blocks / free parameters
    unsigned long long seed;
    Rows2 cmrows;                 // this launch's correction totals as the mailbox delivers them
    Records rec;
    const ModelDev *md;
    Prop2Glob *out;
};
template <int D>
__device__ void k2_prepare_block(DevState *st, Ctl2 *ctl, const Prep2Args &pb, int n);

template <int D, bool TAIL>
__global__ void __launch_bounds__(T1) k2_correct(CloudPtrs cl, DevState *st, Ctl2 *ctl, Geo2 g, int n, int begin_done, int spec_expected,
                                                 Rows2 mrows, const double *sched, Records rec, double *rows_cm, double *csum, double *wt,
                                                 double *hist_w, long long hist_ld, Rng2 ra, Tail2 tail, Prep2Args pb, long long *prof = nullptr) {
    constexpr int DA = D + 1, NP = DA * (DA + 1) / 2, NPF = NP + 2;
    constexpr int NCH = (NPF + 63) / 64, NW = T1 / 64;
    if ((int)blockIdx.x >= g.Vl * g.nb1) {
        int xb = (int)blockIdx.x - g.Vl * g.nb1, nx = (int)gridDim.x - g.Vl * g.nb1;
        if constexpr (TAIL && D <= 10) {
            if (pb.enable) {
                if (xb == 0) { k2_prepare_block<D>(st, ctl, pb, n); return; }
                --xb; --nx;
            }
        }
        // the idle CUs draw the mutation's random numbers (Rng2)
        // (drawn whether or not the stage goes ahead: a stalled stage is redone with the same numbers)
        if (ra.zbuf && ctl->ps[(n - 1) & 1].stage == n - 1) rng2_block<D>(g, ra, n, xb, nx, 0);
        return;
    }
    __shared__ double red[NW * cm_row_ld(NPF)];
    __shared__ Post2 s_po;
    __shared__ Begin2 s_bg;
    __shared__ double s_vt[V2_MAXV * RMUT * 4], s_tot[RMUT], s_sw[64];
    __shared__ int s_act;
    K2_STAMP(prof, 0);
    const double pw = st->rp.pw, logp_old = st->rp.logp_old;
    const bool hist = st->rp.store_history && hist_w != nullptr;
    const int R = cl.R;
    const double *loglh = col(cl, 0, R - 5), *old = col(cl, 0, R - 3), *w = col(cl, 0, R - 1);
    long long beg, end;
    vchunk(g, blockIdx.x / g.nb1, blockIdx.x % g.nb1, g.per1, beg, end);
    if (!begin_done) {
        const int act = begin2_block<T1>(n, st, ctl, mrows, spec_expected, sched, rec, &s_po, &s_bg, s_vt, s_tot, s_sw, &s_act, prof);
        if (act != 0) return;
    } else {
        constexpr int NWB = sizeof(Begin2) / sizeof(double), NWP = sizeof(Post2) / sizeof(double);
        if (threadIdx.x < NWB) reinterpret_cast<double *>(&s_bg)[threadIdx.x] = reinterpret_cast<const double *>(&ctl->bg)[threadIdx.x];
        if (threadIdx.x < NWP) reinterpret_cast<double *>(&s_po)[threadIdx.x] = reinterpret_cast<const double *>(&ctl->ps[(n - 1) & 1])[threadIdx.x];
        __syncthreads();
        if (s_bg.stage != n || !s_bg.final || s_po.stage != n - 1) return;
    }
    const double phi = s_bg.phi_n, phi_prev = s_bg.phi_prev, esh = pw == 0.0 ? s_bg.e_shift : 0.0;
    const double *sh = s_po.shift;           // read from LDS where used (uniform address): twenty registers the accumulators need
    const double unshift = hist ? exp((phi - phi_prev) * esh) : 1.0;                      // history keeps the true exp(δ e)
    if constexpr (D <= 10) {
    double acc[NCH * 64];
#pragma unroll
    for (int q = 0; q < NCH * 64; ++q) acc[q] = 0.0;
    // (the geometry gives a thread at most two particles while the cloud is small; requesting the first one ahead of the prologue
    // was tried: the 68 accumulator pairs leave no registers to carry it, the spills inside the loop cost more than the round trip)
    for (long long i = beg + threadIdx.x; i < end; i += T1) {
        double inc;
        const double v = k2_cm_particle<D>(acc, [&](int a) { return col(cl, 0, a)[i]; }, sh, loglh[i], old[i], w[i], esh, phi, phi_prev, pw, logp_old, &inc);
        wt[i] = v;
        if (hist) hist_w[(long long)(n - 1) * hist_ld + i] = inc * unshift;
    }
    K2_STAMP(prof, 4);
    k2_cm_row<D>(acc, red, rows_cm + (long long)blockIdx.x * pad2(NPF), csum + blockIdx.x, TAIL && tail.tick != nullptr);
    } else {
        // n_para > 10: 100 - 155 sums do not fit a thread's registers as accumulators.  The geometry gives every thread ONE particle of the
        // row (per1 = 512: make_geo2), and the row's sums are formed sixteen at a time where the butterflies need them (k2_cm_row_one, the
        // form the persistent segment kernel uses): nothing is accumulated, 0 + W̃ x̃_a x̃_b is what an accumulator would hold.
        const long long i = beg + threadIdx.x;
        const bool live = i < end;
        const long long il = live ? i : (end > beg ? end - 1 : 0);
        double xx[D + 1], inc;
        double v = k2_cm_weight<D>([&](int a) { return col(cl, 0, a)[il]; }, sh, loglh[il], old[il], w[il], esh, phi, phi_prev, pw, logp_old, xx, &inc);
        if (live) {
            wt[i] = v;
            if (hist) hist_w[(long long)(n - 1) * hist_ld + i] = inc * unshift;
        } else v = 0.0;
        K2_STAMP(prof, 4);
        double *out = rows_cm + (long long)blockIdx.x * pad2(NPF);
        const bool coh = TAIL && tail.tick != nullptr;
        k2_cm_row_one<D, NW>(v, xx, live, red, [&](int idx, double val) {
            row_store(out + idx, val, coh);
            if (idx == 0) csum[blockIdx.x] = val;
        });
    }
    K2_STAMP(prof, 5);
    if constexpr (TAIL) tail_reduce<T1, (pad2(NPF) > 72 ? 160 : 72)>(tail, rows_cm, (int)blockIdx.x / g.nb1, g.nb1, pad2(NPF), -1, 0);
    // off the critical path: the step-size multiplier K2 applies (two exponentials) - nobody in this launch reads it (a helper block
    // computes its own: the mutation launch behind it takes everything from Prop2Glob)
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(TAIL && pb.enable)) {
        const double a = s_bg.accept, tg = st->rp.target;
        ctl->bg.cfac = 0.95 + 0.10 * exp(16.0 * (a - tg)) / (1.0 + exp(16.0 * (a - tg)));
    }
}

// ------------------------------------------------------------------------------------------------ post-correction decision
// ESS of the corrected weights, verification of a predicted ϕ_n (see k_prepare_mutation), selection decision
// (src/smc_main.jl:427-435).  Same inputs -> same result in every block / kernel that calls it.
// 0: go, no resampling; 1: go, resample; 4: prediction not verified; < 0: error code
__device__ inline int decide2(const Begin2 &bg, double threshold, double phi_rtol, double s1, double s2, double *ess_out) {
    const double ess = s1 * s1 / s2;
    *ess_out = ess;
    if (bg.spec) {
        bool verified;
        if (bg.phi_n < 1.0) verified = fabs(ess - bg.ess_bar) <= fabs(bg.gprime) * fmax(phi_rtol, SPEC_VERIFY_RTOL) * bg.phi_n;
        else verified = ess >= bg.ess_bar * (1.0 - 1e-13);
        if (!verified) return 4;
    }
    // check_nan_ess, helpers.jl:270-305 (sums that overflowed count as well: under a lagged energy shift - Begin2::e_seen - a cloud whose
    // largest energy grew by more than ~350 / (ϕ_n - ϕ_{n-1}) in ONE mutation overflows Σ W̃²; the host then redoes the stage with the exact shift)
    if (isnan(ess) || fabs(s1) > 1.7e308 || fabs(s2) > 1.7e308) return SMCMI_ERR_NAN_ESS;
    return ess < threshold ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ selection: scan
// Inclusive scan of W̃ / ΣW̃ (cumsum(weights ./ sum(weights)), src/resample.jl:29,47) over the WHOLE cloud (sharded runs hand in the
// all-gathered W̃ and chunk sums) in the correction's chunks: one block per chunk, chunk offsets = running sum of the chunk sums
// in chunk order.  Does nothing unless this stage resamples (the decision is re-derived from the correction rows).
// The pieces are functions of a block of TS = 512 threads with one value per thread - what a worker of a persistent segment (stage3.hpp)
// holds - so that a segment that resamples without leaving writes the cum column this kernel writes, bit for bit.
constexpr int TS = 512;
// exclusive prefix of the chunk sums cs(b), b < nchunks <= 1024, in chunk order -> s_off[b]: 256 threads x 4 chunks each, then a sequential
// carry.  All threads of the block call; scratch: 256 doubles, s_off: nchunks doubles of LDS; ends with a barrier.
template <class CS>
__device__ inline void sel_chunk_offsets(CS cs, int nchunks, double *scratch, double *s_off) {
    const int t = threadIdx.x;
    double loc[4], run = 0.0;
    if (t < 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int b = t * 4 + q; loc[q] = run; run += (b < nchunks) ? cs(b) : 0.0; }
        scratch[t] = run;
    }
    __syncthreads();
    // (threads beyond the last chunk hold 0: the carry past them is never read)
    if (t == 0) { double carry = 0.0; const int nq = (nchunks + 3) / 4; for (int q = 0; q < nq; ++q) { const double x = scratch[q]; scratch[q] = carry; carry += x; } }
    __syncthreads();
    if (t < 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (t * 4 + q < nchunks) s_off[t * 4 + q] = scratch[t] + loc[q];
    }
    __syncthreads();
}
// inclusive scan of one tile of TS values (thread t holds value t; 0 beyond the end of the data): inside a wavefront by shuffles
// (x_t += x_{t - off}, off = 1 .. 32), the wavefronts' totals added up in ascending order.  Returns the inclusive sum at this thread,
// *tile_total the tile's total.  s_w: TS / 64 doubles of LDS; all TS threads call; two barriers.
__device__ inline double sel_tile_scan(double w, double *s_w, double *tile_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(x, off, 64);
        x = lane >= off ? x + y : x;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    double ex = 0.0, tot = 0.0;
#pragma unroll
    for (int k = 0; k < TS / 64; ++k) { const double v = s_w[k]; ex = k < wave ? ex + v : ex; tot += v; }
    __syncthreads();
    *tile_total = tot;
    return ex + x;
}
// c_begin / c_end / i_off: scan the chunks [c_begin, c_end) only, reading and writing at (global index - i_off) - a handle of a sharded
// run scans ITS particles' weights into its local cum column from the all-gathered chunk sums (the offsets, the total and therefore
// the values are those of a scan over the whole cloud).
#ifndef SMCMI_INST_UNIT
static __global__ void __launch_bounds__(TS) k2_scan(Ctl2 *ctl, const DevState *st, Geo2 g, int n, Rows2 cmrows, const double *wt_full,
                                              const double *csum_full, double *cum, int c_begin = 0, int c_end = -1, long long i_off = 0) {
    __shared__ double s_vt[V2_MAXV * 2 * 8], s_tot[2], scratch[256], s_off[1024], s_w[TS / 64];
    __shared__ Begin2 s_bg;
    constexpr int NWB = sizeof(Begin2) / sizeof(double);
    if (threadIdx.x < NWB) reinterpret_cast<double *>(&s_bg)[threadIdx.x] = reinterpret_cast<const double *>(&ctl->bg)[threadIdx.x];
    __syncthreads();
    if (s_bg.stage != n || !s_bg.final || ctl->ps[(n - 1) & 1].stage != n - 1) return;
    reduce_rows<2, 8, TS>(cmrows, s_vt, s_tot);
    double ess;
    if (decide2(s_bg, st->rp.threshold, st->rp.phi_rtol, s_tot[0], s_tot[1], &ess) != 1) return;
    const int nchunks = g.V * g.nb1, t = threadIdx.x;
    sel_chunk_offsets([&](int b) { return csum_full[b]; }, nchunks, scratch, s_off);
    const double total = s_tot[0];
    wt_full -= i_off; cum -= i_off;
    for (int c = c_begin + (int)blockIdx.x; c < (c_end < 0 ? nchunks : c_end); c += gridDim.x) {
        const int v = c / g.nb1, r = c % g.nb1;
        const long long v_beg0 = (long long)v * g.nv, v_beg = v_beg0 < g.N ? v_beg0 : g.N, v_end = v_beg + g.nv < g.N ? v_beg + g.nv : g.N;
        long long beg = v_beg + (long long)r * g.per1, end = beg + g.per1 < v_end ? beg + g.per1 : v_end;
        if (beg > v_end) beg = v_end;
        double carry = s_off[c];
        for (long long base = beg; base < end; base += TS) {
            const long long i = base + t;
            double tt;
            const double incl = sel_tile_scan(i < end ? wt_full[i] : 0.0, s_w, &tt);
            if (i < end) cum[i] = (carry + incl) / total;
            carry += tt;
        }
    }
}
#endif

// Systematic resampling, sharded, owner side: which of THIS handle's rows does handle r need?  The ancestors of r's
// slots k0 .. k1 (thresholds t = (k + u) / N) are the first j with cum[j] > t; among this handle's rows they lie between its first
// row with cum > t(k0) and its first row with cum > t(k1) (its last row if there is none) - a superset by at most one row, from this
// handle's own cum column alone.  out[2 r], out[2 r + 1] = global indices of that range, -1 / -1 if r needs none of the rows;
// out[2 world] = 1 if the stage resamples, else 0 (nothing else is valid then).
#ifndef SMCMI_INST_UNIT
static __global__ void k2_owner_ranges(Ctl2 *ctl, const DevState *st, int n, Rows2 cmrows, const double *cum_local, long long N, long long n_local,
                                       long long gid0, int world, unsigned long long seed, long long *out, const double *csum_full, int c_first) {
    __shared__ double s_vt[V2_MAXV * 2 * 8], s_tot[2];
    const int r = threadIdx.x;
    const Begin2 bg = ctl->bg;
    bool go = bg.stage == n && bg.final && ctl->ps[(n - 1) & 1].stage == n - 1;
    if (cmrows.mb && !go) { if (r == 0) out[2 * world] = 0; return; }       // (mailbox: no waiting for rows of a stage that does not run)
    reduce_rows<2, 8, 64>(cmrows, s_vt, s_tot);
    double ess;
    if (go) go = decide2(bg, st->rp.threshold, st->rp.phi_rtol, s_tot[0], s_tot[1], &ess) == 1;
    if (r == 0) out[2 * world] = go ? 1 : 0;
    if (!go || r >= world) return;
    double ua, ub;
    uniform_pair(seed, 0ull, (unsigned)n, rng_tag(P_RES, 0, 0), ua, ub);
    long long res[2];
    for (int e = 0; e < 2; ++e) {
        const long long slot = e == 0 ? (long long)r * n_local : (long long)(r + 1) * n_local - 1;
        const double thr = ((double)slot + ua) / (double)N;
        long long lo = 0, hi = n_local;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (cum_local[mid] > thr) hi = mid; else lo = mid + 1;
        }
        res[e] = lo;                                        // first own row with cum > thr (n_local: none)
    }
    // cumulative weight in front of this handle's first row (to a few ulps: the chunk sums are block reductions, the cum column a
    // scan): if even that - less a safety margin - exceeds t(k1), r's last ancestor lies before this handle's rows
    double before = 0.0;
    for (int c = 0; c < c_first; ++c) before += csum_full[c];
    const double thr1 = ((double)((long long)(r + 1) * n_local - 1) + ua) / (double)N;
    const bool behind = (before / s_tot[0]) * (1.0 - 1e-9) > thr1;
    const bool none = behind || res[0] >= n_local;          // or every own cum <= t(k0): all of r's ancestors lie behind this handle's rows
    out[2 * r] = none ? -1 : gid0 + res[0];
    out[2 * r + 1] = none ? -1 : gid0 + (res[1] < n_local ? res[1] : n_local - 1);
}
#endif

// ------------------------------------------------------------------------------------------------ selection: gather + moments
// Output slot k: ancestor = first j with cum[j] > threshold (src/resample.jl:33-70; fall-through clamps to the last index), its
// R-1 columns are copied into cloud buffer 1 (K2 reads the resampled cloud from there and writes buffer 0), and the moments of
// the resampled cloud (all weights 1; smc_main.jl:440-446, 457-465) are accumulated on the way -> one row of pair sums per block.
// full: rows come from the per-handle [R][full_shard_n] shard buffers of an exchange (sharded runs), else from buffer 0.
template <int D>
__global__ void __launch_bounds__(TB) k2_gather_wide(CloudPtrs cl, Ctl2 *ctl, const DevState *st, Geo2 g, int n, Rows2 cmrows, const double *cum,
                                                int method, unsigned long long seed, long long gid0, long long *anc, const double *full,
                                                long long full_shard_n, double *rows_gm, long long s_lo = 0, long long s_hi = -1) {
    constexpr int DA = D + 1, NP = DA * (DA + 1) / 2, NCH = (NP + 63) / 64;
    __shared__ double red[(TB / 64) * 64];
    __shared__ double s_vt[V2_MAXV * 2 * 8], s_tot[2];
    __shared__ Begin2 s_bg;
    __shared__ double s_sh[D];
    constexpr int NWB = sizeof(Begin2) / sizeof(double);
    if (threadIdx.x < NWB) reinterpret_cast<double *>(&s_bg)[threadIdx.x] = reinterpret_cast<const double *>(&ctl->bg)[threadIdx.x];
    const Post2 &po = ctl->ps[(n - 1) & 1];
    const int pstage = po.stage;
    if (threadIdx.x < D) s_sh[threadIdx.x] = po.shift[threadIdx.x];
    __syncthreads();
    if (s_bg.stage != n || !s_bg.final || pstage != n - 1) return;
    reduce_rows<2, 8, TB>(cmrows, s_vt, s_tot);
    double ess;
    if (decide2(s_bg, st->rp.threshold, st->rp.phi_rtol, s_tot[0], s_tot[1], &ess) != 1) return;
    const int R = cl.R;
    const long long N = g.N;
    double acc[NCH * 64];
#pragma unroll
    for (int q = 0; q < NCH * 64; ++q) acc[q] = 0.0;
    long long beg, end;
    vchunk(g, blockIdx.x / g.nbg, blockIdx.x % g.nbg, g.perg, beg, end);
    double u_sys = 0.0, ub;
    if (method != SMCMI_RESAMPLE_MULTINOMIAL) uniform_pair(seed, 0ull, (unsigned)n, rng_tag(P_RES, 0, 0), u_sys, ub);
    // (s_lo .. s_hi: the global rows whose cum this handle holds - a sharded run exchanges only its slots' ancestor range)
    const long long s_end = s_hi < 0 ? N : s_hi + 1;
    auto gather_row = [&](long long a, long long k) {
        if (anc) anc[k] = a;
        const double *from = cl.buf[0];
        long long ldf = cl.n, a_row = a;
        if (full) {
            from = full + (a / full_shard_n) * (long long)R * full_shard_n;
            ldf = full_shard_n;
            a_row = a % full_shard_n;
        }
        double row[D + 3];
#pragma unroll
        for (int q = 0; q < D + 3; ++q) row[q] = from[(long long)q * ldf + a_row];
#pragma unroll
        for (int q = 0; q < D + 3; ++q) col(cl, 1, q)[k] = row[q];
        col(cl, 1, D + 3)[k] = from[(long long)(D + 3) * ldf + a_row];
        double xx[DA];
        xx[0] = 1.0;
#pragma unroll
        for (int q = 0; q < D; ++q) xx[q + 1] = row[q] - s_sh[q];
        int p = 0;
#pragma unroll
        for (int a2 = 0; a2 < DA; ++a2) {
#pragma unroll
            for (int b = a2; b < DA; ++b) { acc[p] += xx[a2] * xx[b]; ++p; }
        }
    };
    // Systematic resampling: the thresholds of consecutive slots ascend, so a tile of TB slots descends from one contiguous range of rows.
    // A binary search over cum in global memory is 17 dependent loads per slot (~10 of this kernel's 16 µs at N = 1e5); instead the block
    // keeps cum at the ends of 512-row chunks in LDS (one round of loads), finds the chunks of the tile's first and last threshold there,
    // stages the rows in between (one more round) and searches in LDS: the same comparisons on the same values, hence the same
    // ancestors.  Ranges that do not fit (a cloud about to collapse can spread a tile over many rows of tiny weight) and multinomial
    // resampling keep the search in global memory.
    constexpr int GCH = 512, G_NC = 2048, G_CAP = 2048;
    __shared__ double s_ce[G_NC], s_cw[G_CAP];
    __shared__ long long s_r0, s_r1;
    const long long span = s_end - s_lo;
    const int nc = (int)((span + GCH - 1) / GCH);
    const bool staged = method != SMCMI_RESAMPLE_MULTINOMIAL && span > 0 && nc <= G_NC;       // (block-uniform)
    if (staged) {
        for (int c = threadIdx.x; c < nc; c += TB) {
            const long long last = s_lo + (long long)(c + 1) * GCH - 1;
            s_ce[c] = cum[last < s_end ? last : s_end - 1];
        }
        __syncthreads();
    }
    const long long n_tiles = (end - beg + TB - 1) / TB;
    for (long long tile = 0; tile < n_tiles; ++tile) {
        const long long k = beg + tile * TB + threadIdx.x;
        const bool live = k < end;
        const long long slot = gid0 + (live ? k : end - 1);                  // (a dead thread stands for the tile's last live slot)
        double ua;
        if (method == SMCMI_RESAMPLE_MULTINOMIAL) uniform_pair(seed, (unsigned long long)slot, (unsigned)n, rng_tag(P_RES, 0, 0), ua, ub);
        else ua = ((double)slot + u_sys) / (double)N;                       // (i - 1 + offset) / n_parts
        if (staged) {
            if (threadIdx.x == 0 || threadIdx.x == TB - 1) {               // the tile's first / last threshold -> first chunk whose end exceeds it
                int lo = 0, hi = nc;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_ce[mid] > ua) hi = mid; else lo = mid + 1; }
                // rows below chunk `lo` end at or below the threshold; chunk `lo` (if any) holds a row above it
                if (threadIdx.x == 0) s_r0 = s_lo + (long long)lo * GCH;
                else s_r1 = lo < nc ? s_lo + (long long)(lo + 1) * GCH : s_end;
            }
            __syncthreads();
            const long long r0 = s_r0 < s_end ? s_r0 : s_end, r1 = s_r1 < s_end ? s_r1 : s_end;
            const bool in_lds = r1 - r0 <= G_CAP;                            // (block-uniform)
            if (in_lds)
                for (long long j = r0 + threadIdx.x; j < r1; j += TB) s_cw[j - r0] = cum[j];
            __syncthreads();
            if (in_lds) {
                // every slot of the tile has its answer in [r0, r1] (r1 itself = "none": clamped below like the full search)
                int lo = 0, hi = (int)(r1 - r0);
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cw[mid] > ua) hi = mid; else lo = mid + 1; }
                const long long lg = r0 + lo;
                if (live) gather_row(lg < s_end ? lg : s_end - 1, k);
                continue;
            }
        }
        if (!live) continue;
        long long lo = s_lo, hi = s_end;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (cum[mid] > ua) hi = mid; else lo = mid + 1;
        }
        gather_row(lo < s_end ? lo : s_end - 1, k);
    }
    double *out = rows_gm + (long long)blockIdx.x * pad2(NP);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        double a64[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) a64[q] = acc[ch * 64 + q];
        const double tot = block_reduce_many<64>(a64, red);
        if (threadIdx.x < 64 && ch * 64 + (int)threadIdx.x < NP) out[ch * 64 + threadIdx.x] = tot;
    }
}

// The same for n_para <= 10 by blocks of TS = 512 threads, one output slot per thread and tile - built from functions a worker of a
// persistent segment (stage3.hpp: 512 particles, one per thread) calls on its own slots, so that a segment that resamples without
// leaving selects the ancestors and leaves the moment row this kernel leaves, bit for bit.
constexpr int SEL_GCH = 512, SEL_NC = 2048, SEL_CAP = 4096;
// threshold of global output slot `slot` (src/resample.jl:33-70)
__device__ inline double sel_threshold(int method, unsigned long long seed, long long slot, int n, double u_sys, long long N) {
    if (method == SMCMI_RESAMPLE_MULTINOMIAL) {
        double ua, ub;
        uniform_pair(seed, (unsigned long long)slot, (unsigned)n, rng_tag(P_RES, 0, 0), ua, ub);
        return ua;
    }
    return ((double)slot + u_sys) / (double)N;                       // (i - 1 + offset) / n_parts
}
// cum at the ends of the SEL_GCH-row chunks of [s_lo, s_end) -> s_ce[0, nc) (all threads call; ends with a barrier)
template <class LDC>
__device__ inline void sel_chunk_ends(LDC ldcum, long long s_lo, long long s_end, int nc, double *s_ce) {
    for (int c = threadIdx.x; c < nc; c += TS) {
        const long long last = s_lo + (long long)(c + 1) * SEL_GCH - 1;
        s_ce[c] = ldcum(last < s_end ? last : s_end - 1);
    }
    __syncthreads();
}
// Ancestor of this thread's threshold ua = first row j of [s_lo, s_end) with cum[j] > ua, clamped to the last row.  Systematic resampling:
// the thresholds of a tile of TS consecutive slots ascend (a dead thread carries the tile's last live threshold), so the tile descends from
// one contiguous range of rows: threads 0 and TS - 1 find its chunks in s_ce, the rows in between are staged in s_cw (cap doubles) and
// searched there - the same comparisons on the same values as a search over the whole column, which is what ranges that do not fit and
// multinomial resampling (staged = false) get.  All TS threads call; s_r: two long longs of LDS.
template <class LDC>
__device__ inline long long sel_search_tile(double ua, bool staged, int nc, const double *s_ce, double *s_cw, long long *s_r, LDC ldcum, long long s_lo, long long s_end,
                                            int cap = SEL_CAP) {
    if (staged) {
        if (threadIdx.x == 0 || threadIdx.x == TS - 1) {
            int lo = 0, hi = nc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_ce[mid] > ua) hi = mid; else lo = mid + 1; }
            // rows below chunk `lo` end at or below the threshold; chunk `lo` (if any) holds a row above it
            if (threadIdx.x == 0) s_r[0] = s_lo + (long long)lo * SEL_GCH;
            else s_r[1] = lo < nc ? s_lo + (long long)(lo + 1) * SEL_GCH : s_end;
        }
        __syncthreads();
        const long long r0 = s_r[0] < s_end ? s_r[0] : s_end, r1 = s_r[1] < s_end ? s_r[1] : s_end;
        const bool in_lds = r1 - r0 <= cap;                            // (block-uniform; whether a range is staged changes no result)
        if (in_lds)
            for (long long j = r0 + threadIdx.x; j < r1; j += TS) s_cw[j - r0] = ldcum(j);
        __syncthreads();
        if (in_lds) {
            int lo = 0, hi = (int)(r1 - r0);
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cw[mid] > ua) hi = mid; else lo = mid + 1; }
            const long long lg = r0 + lo;
            return lg < s_end ? lg : s_end - 1;
        }
    }
    long long lo = s_lo, hi = s_end;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (ldcum(mid) > ua) hi = mid; else lo = mid + 1;
    }
    return lo < s_end ? lo : s_end - 1;
}
// the moment row of a block of TS threads from per-thread pair sums acc[p] = Σ x̃_a x̃_b over the thread's slots, p < NP, as the correction
// rows are formed (cm_row_chunks: sixteen sums at a time through the butterflies) -> store(p, total)
template <int D, class ST>
__device__ inline void sel_moment_row(const double (&acc)[(D + 1) * (D + 2) / 2], double *red, ST store) {
    constexpr int NP = (D + 1) * (D + 2) / 2;
    cm_row_chunks<NP, TS / 64>(red, [&](auto C, double (&a)[CMW]) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
#pragma unroll
        for (int q = 0; q < CMW; ++q) a[q] = (c * CMW + q < NP) ? acc[(c * CMW + q < NP) ? c * CMW + q : 0] : 0.0;
    }, store);
}
template <int D>
__global__ void __launch_bounds__(TS) k2_gather(CloudPtrs cl, Ctl2 *ctl, const DevState *st, Geo2 g, int n, Rows2 cmrows, const double *cum,
                                                int method, unsigned long long seed, long long gid0, long long *anc, const double *full,
                                                long long full_shard_n, double *rows_gm, long long s_lo = 0, long long s_hi = -1) {
    constexpr int DA = D + 1, NP = DA * (DA + 1) / 2;
    __shared__ double red[(TS / 64) * cm_row_ld(NP)];
    __shared__ double s_vt[V2_MAXV * 2 * 8], s_tot[2];
    __shared__ Begin2 s_bg;
    __shared__ double s_sh[D];
    __shared__ double s_ce[SEL_NC], s_cw[SEL_CAP];
    __shared__ long long s_r[2];
    constexpr int NWB = sizeof(Begin2) / sizeof(double);
    if (threadIdx.x < NWB) reinterpret_cast<double *>(&s_bg)[threadIdx.x] = reinterpret_cast<const double *>(&ctl->bg)[threadIdx.x];
    const Post2 &po = ctl->ps[(n - 1) & 1];
    const int pstage = po.stage;
    if (threadIdx.x < D) s_sh[threadIdx.x] = po.shift[threadIdx.x];
    __syncthreads();
    if (s_bg.stage != n || !s_bg.final || pstage != n - 1) return;
    reduce_rows<2, 8, TS>(cmrows, s_vt, s_tot);
    double ess;
    if (decide2(s_bg, st->rp.threshold, st->rp.phi_rtol, s_tot[0], s_tot[1], &ess) != 1) return;
    const int R = cl.R;
    const long long N = g.N;
    double acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = 0.0;
    long long beg, end;
    vchunk(g, blockIdx.x / g.nbg, blockIdx.x % g.nbg, g.perg, beg, end);
    double u_sys = 0.0, ub;
    if (method != SMCMI_RESAMPLE_MULTINOMIAL) uniform_pair(seed, 0ull, (unsigned)n, rng_tag(P_RES, 0, 0), u_sys, ub);
    // (s_lo .. s_hi: the global rows whose cum this handle holds - a sharded run exchanges only its slots' ancestor range)
    const long long s_end = s_hi < 0 ? N : s_hi + 1;
    const long long span = s_end - s_lo;
    const int nc = (int)((span + SEL_GCH - 1) / SEL_GCH);
    const bool staged = method != SMCMI_RESAMPLE_MULTINOMIAL && span > 0 && nc <= SEL_NC;       // (block-uniform)
    auto ldcum = [&](long long j) { return cum[j]; };
    if (staged) sel_chunk_ends(ldcum, s_lo, s_end, nc, s_ce);
    const long long n_tiles = (end - beg + TS - 1) / TS;
    for (long long tile = 0; tile < n_tiles; ++tile) {
        const long long k = beg + tile * TS + threadIdx.x;
        const bool live = k < end;
        const long long slot = gid0 + (live ? k : end - 1);                  // (a dead thread stands for the tile's last live slot)
        const double ua = sel_threshold(method, seed, slot, n, u_sys, N);
        const long long a = sel_search_tile(ua, staged, nc, s_ce, s_cw, s_r, ldcum, s_lo, s_end);
        if (!live) continue;
        if (anc) anc[k] = a;
        const double *from = cl.buf[0];
        long long ldf = cl.n, a_row = a;
        if (full) {
            from = full + (a / full_shard_n) * (long long)R * full_shard_n;
            ldf = full_shard_n;
            a_row = a % full_shard_n;
        }
        double row[D + 3];
#pragma unroll
        for (int q = 0; q < D + 3; ++q) row[q] = from[(long long)q * ldf + a_row];
#pragma unroll
        for (int q = 0; q < D + 3; ++q) col(cl, 1, q)[k] = row[q];
        col(cl, 1, D + 3)[k] = from[(long long)(D + 3) * ldf + a_row];
        double xx[DA];
        xx[0] = 1.0;
#pragma unroll
        for (int q = 0; q < D; ++q) xx[q + 1] = row[q] - s_sh[q];
        int p = 0;
#pragma unroll
        for (int a2 = 0; a2 < DA; ++a2) {
#pragma unroll
            for (int b2 = a2; b2 < DA; ++b2) { acc[p] += xx[a2] * xx[b2]; ++p; }
        }
    }
    double *out = rows_gm + (long long)blockIdx.x * pad2(NP);
    sel_moment_row<D>(acc, red, [&](int idx, double v) { out[idx] = v; });
}

// ------------------------------------------------------------------------------------------------ K2: decision + proposal + mutation
// Decision and proposal of one stage as ONE block computed them (k2_prepare): large clouds run many rounds of mutation blocks per
// CU, and a prologue every block repeats (row totals, covariance, Cholesky: ~10 µs of latency) then costs more than a launch.
struct Prop2Glob {
    int stage, go, rs, pad_;
    double s1, phi_n, e_center;
    double Lraw[100], logdet[10], mub[10], sdd[10], sdn[10];
    int ball[10], bptr[12], loff[10];
};

struct Mut2Args {
    unsigned long long seed;
    long long gid0;
    int n;                     // stage index of this launch
    int sel_enqueued;          // k2_scan / k2_gather were enqueued in front of this launch
    int adaptive;              // leave energy power sums for the next stage's ϕ predictor
    Rows2 cmrows, gmrows;
    Prop2Glob *pre;            // non-null: the stage's decision / proposal were computed by k2_prepare - load them
    LikDev lik[2];             // the model's likelihood descriptors (copies of ModelDev::lik: kernel arguments arrive in SGPRs, a
                               // descriptor read from memory would put two dependent round trips in front of the data loads)
    int n_steps, store_history, has_other;
    double alpha, n_parts;
    const double *zbuf;        // random numbers drawn ahead by K1's extra blocks (Rng2 layout) or null
    int z_ahead;               // k2b_mutate only (stage2b.hpp): the first z_ahead proposals of every particle come from zbuf, the rest are drawn
    const double *wt;          // unnormalised weights W̃ of the correction
    double *rows_mut;          // [blocks][RMUT]
    Tail2 tail;                // the last block of a virtual shard totals its mutation rows (sharded runs, large clouds)
    double *hist_W;
    long long hist_ld;
    Records rec;
    int debug;
    long long *prof;           // development only (K2_STAMP)
};

// Post-correction bookkeeping of stage n (src/smc_main.jl:427-455; see post_write in kernels.hpp); one thread per block
// computes it into LDS, block 0 stores it and the per-stage records.
__device__ inline void post2(int n, const Begin2 &bg, const Post2 &po, const RunParams &rp, double s1, double s2, double ess, int rs, Post2 *out) {
    const double a = bg.accept;
    Post2 p = po;
    p.stage = n; p.j = bg.j; p.resampled_last = rs; p.do_resample = rs; p.resamples = po.resamples + rs; p.fold_valid = 1;
    p.phi_n = bg.phi_n; p.phi_prop = bg.phi_prop; p.ess = ess; p.sumw = s1; p.sumw2 = s2;
    const double dlz = (bg.phi_n - bg.phi_prev) * (rp.pw == 0.0 ? bg.e_shift : 0.0);      // log of the common factor the shifted weights left out
    p.logz = po.logz + (log(s1 / (double)rp.n_parts) + dlz);
    p.c = po.c * bg.cfac;
    p.accept = a; p.e_center = bg.e_center; p.e_shift = bg.e_shift; p.e_seen = bg.e_seen;
    *out = p;
}

// The proposal of stage n from the moment totals T (augmented pair sums about `shift`, T[0] = Σ weights): θ̄, R, free subset +
// symmetrisation (smc_main.jl:457-465), random blocks (helpers.jl:215-260, Fisher-Yates on Philox), per block c²Σ_b = L Lᵀ
// (mutation.jl:81, once per stage) - the arithmetic of k_prepare_mutation, by a block of TT >= 128 threads, results in LDS only.
// Three barriers: [covariance | shuffle] -> [block matrices, marginal scales] -> [Cholesky, one wavefront per block] -> done.
struct Prop2 {                 // LDS pointers (k2_mutate carves them out of its dynamic LDS)
    double *covl, *A, *mean;
    int *bfree, *bptr, *fi;
    double *Lraw, *logdet, *mub, *sdd, *sdn;
    int *ball, *loff;
};
// chol_rows_in_regs with the dimension as a compile-time constant: straight-line code the scheduler can overlap across columns
template <int DB>
__device__ inline bool chol_full(double (&r)[12], int lane) {
    double q[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) q[k] = r[k];
    const bool ok = chol_rows_in_regs<DB, true, true>(q, DB, lane);
#pragma unroll
    for (int k = 0; k < DB; ++k) r[k] = q[k];
    return ok;
}
__device__ inline bool chol_full_dispatch(double (&r)[12], int d, int lane) {
    switch (d) {
    case 2: return chol_full<2>(r, lane); case 3: return chol_full<3>(r, lane); case 4: return chol_full<4>(r, lane);
    case 5: return chol_full<5>(r, lane); case 6: return chol_full<6>(r, lane); case 7: return chol_full<7>(r, lane);
    case 8: return chol_full<8>(r, lane); case 9: return chol_full<9>(r, lane); default: return chol_full<10>(r, lane);
    }
}

// Fisher-Yates partner of position i (helpers.jl:216) for lane i of a wavefront: depends on (seed, stage) only, so callers draw it
// while their loads are in flight
__device__ inline int shuffle_partner(unsigned long long seed, unsigned stage, int i0, int nf) {
    int jx = 0;
    if (i0 >= 1 && i0 < nf) {
        double ua, ub;
        uniform_pair(seed, 0ull, stage, rng_tag(P_BLK, (unsigned)i0, 0), ua, ub);
        jx = (int)(ua * (double)(i0 + 1));
        if (jx > i0) jx = i0;
    }
    return jx;
}

// jx_pre: shuffle_partner() of lane (threadIdx.x & 63), valid in wavefront 1 (threads 64..127)
// CHM: rows of the in-register Cholesky (12 serves n_para <= 12; 16 the wide kernels)
template <int CHM = 12>
__device__ inline bool proposal2(const double *T, const double *shift, int d, int nf, int nb, double c, unsigned long long seed,
                                 unsigned stage, const Prop2 &P, int *s_fail, int TT, int jx_pre, long long *prof = nullptr, long long *prof_any = nullptr) {
    const int t = threadIdx.x, da = d + 1;
    const double sw = T[0];
    if (t == 0) *s_fail = 0;
    for (int e = t; e < d * d; e += TT) {
        int a = e / d, b = e % d;
        if (a > b) { const int tmp = a; a = b; b = tmp; }
        const int ra = a + 1, rb = b + 1;
        const int p = ra * da - ra * (ra - 1) / 2 + (rb - ra);
        P.covl[e] = T[p] / sw - (T[a + 1] / sw) * (T[b + 1] / sw);
    }
    for (int a = t; a < d; a += TT) P.mean[a] = shift[a] + T[a + 1] / sw;
    if (t >= 64 && t < 128) {            // wave 1 shuffles while the others finish the covariance
        // Fisher-Yates (helpers.jl:216: i = nf-1 .. 1, swap(i, j_i)) without a serial pass over memory: lane l traces position l
        // back through the swaps in reverse order of application; what it ends at is the identity's entry that lands on l.
        const int i0 = t - 64;
        const int jx = jx_pre;
        int pos = i0;
        for (int i = 1; i < nf; ++i) {
            const int ji = __shfl(jx, i, 64);
            pos = pos == i ? ji : (pos == ji ? i : pos);
        }
        if (i0 < nf) P.bfree[i0] = pos;
        const int sub = (nf + nb - 1) / nb;
        if (i0 < nb) P.bptr[i0] = i0 * sub;
        if (i0 == nb) P.bptr[nb] = nf;
    }
    __syncthreads();
    K2_STAMP(prof, 4);
    K3_STAMP_ANY(prof_any, 0);
    // R_fr[f][g] = (R[fi f][fi g] + R[fi g][fi f]) / 2 (smc_main.jl:462-465), formed where it is used
    auto sig = [&](int f, int g2) { const int a = P.fi[f], b = P.fi[g2]; return (P.covl[a * d + b] + P.covl[b * d + a]) / 2.0; };
    for (int i = t; i < nf; i += TT) {
        const int f = P.bfree[i];
        const double sff = sig(f, f);
        P.ball[i] = P.fi[f];
        P.mub[i] = P.mean[P.fi[f]];
        P.sdd[i] = sqrt(c * c * sff);
        P.sdn[i] = sqrt(sff);
    }
    {
        const int sub = (nf + nb - 1) / nb;          // every block but the last has `sub` entries: block b's matrix starts at b sub²
        for (int e = t; e < nf * nf; e += TT) {      // (generous bound; entries beyond the packed total are skipped below)
            int b = 0, rem = e;
            while (b < nb - 1 && rem >= sub * sub) { rem -= sub * sub; ++b; }
            const int p0 = P.bptr[b], db = P.bptr[b + 1] - p0;
            if (rem < db * db) P.A[b * sub * sub + rem] = c * c * sig(P.bfree[p0 + rem / db], P.bfree[p0 + rem % db]);
        }
    }
    __syncthreads();
    K2_STAMP(prof, 5);
    K3_STAMP_ANY(prof_any, 1);
    // Right-looking Cholesky of block b inside ONE wavefront (wave b mod 4; lane i owns row i in registers, pivots and multipliers
    // broadcast with v_readlane): same k-ascending subtraction order per entry as the oracle's left-looking loop.
    {
        const int sub = (nf + nb - 1) / nb, wave = t >> 6, lane = t & 63, nwv = TT >> 6;
        for (int b = wave; b < nb; b += nwv) {
            const int p0 = P.bptr[b], db = P.bptr[b + 1] - p0, off = b * sub * sub;
            double r[CHM];
#pragma unroll
            for (int k = 0; k < CHM; ++k) r[k] = (lane < db && k < db) ? P.A[off + lane * db + k] : 0.0;
            bool ok;
            // (the whole triangle in every lane's registers - no lane exchange, ~220 dependent FP64 operations - was measured: the same
            // 3.2 µs; a dependent FP64 operation costs ~16 cycles with one wavefront per SIMD, whichever way the ten columns are cut)
            if constexpr (CHM == 12) {
                if (db == d && d <= 10 && d >= 2) ok = chol_full_dispatch(r, d, lane);      // one block over all parameters: no per-column branches
                else ok = chol_rows_in_regs<12, true>(r, db, lane);
            } else ok = chol_rows_in_regs<CHM, true>(r, db, lane);
            if (!ok && lane == 0) *s_fail = 1;
#pragma unroll
            for (int k = 0; k < CHM; ++k)
                if (lane < db && k < db) P.Lraw[off + lane * db + k] = k <= lane ? r[k] : 0.0;
            double dg = 1.0;                        // own diagonal entry L[lane][lane]
#pragma unroll
            for (int k = 0; k < CHM; ++k) dg = (k == lane) ? r[k] : dg;
            const double lg = (lane < db) ? log(dg) : 0.0;
            double ld = 0.0;
#pragma unroll
            for (int i = 0; i < CHM; ++i)
                if (i < db) ld += bcast_lane(lg, i);
            if (lane == 0) { P.logdet[b] = 2.0 * ld; P.loff[b] = off; }
        }
    }
    __syncthreads();
    K2_STAMP(prof, 6);
    K3_STAMP_ANY(prof_any, 2);
    return *s_fail == 0;
}

// LDS arrays of k2_mutate (the mutation's, as in k_mutate_reg, then the prologue's scratch)
template <int D>
struct Mut2Lds {
    static constexpr int DA = D + 1, NP = DA * (DA + 1) / 2, NPF = NP + 2;
    double *Ls, *mu_s, *sdd_s, *sdn_s, *red, *m_lo, *m_hi, *m_a, *m_b, *m_k, *l_par, *l_dat, *Lraw, *logdet_s, *mub_raw, *sdd_raw, *sdn_raw;
    int *ball_s, *m_fix, *m_fam, *bptr_s, *loff_s, *ball_raw;
    double *s_vt, *s_tot, *covl, *sig_f, *Aw, *Lw, *mean_s, *mu_f;
    int *bfree, *fi, *fi_j;
    // lik_cap: doubles reserved for staged likelihood data (the generic mutation body of n_para > 10 reads its data where it is: 0)
    __device__ explicit Mut2Lds(double *sm, int lik_cap = LIK_LDS_CAP) {
        Ls = sm; mu_s = Ls + D * D; sdd_s = mu_s + D; sdn_s = sdd_s + D; red = sdn_s + D;
        m_lo = red + 8; m_hi = m_lo + D; m_a = m_hi + D; m_b = m_a + D; m_k = m_b + D;
        l_par = m_k + D; l_dat = l_par + 2 * LIK_PAR_MAX; Lraw = l_dat + lik_cap; logdet_s = Lraw + D * D;
        mub_raw = logdet_s + D; sdd_raw = mub_raw + D; sdn_raw = sdd_raw + D;
        ball_s = (int *)(sdn_raw + D); m_fix = ball_s + D + (D & 1); m_fam = m_fix + D;
        bptr_s = m_fam + D; loff_s = bptr_s + D + 1; ball_raw = loff_s + D;
        s_vt = (double *)(((uintptr_t)(ball_raw + D + 1) + 15) & ~(uintptr_t)15);
        s_tot = s_vt + V2_MAXV * pad2(NPF);
        covl = s_tot + pad2(NPF) + 4; sig_f = covl + D * D; Aw = sig_f + D * D; Lw = Aw + D * D;
        mean_s = Lw + D * D; mu_f = mean_s + D;
        bfree = (int *)(mu_f + D); fi = bfree + D; fi_j = fi + D;
    }
};
// ... of the mutation body alone (Ls .. ball_raw: what a kernel that loads a finished proposal needs - k2b_mutate, stage2b.hpp)
constexpr size_t k2_lds_bytes_body(int D, int lik_cap = LIK_LDS_CAP) {
    return (size_t)(2 * D * D + 12 * D + 8 + 2 * LIK_PAR_MAX + lik_cap) * sizeof(double) + (size_t)(6 * D + 8) * sizeof(int) + 32;
}
constexpr size_t k2_lds_bytes(int D, int lik_cap = LIK_LDS_CAP) {
    const size_t np = (size_t)(D + 1) * (D + 2) / 2, npf = np + 2;
    return (size_t)(2 * D * D + 12 * D + 8 + 2 * LIK_PAR_MAX + lik_cap) * sizeof(double) + (size_t)(6 * D + 8) * sizeof(int) + 32 +
           (V2_MAXV * (npf + 1) + npf + 5 + 4 * D * D + 2 * D) * sizeof(double) + (size_t)(3 * D + 4) * sizeof(int) + 32;
}

struct Mut2Stage {             // what the prologue hands to the mutation body (LDS)
    Begin2 bg;
    Post2 po;
    int fail;
};

// Prologue of K2, every block: state + model constants into LDS, totals of the correction rows, decision, proposal.  Nothing in
// it goes through a single thread: every thread derives the (identical) decision from the totals itself.  Returns false when
// the block must not mutate (stale launch, stall, error); *rs_out = this stage resamples.
// writer_ov: as in begin2_block.  own_cfac: the step-size multiplier K1's block 0 leaves in Begin2 is computed here instead (a helper block of
// the K1 launch itself - stage2b.hpp - must not read a word another block of its launch writes; same expression, same bits).
template <int D, int T>
__device__ inline bool k2_prologue(DevState *st, Ctl2 *ctl, const ModelDev *md, const Mut2Args &ma, const Mut2Lds<D> &L, Mut2Stage *S, int nb, int nf,
                                   int *rs_out, int writer_ov = -1, bool own_cfac = false) {
    constexpr int NP = Mut2Lds<D>::NP, NPF = Mut2Lds<D>::NPF;
    const int tid = threadIdx.x, n = ma.n;
    const bool writer = writer_ov >= 0 ? writer_ov != 0 : blockIdx.x == 0;
    constexpr int NWB = sizeof(Begin2) / sizeof(double), NWP = sizeof(Post2) / sizeof(double);
    if (tid < NWB) reinterpret_cast<double *>(&S->bg)[tid] = reinterpret_cast<const double *>(&ctl->bg)[tid];
    if (tid < NWP) reinterpret_cast<double *>(&S->po)[tid] = reinterpret_cast<const double *>(&ctl->ps[(n - 1) & 1])[tid];
    const double thr = st->rp.threshold, rtol = st->rp.phi_rtol;
    for (int k = tid; k < D; k += T) {
        L.m_lo[k] = md->lo[k]; L.m_hi[k] = md->hi[k]; L.m_a[k] = md->prior_a[k]; L.m_b[k] = md->prior_b[k]; L.m_k[k] = md->prior_k[k];
        L.m_fix[k] = md->fixed[k]; L.m_fam[k] = md->prior_family[k];
    }
    for (int k = tid; k < 2 * LIK_PAR_MAX; k += T) L.l_par[k] = md->lik[k / LIK_PAR_MAX].par[k % LIK_PAR_MAX];
    if (tid < nf) L.fi[tid] = md->free_inds[tid];
    const int jx_pre = (tid >= 64 && tid < 128) ? shuffle_partner(ma.seed, (unsigned)n, tid - 64, nf) : 0;     // under the loads' latency
    K2_STAMP(ma.prof, 1);
    if (ma.cmrows.mb) {                                                // mailbox: see begin2_block
        __syncthreads();
        if (S->bg.stage != n || !S->bg.final || S->po.stage != n - 1) return false;
    }
    reduce_rows<pad2(NPF), 1, T>(ma.cmrows, L.s_vt, L.s_tot);          // (its barriers also publish the LDS copies above)
    K2_STAMP(ma.prof, 2);
    if (S->bg.stage != n || !S->bg.final || S->po.stage != n - 1) return false;
    if (own_cfac) {
        if (tid == 0) { const double a = S->bg.accept, tg = st->rp.target; S->bg.cfac = 0.95 + 0.10 * exp(16.0 * (a - tg)) / (1.0 + exp(16.0 * (a - tg))); }
        __syncthreads();
    }
    double ess;
    const int dec = decide2(S->bg, thr, rtol, L.s_tot[0], L.s_tot[1], &ess);
    const bool stall = dec == 4 || dec < 0 || (dec == 1 && !ma.sel_enqueued);
    if (stall) {
        if (writer && tid == 0) {
            if (dec < 0) { ma.rec.phi[n - 1] = S->bg.phi_n; ma.rec.ess[n - 1] = ess; ctl->status.err = dec; }
            ctl->status.stage = n;
            ctl->status.code = dec == 4 ? 4 : (dec < 0 ? 9 : 3);
        }
        return false;
    }
    const int rs = dec == 1 ? 1 : 0;
    *rs_out = rs;
    K2_STAMP(ma.prof, 3);
    // moments of the resampled cloud (k2_gather's rows) replace the correction's on resample stages
    if (rs) reduce_rows<pad2(NP), 1, T>(ma.gmrows, L.s_vt, L.s_tot + 2);
    Prop2 P{L.covl, L.Aw, L.mean_s, L.bfree, L.bptr_s, L.fi, L.Lraw, L.logdet_s, L.mub_raw, L.sdd_raw, L.sdn_raw, L.ball_raw, L.loff_s};
    if (!proposal2<(D > 12 ? 16 : 12)>(L.s_tot + 2, S->po.shift, D, nf, nb, S->po.c * S->bg.cfac, ma.seed, (unsigned)n, P, &S->fail, T, jx_pre, ma.prof)) {
        // PosDefException aborts the run (mutation.jl:81)
        if (writer && tid == 0) { ctl->status.err = SMCMI_ERR_POSDEF; ctl->status.stage = n; ctl->status.code = 9; }
        return false;
    }
    K2_STAMP(ma.prof, 7);
    return true;
}

// Block 0, after its mutation: the stage's bookkeeping (src/smc_main.jl:427-455) - nothing in this launch reads it, so the
// logarithm of the log-MDD increment and the records stay off every block's critical path.
template <int D, int T>
__device__ inline void k2_bookkeeping(DevState *st, Ctl2 *ctl, const Mut2Args &ma, const Mut2Lds<D> &L, const Mut2Stage *S, int rs) {
    const int tid = threadIdx.x, n = ma.n;
    __shared__ Post2 s_ps;
    if (tid == 0) {
        const double s1 = L.s_tot[0], s2 = L.s_tot[1];
        post2(n, S->bg, S->po, st->rp, s1, s2, s1 * s1 / s2, rs, &s_ps);
        ma.rec.phi[n - 1] = s_ps.phi_n; ma.rec.ess[n - 1] = s_ps.ess; ma.rec.resampled[n - 1] = rs; ma.rec.c[n - 1] = s_ps.c;
    }
    __syncthreads();
    if (tid < D) { s_ps.shift[tid] = L.mean_s[tid]; st->mean[tid] = L.mean_s[tid]; }       // st->mean / cov: diagnostics, stand-alone readers
    for (int e = tid; e < D * D; e += T) st->cov[e] = L.covl[e];
    __syncthreads();
    constexpr int NWP = sizeof(Post2) / sizeof(double);
    if (tid < NWP) reinterpret_cast<double *>(&ctl->ps[n & 1])[tid] = reinterpret_cast<const double *>(&s_ps)[tid];
}

// The two likelihood descriptors as views for the mutation body; data that fits is staged in LDS (l_dat, LIK_LDS_CAP doubles) by the
// whole block - the caller's next barrier publishes it.
template <int T>
__device__ inline void k2_stage_lik(const LikDev &ld0, const LikDev &ld1, double *l_par, double *l_dat, LikView (&lv)[2]) {
    const int tid = threadIdx.x;
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

// The MH steps of one particle (src/mutation.jl:86-138 over steps x blocks; helpers.jl:87-164 for alpha < 1): the body K2 and the
// persistent segment kernel (stage3.hpp) share - same arithmetic in the same order.  The proposal's arrays are in LDS (L.Lraw,
// L.logdet_s, L.mub_raw, L.sdd_raw, L.sdn_raw, L.ball_raw, L.bptr_s, L.loff_s), published by a barrier before the call; proposal 0's
// random numbers may arrive in (step_prob, uc, z) (PREDRAW, or ma.zbuf).  ldz: leading dimension of ma.zbuf.  All threads call.
template <int D, bool ALPHA1, int T, bool PREDRAW, bool ZPART = false>
__device__ inline void k2_mh_steps(const Mut2Lds<D> &L, double *mixbuf, int *mixpos, double *mixzt, const Mut2Args &ma, long long ldz,
                                   const LikView (&lv)[2], const ModelView &mv, int nb, int nf, bool live, long long i, unsigned long long pid,
                                   unsigned stage, double phi_n, double (&x)[D], double &like, double &lprior, double &like_prev, double &accept,
                                   double &step_prob, double &uc, double (&z)[D]) {
SMCMI_FP_CONTRACT
    const int tid = threadIdx.x, n_steps = ma.n_steps, has_other = ma.has_other;
    const double c_alpha = ma.alpha;
    double *Ls = L.Ls, *Lraw = L.Lraw, *logdet_s = L.logdet_s, *mub_raw = L.mub_raw, *sdd_raw = L.sdd_raw, *sdn_raw = L.sdn_raw;
    int *bptr_s = L.bptr_s, *loff_s = L.loff_s, *ball_raw = L.ball_raw;
    auto XN = [&](int k) { return x[k]; };
    const MixDense<D> MX(mixbuf, mixpos);
    double *Wraw = mixbuf + (ALPHA1 ? 0 : MixDense<D>::DOUBLES);
    if constexpr (!ALPHA1) mix_invert_factors<D, T>(Lraw, Wraw, loff_s, bptr_s, nb, tid);     // (prologue / pre-load ended with a barrier)
    for (int step = 0; step < n_steps; ++step) {
        for (int b = 0; b < nb; ++b) {
            if ((nb > 1 || step == 0) && (step | b) != 0) __syncthreads();   // previous block's readers (the prologue ends with a barrier)
            const int p0 = bptr_s[b], db = bptr_s[b + 1] - p0;
            if (nb > 1 || step == 0) {              // expand this block's constants to the padded D x D form
                const double *Lb = Lraw + loff_s[b];
                if constexpr (ALPHA1) {
                    for (int e = tid; e < D * D; e += T) Ls[e] = 0.0;
                    __syncthreads();
                    for (int e = tid; e < db * db; e += T) {
                        const int r = e / db, cidx = e % db;
                        if (cidx <= r) Ls[cidx * D + ball_raw[p0 + r]] = Lb[r * db + cidx];   // transposed: Ls[e][k] = M[k][e]
                    }
                } else
                    mix_expand<D, T>(MX, Lb, Wraw + loff_s[b], ball_raw + p0, mub_raw + p0, sdd_raw + p0, sdn_raw + p0, db, logdet_s[b], tid);
                __syncthreads();
            }
            if (!live) continue;
            const unsigned t = (unsigned)(step * nb + b);
            if (ma.zbuf && (!ZPART || (int)t < ma.z_ahead)) {
                if (ZPART || t != 0) {            // (ZPART: proposal 0 is loaded here as well, nothing is carried into the loop)
                    const double *zt = ma.zbuf + (long long)t * (D + 2) * ldz + i;
                    step_prob = zt[0];
                    uc = zt[ldz];
#pragma unroll
                    for (int e = 0; e < D; ++e) z[e] = zt[(long long)(2 + e) * ldz];
                }
            } else if (!PREDRAW || t != 0) draw2<D>(ma.seed, pid, stage, t, db, ma.debug, step_prob, uc, z);   // (proposal 0 may have been drawn ahead of the prologue)
            double prior_new = SMCMI_NEG_INF, like_new = SMCMI_NEG_INF, like_old_data = SMCMI_NEG_INF;
            double q0 = 0.0, q1 = 0.0;
            double xo[D];
            if constexpr (ALPHA1) {
                double zz2 = 0.0;
#pragma unroll
                for (int e = 0; e < D; ++e) zz2 += z[e] * z[e];
                q1 = (-((double)db * LOG2PI + logdet_s[b] + zz2) / 2.0 < -745.1332191019412) ? __builtin_nan("") : 0.0;
                double sacc[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { xo[k] = x[k]; sacc[k] = 0.0; }
#pragma unroll
                for (int e = 0; e < D; ++e) {
#pragma unroll
                    for (int k = 0; k < D; ++k) sacc[k] += Ls[e * D + k] * z[e];
#pragma unroll
                    for (int k = 0; k < D; ++k) asm volatile("" : "+v"(sacc[k]));
                }
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = xo[k] + sacc[k];
                if (ma.debug & 2) { prior_new = lprior - 0.1 * zz2; like_new = like - 0.2; like_old_data = 0.0; }
                else if (in_bounds_s<D>(mv, XN)) {
                    prior_new = logprior_s<D>(mv, XN, has_other);
                    like_new = loglik_s<D>(lv[0], XN);
                    if (like_new == SMCMI_NEG_INF) prior_new = SMCMI_NEG_INF;
                    like_old_data = (lv[1].family == SMCMI_LIK_NONE) ? 0.0 : loglik_s<D>(lv[1], XN);
                }
            } else {
            // mixture draw + proposal densities in parameter order (mix_propose, kernels.hpp): no gather / scatter, no division
            double zz2 = 0.0;
#pragma unroll
            for (int e = 0; e < D; ++e) zz2 += z[e] * z[e];
            double xn[D];
            q0 = mix_propose<D, T>(MX, x, z, uc, c_alpha, (ma.debug & 4) != 0, mixzt + tid, xn);       // q0 - q1 as one number
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
}

// One mutation block's row for the next stage's begin (energy power sums on adaptive schedules, Σ accept, energy maximum) from the
// particles' post-mutation values; scratch: T / 64 * ES doubles of LDS (the likelihood data is dead by now), red: 8 doubles.
// coh: the row is totalled inside this launch (Tail2 / stage3.hpp).  All threads call; starts and ends with a barrier.
template <int T, class ST>
__device__ inline void k2_mut_row_f(bool adaptive, double like, double like_prev, double w_part, double acc_val, double e_center, bool live,
                                    bool rs, double *scratch, double *red, ST store) {
    const int tid = threadIdx.x;
    __syncthreads();
    double em = energy_or_ninf(like, like_prev, rs ? 1.0 : w_part, live);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) em = fmax(em, __shfl_xor(em, off, 64));
    __shared__ double emx[T / 64];
    if ((tid & 63) == 0) emx[tid >> 6] = em;
    if (adaptive) {
        double es[ES];
        energy_terms(es, w_part, like, like_prev, e_center, live, rs);
        es[EACC] = acc_val;
        const double tot = block_reduce_es2<T / 64>(es, scratch);
        if (tid < ES) store(tid, tot);
    } else {
        double a1[1] = {acc_val};
        Butterfly<0, 32>::run(a1, tid & 63);
        if ((tid & 63) == 0) red[tid >> 6] = a1[0];
        __syncthreads();
        if (tid < ES) {
            double sacc = ((red[0] + red[1]) + red[2]) + red[3];
            if constexpr (T == 512) sacc += ((red[4] + red[5]) + red[6]) + red[7];
            store(tid, tid == EACC ? sacc : 0.0);
        }
    }
    __syncthreads();
    if (tid == 0) {
        double m = emx[0];
#pragma unroll
        for (int w = 1; w < T / 64; ++w) m = fmax(m, emx[w]);
        store(RMAX_IDX, m);
    }
}
template <int T>
__device__ inline void k2_mut_row(double *row, bool adaptive, double like, double like_prev, double w_part, double acc_val, double e_center, bool live,
                                  bool rs, double *scratch, double *red, bool coh) {
    k2_mut_row_f<T>(adaptive, like, like_prev, w_part, acc_val, e_center, live, rs, scratch, red, [&](int idx, double v) { row_store(row + idx, v, coh); });
}

// K2.  The mutation body is k_mutate_reg's (src/mutation.jl:56-138, helpers.jl:87-164; same arithmetic in the same order), fed
// from LDS by the prologue instead of from DevState; it reads the particle from buffer 0 (buffer 1 on resample stages: the
// gathered cloud) and always writes buffer 0, applies normalize_weights! (particle.jl:362-366: W̃ N / ΣW̃, two roundings; 1 after a
// resample) to the weight column and its history, and leaves one row of RMUT sums for the next stage's begin.
template <int D, bool ALPHA1, int T, bool TAIL>
__global__ void __launch_bounds__(T, T == 512 ? 2 : 3) k2_mutate(CloudPtrs cl, DevState *st, Ctl2 *ctl, const ModelDev *md, Geo2 g,
                                                                               Mut2Args ma, int nb, int nf) {
SMCMI_FP_CONTRACT
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ Mut2Stage S;
    const Mut2Lds<D> L(sm);
    const int tid = threadIdx.x, n = ma.n;
    K2_STAMP(ma.prof, 0);
    const double nrm_N = ma.n_parts;
    const int nrm_hist = ma.store_history;
    const LikDev &ld0 = ma.lik[0], &ld1 = ma.lik[1];
    double *red = L.red, *l_dat = L.l_dat, *Lraw = L.Lraw, *logdet_s = L.logdet_s;
    double *mub_raw = L.mub_raw, *sdd_raw = L.sdd_raw, *sdn_raw = L.sdn_raw;
    int *bptr_s = L.bptr_s, *loff_s = L.loff_s, *ball_raw = L.ball_raw;
    // ---- the particle (speculatively from buffer 0: only resample stages read the gathered cloud in buffer 1) and the likelihood
    // data: none of it depends on the stage's decision, so the loads are in flight while the prologue totals rows and factorises
    long long beg, end;
    vchunk(g, blockIdx.x / g.nb2, blockIdx.x % g.nb2, T, beg, end);
    const long long i = beg + tid;
    const bool live = i < end;
    const long long il = live ? i : (end > beg ? end - 1 : 0);           // unconditional loads (clamped row)
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    double like, lprior, like_prev, accept = 0.0, wt_i;
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = col(cl, 0, k)[il];
    like = col(cl, 0, D)[il]; lprior = col(cl, 0, D + 1)[il]; like_prev = col(cl, 0, D + 2)[il];
    wt_i = ma.wt[il];
    ModelView mv{D, L.m_fix, L.m_fam, L.m_lo, L.m_hi, L.m_a, L.m_b, L.m_k};
    LikView lv[2];
    k2_stage_lik<T>(ld0, ld1, L.l_par, l_dat, lv);
    // ---- the first proposal's random numbers depend on (seed, particle, stage) only: drawing them here puts ~40 % of the mutation's
    // arithmetic under the latency of the loads above and of the prologue's row totals
    // (512-thread blocks only: with 3 wavefronts per SIMD the 168-register budget has no room to carry them across the prologue)
    constexpr bool PREDRAW = T == 512;
    double step_prob, uc, z[D];
    if (ma.zbuf) {                 // drawn ahead by K1's extra blocks: D + 2 coalesced loads
        const double *zt = ma.zbuf + il;
        step_prob = zt[0];
        uc = zt[g.n];
#pragma unroll
        for (int e = 0; e < D; ++e) z[e] = zt[(long long)(2 + e) * g.n];
    } else if constexpr (PREDRAW) draw2<D>(ma.seed, pid, (unsigned)n, 0u, nb == 1 ? nf : (nf + nb - 1) / nb, ma.debug, step_prob, uc, z);
    int rs = 0;
    double phi_n, e_center, nrm_sumw;
    if (ma.pre) {
        // decision and proposal from k2_prepare: model constants + the proposal's arrays into LDS, one barrier
        const Prop2Glob *G = ma.pre;
        const int pstage = G->stage, pgo = G->go;
        rs = G->rs; nrm_sumw = G->s1; phi_n = G->phi_n; e_center = G->e_center;
        for (int k = tid; k < D; k += T) {
            L.m_lo[k] = md->lo[k]; L.m_hi[k] = md->hi[k]; L.m_a[k] = md->prior_a[k]; L.m_b[k] = md->prior_b[k]; L.m_k[k] = md->prior_k[k];
            L.m_fix[k] = md->fixed[k]; L.m_fam[k] = md->prior_family[k];
        }
        for (int k = tid; k < 2 * LIK_PAR_MAX; k += T) L.l_par[k] = md->lik[k / LIK_PAR_MAX].par[k % LIK_PAR_MAX];
        for (int e = tid; e < nf * nf; e += T) Lraw[e] = G->Lraw[e];
        for (int e = tid; e < nf; e += T) { mub_raw[e] = G->mub[e]; sdd_raw[e] = G->sdd[e]; sdn_raw[e] = G->sdn[e]; ball_raw[e] = G->ball[e]; }
        for (int b = tid; b < nb; b += T) { loff_s[b] = G->loff[b]; logdet_s[b] = G->logdet[b]; }
        for (int b = tid; b <= nb; b += T) bptr_s[b] = G->bptr[b];
        if (pstage != n || !pgo) return;
        __syncthreads();
    } else {
        if (!k2_prologue<D, T>(st, ctl, md, ma, L, &S, nb, nf, &rs)) return;
        phi_n = S.bg.phi_n; e_center = S.bg.e_center; nrm_sumw = L.s_tot[0];
    }
    const unsigned stage = (unsigned)n;
    if (rs) {                       // the resampled cloud is in buffer 1 (k2_gather)
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = col(cl, 1, k)[il];
        like = col(cl, 1, D)[il]; lprior = col(cl, 1, D + 1)[il]; like_prev = col(cl, 1, D + 2)[il];
    }
    double w_part = 0.0;
    if (live) {
        w_part = rs ? 1.0 : (wt_i * nrm_N) / nrm_sumw;                      // W·N then /ΣW̃, two roundings like the reference
        col(cl, 0, D + 4)[i] = w_part;
        if (ma.hist_W && nrm_hist) ma.hist_W[(long long)(n - 1) * ma.hist_ld + i] = w_part;
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = 0.0;
        like = lprior = like_prev = 0.0;
    }
    __shared__ double mixbuf[ALPHA1 ? 1 : MixDense<D>::DOUBLES + D * D];
    __shared__ int mixpos[ALPHA1 ? 1 : D];
    __shared__ double mixzt[ALPHA1 ? 1 : T * D];               // private z columns of the diagonal component's draw
    K2_STAMP(ma.prof, 8);
    k2_mh_steps<D, ALPHA1, T, PREDRAW>(L, mixbuf, mixpos, mixzt, ma, g.n, lv, mv, nb, nf, live, i, pid, stage, phi_n, x, like, lprior, like_prev, accept,
                                       step_prob, uc, z);
    K2_STAMP(ma.prof, 9);
    double acc_val = 0.0;
    if (live) {
        // a particle that accepted nothing still holds what buffer 0 holds (three quarters of the cloud at the target acceptance
        // rate): its 13 value columns are not written again - the stage's HBM write traffic drops by two thirds.  After a resample
        // the particle came from buffer 1 and every column is written.
        if (accept > 0.0 || rs) {
#pragma unroll
            for (int k = 0; k < D; ++k) col(cl, 0, k)[i] = x[k];
            col(cl, 0, D)[i] = like;
            col(cl, 0, D + 1)[i] = lprior;
            col(cl, 0, D + 2)[i] = like_prev;
        }
        acc_val = accept / (double)nf;                      // quirk Q2: normalised by n_free only
        col(cl, 0, D + 3)[i] = acc_val;
    }
    // ---- this block's row for the next stage's begin: energy power sums (adaptive schedules), Σ accept, energy maximum
    k2_mut_row<T>(ma.rows_mut + (long long)blockIdx.x * RMUT, ma.adaptive != 0, like, like_prev, w_part, acc_val, e_center, live, rs != 0, l_dat, red,
                  TAIL && ma.tail.tick != nullptr);
    K2_STAMP(ma.prof, 10);
    if constexpr (TAIL) tail_reduce<T>(ma.tail, ma.rows_mut, (int)blockIdx.x / g.nb2, g.nb2, RMUT, RMAX_IDX, T == 256 ? 1 : 0);
    if (blockIdx.x == 0 && !ma.pre) k2_bookkeeping<D, T>(st, ctl, ma, L, &S, rs);
    K2_STAMP(ma.prof, 11);
}

// K2 for n_para > 10 (the Kalman-filter family of BASELINE config 5, and any device family with 11 - 16 parameters): the same prologue
// as k2_mutate - every block totals the V x m correction table, decides, builds the proposal in LDS (k2_prologue) - in front of the
// GENERIC mutation body (kernels.hpp mutate_generic: per-particle vectors in LDS columns; the lgss_kalman filters with their structure
// values in DPP operands), which engine 1 reaches through k_prepare_mutation + k_mutate.  LS = 1: one thread per particle, 256 particles
// per block; LS = 4 (lgss_kalman, small clouds): four lanes per particle, 64 particles per block, the quad runs the proposal redundantly
// and shares the filter (model.hpp kalman_lgss_quad).  One mutation row per block (never paired), totalled per virtual shard by the last
// block to finish (Tail2): with global particle ids in the RNG the results do not depend on the number of handles, bit for bit.
// Replaces, for these models, engine 1's eight-launch stage (src/smc_main.jl:427-484 as k_post_correct ... k_mutate) by K1 -> K2.
constexpr size_t k2w_lds_bytes(int D, int LS) {
    const size_t pro = (k2_lds_bytes(D, 0) + 15) / 16 * 16;
    return pro + (LS == 4 ? (size_t)4 * mutate_wave_bytes_ls4(13) : (size_t)4 * D * 256 * sizeof(double)) + 64;
}
template <int D, int LS>
__global__ void __launch_bounds__(256, 1) k2w_mutate(CloudPtrs cl, DevState *st, Ctl2 *ctl, const ModelDev *md, Geo2 g, Mut2Args ma, int nb, int nf) {
    static_assert(LS == 1 || (LS == 4 && D == 13), "four lanes per particle: the lgss_kalman family (13 parameters)");
    constexpr int TB2 = 256, P = LS == 4 ? 64 : 256;           // threads / particles of a block
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ Mut2Stage S;
    const Mut2Lds<D> L(sm, 0);
    const int n = ma.n;
    int rs = 0;
    if (!k2_prologue<D, TB2>(st, ctl, md, ma, L, &S, nb, nf, &rs)) return;
    const double phi_n = S.bg.phi_n, e_center = S.bg.e_center, nrm_sumw = L.s_tot[0], nrm_N = ma.n_parts;
    // ---- the per-particle LDS vectors behind the prologue's area
    double *gen = sm + (k2_lds_bytes(D, 0) + 15) / 16 * 2;
    const int T = LS == 4 ? 16 : TB2;                                                // stride of the per-particle vectors
    const int tid = LS == 4 ? (((int)threadIdx.x & 63) >> 2) : (int)threadIdx.x;     // column in them
    const int quad_lane = threadIdx.x & 3;
    const bool lead = LS == 1 || quad_lane == 0;
    double *wave_base = LS == 4 ? gen + (long long)(threadIdx.x >> 6) * (mutate_wave_bytes_ls4(13) / 8) : gen;
    double *th = wave_base, *tn = th + (long long)D * T, *y = tn + (long long)D * T, *v = y + (long long)D * T;
    const int vl = (int)blockIdx.x / g.nb2, r = (int)blockIdx.x % g.nb2;
    long long beg, end;
    vchunk(g, vl, r, P, beg, end);
    const long long i = beg + (LS == 4 ? (long long)(threadIdx.x >> 2) : (long long)threadIdx.x);
    const bool live = i < end;
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    double w_part = 0.0;
    if (live) {
        w_part = rs ? 1.0 : (ma.wt[i] * nrm_N) / nrm_sumw;                  // W·N then /ΣW̃, two roundings like the reference (particle.jl:362-366)
        if (lead) {
            col(cl, 0, D + 4)[i] = w_part;
            if (ma.hist_W && ma.store_history) ma.hist_W[(long long)(n - 1) * ma.hist_ld + i] = w_part;
        }
    }
    MutArgs ga{};
    ga.seed = ma.seed; ga.gid0 = ma.gid0; ga.debug = ma.debug;
    const ModelView mv{D, L.m_fix, L.m_fam, L.m_lo, L.m_hi, L.m_a, L.m_b, L.m_k};
    double like = 0.0, lprior = 0.0, like_prev = 0.0, accept = 0.0;
    // (the resampled cloud is in buffer 1: k2_gather)
    mutate_generic<0, LS>(cl, md, ga, th, tn, y, v, T, tid, quad_lane, i, live, rs, pid, (unsigned)n, ma.alpha, phi_n, nb, ma.n_steps, D, L.Lraw, L.mub_raw, L.sdd_raw,
                          L.sdn_raw, L.logdet_s, L.bptr_s, L.ball_raw, L.loff_s, mv, like, lprior, like_prev, accept);
    double acc_val = 0.0;
    const bool counted = live && lead;                          // lanes 1..3 of a quad carry copies: they store nothing and add nothing to the row
    if (counted) {
        for (int k = 0; k < D; ++k) col(cl, 0, k)[i] = th[k * T + tid];
        col(cl, 0, D)[i] = like;
        col(cl, 0, D + 1)[i] = lprior;
        col(cl, 0, D + 2)[i] = like_prev;
        acc_val = accept / (double)nf;                          // quirk Q2: normalised by n_free only
        col(cl, 0, D + 3)[i] = acc_val;
    }
    // ---- this block's row for the next stage's begin (the per-particle vectors are dead: their area is the reduction's scratch)
    k2_mut_row<TB2>(ma.rows_mut + (long long)blockIdx.x * RMUT, ma.adaptive != 0, like, like_prev, counted ? w_part : 0.0, counted ? acc_val : 0.0, e_center, counted,
                    rs != 0, gen, L.red, ma.tail.tick != nullptr);
    tail_reduce<TB2>(ma.tail, ma.rows_mut, vl, g.nb2, RMUT, RMAX_IDX, 0);
    if (blockIdx.x == 0) k2_bookkeeping<D, TB2>(st, ctl, ma, L, &S, rs);
}

// Decision, bookkeeping and proposal of stage n by one block (large clouds, sharded runs): K2's prologue as its own launch.
template <int D>
__global__ void __launch_bounds__(256) k2_prepare(DevState *st, Ctl2 *ctl, const ModelDev *md, Mut2Args ma, int nb, int nf, Prop2Glob *out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ Mut2Stage S;
    const Mut2Lds<D> L(sm);
    const int tid = threadIdx.x, n = ma.n;
    int rs = 0;
    ma.pre = nullptr;
    if (!k2_prologue<D, 256>(st, ctl, md, ma, L, &S, nb, nf, &rs)) return;      // (a stall leaves out->stage stale: K2 does nothing)
    for (int e = tid; e < nf * nf; e += 256) out->Lraw[e] = L.Lraw[e];
    for (int e = tid; e < nf; e += 256) { out->mub[e] = L.mub_raw[e]; out->sdd[e] = L.sdd_raw[e]; out->sdn[e] = L.sdn_raw[e]; out->ball[e] = L.ball_raw[e]; }
    for (int b = tid; b < nb; b += 256) { out->loff[b] = L.loff_s[b]; out->logdet[b] = L.logdet_s[b]; }
    for (int b = tid; b <= nb; b += 256) out->bptr[b] = L.bptr_s[b];
    if (tid == 0) { out->rs = rs; out->s1 = L.s_tot[0]; out->phi_n = S.bg.phi_n; out->e_center = S.bg.e_center; out->go = 1; out->stage = n; }
    k2_bookkeeping<D, 256>(st, ctl, ma, L, &S, rs);
}

// ... and as the helper block of K1 (Prep2Args above): T1 threads, the launch's dynamic LDS.  A decision to resample stalls the stage
// (status 3) like any launch enqueued without its selection kernels: nothing is left in Prop2Glob, the mutation launch does nothing.
template <int D>
__device__ void k2_prepare_block(DevState *st, Ctl2 *ctl, const Prep2Args &pb, int n) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ Mut2Stage S;
    const Mut2Lds<D> L(sm);
    const int tid = threadIdx.x;
    Mut2Args ma{};
    ma.n = n; ma.seed = pb.seed; ma.cmrows = pb.cmrows; ma.sel_enqueued = 0; ma.rec = pb.rec;
    int rs = 0;
    if (!k2_prologue<D, T1>(st, ctl, pb.md, ma, L, &S, pb.nb, pb.nf, &rs, 1, true)) return;
    Prop2Glob *out = pb.out;
    const int nb = pb.nb, nf = pb.nf;
    for (int e = tid; e < nf * nf; e += T1) out->Lraw[e] = L.Lraw[e];
    for (int e = tid; e < nf; e += T1) { out->mub[e] = L.mub_raw[e]; out->sdd[e] = L.sdd_raw[e]; out->sdn[e] = L.sdn_raw[e]; out->ball[e] = L.ball_raw[e]; }
    for (int b = tid; b < nb; b += T1) { out->loff[b] = L.loff_s[b]; out->logdet[b] = L.logdet_s[b]; }
    for (int b = tid; b <= nb; b += T1) out->bptr[b] = L.bptr_s[b];
    if (tid == 0) { out->rs = rs; out->s1 = L.s_tot[0]; out->phi_n = S.bg.phi_n; out->e_center = S.bg.e_center; out->go = 1; out->stage = n; }
    k2_bookkeeping<D, T1>(st, ctl, ma, L, &S, rs);
}

}  // namespace smcmi
