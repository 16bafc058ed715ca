// run2.hpp - host driver of engine 2 (stage2.hpp): the two-launch stage for one handle, for a group of lock-stepped handles in one
// process and for one handle per process with RCCL.  Included by smcmi.hip after sharded.hpp (ShardGroup, RCCL loader).
//
// Per stage (src/smc_main.jl:377-508) the driver enqueues
//     K1 k2_correct -> [k2_scan -> k2_gather where a resample is possible] -> K2 k2_mutate
// (with k2_begin -> P x k2_pass -> k2_finish in front where no verified prediction of ϕ_n is available), never reads the device
// inside a batch of stages, and resumes a stage that stalled (Status2::code 2 / 3 / 4) at its next sync.  With more than one
// handle every row set is first totalled per virtual shard (k2_reduce) and all-gathered (V x m doubles: RCCL over xGMI, or
// device copies inside a process); every handle then totals the same V rows in the same order, so all take the same decisions
// and the results do not depend on the number of handles.
#pragma once
#include <atomic>
#include <cstring>
#include <map>


constexpr int SEG3_MAX_LAUNCHES = 4096;     // segment launches of one run whose stage counts are kept for the profile (more are not timed)
static void free_eng2(Eng2 *e) {
    if (!e) return;
    void *ptrs[] = {e->d_ctl, e->rows_mut, e->rows_cm, e->csum, e->csum_full, e->rows_gm, e->rows_pass[0], e->rows_pass[1], e->vt_mut, e->vt_cm,
                    e->vt_gm, e->vt_pass, e->d_ranges, e->d_ranges_all, e->d_prof, e->d_pre, e->d_tick, e->d_tick3, e->d_rec3, e->d_to3, e->d_done3, e->d_gran3, e->d_sel3, e->d_transit3};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (e->h_note3) hipHostFree(e->h_note3);
    delete e;
}

// virtual-shard geometry of a handle that is shard `rank` of `world` (a function of N and the shard count's divisibility only)
static bool make_geo2(const smcmi_handle *h, int world, int rank, bool single, Geo2 *out) {
    Geo2 g{};
    g.N = h->cfg.n_parts; g.n = h->n;
    if (world < 1 || g.n * world != g.N) return false;
    int V = 0;
    for (int cand : {8, 4, 2, 1})
        if (cand % world == 0 && g.n % (cand / world) == 0) { V = cand; break; }
    if (!V) { if (world <= V2_MAXV) V = world; else return false; }
    g.V = V; g.Vl = V / world; g.v0 = rank * g.Vl; g.nv = g.n / g.Vl;
    if (g.nv < 1) return false;
    // n_para > 10: the generic mutation body behind engine 2's prologues (stage2.hpp k2w_mutate) - 256 particles per block with one
    // thread per particle, 64 with four lanes per particle (lgss_kalman on small clouds); rows always totalled per virtual shard (Tail2)
    g.wide = h->d > 10 ? (use_ls4_mutate(h) ? 4 : 1) : 0;
    if (g.wide) {
        g.t2 = g.wide == 4 ? 64 : 256;
        g.nb2 = (int)((g.nv + g.t2 - 1) / g.t2);
        if (g.nv > 65536 || (long long)g.nb2 * g.Vl > 1024) return false;      // (a correction row is 512 particles, one per thread: nb1 <= 128)
        g.direct = 0; g.inker = 1;
        g.nb1 = (int)std::max<long long>(1, (g.nv + 511) / 512);
        g.per1 = T1;
        g.nbg = (int)std::max<long long>(1, std::min<long long>((g.nv + 511) / 512, 256));
        g.perg = ((g.nv + g.nbg - 1) / g.nbg + 255) / 256 * 256;
        *out = g;
        return true;
    }
    g.t2 = 512;
    g.nb2 = (int)((g.nv + g.t2 - 1) / g.t2);
    // direct: every block totals the per-block rows itself - one handle, <= GRP rows per virtual shard, and the 512-thread mutation
    // blocks (one per CU) resident at once
    // (beyond 256 blocks - up to 62 per virtual shard - the persistent segments give every worker two chunks: stage3.hpp k3_segment<D, true, RIDE, 2>;
    // eng2_eligible keeps such a cloud on engine 1 unless its run qualifies for them)
    g.direct = (single && world == 1 && g.nb2 <= GRP && (long long)g.nb2 * V <= 2 * (256 - V2_MAXV)) ? 1 : 0;
    if (getenv("SMCMI_E2_REDUCED")) g.direct = 0;                                    // development: force the k2_reduce path on one handle
    // several handles with small shards: one 512-thread mutation block per CU as well, prologues in the kernels, fed by the gathered totals
    g.inker = (g.direct || (!single && (long long)g.nb2 * g.Vl <= 256)) ? 1 : 0;
    // large shards (stage2b.hpp k2b_mutate): the same 512-particle mutation blocks at half the registers - two per CU, 4 wavefronts per SIMD
    // (256-thread blocks - 489 raw rows per virtual shard at 125 000 particles, paired into canonical rows by whoever totals them - cost
    // the block that totals a shard's rows ~10 µs at the END of every mutation launch: twice the loads, a quarter of them in flight)
    // correction blocks per virtual shard: 1024 particles per block (two passes of its 512 threads), at most 16 rows per virtual shard for
    // K2's prologue to total while the cloud is small
    // (the direct geometry: one correction row per 512 particles, the same particles as a mutation row - the persistent segment kernel
    // of stage3.hpp holds one particle per thread and writes exactly these rows, so both engines total the same numbers)
    // every geometry cuts a virtual shard the same way, so all of them total the same rows (<= 128 rows per virtual shard: the blocks
    // grow beyond 512 particles for nv > 65 536, where the direct geometry does not exist)
    g.nb1 = (int)std::max<long long>(1, g.direct ? g.nb2 : std::min<long long>((g.nv + 511) / 512, 128));
    if (getenv("SMCMI_E2_NB1")) g.nb1 = std::max(1, std::min(atoi(getenv("SMCMI_E2_NB1")), g.direct ? 64 : 128));   // development only (tools/shard_rank_prof.sh: one rank's share of a larger run)
    g.per1 = ((g.nv + g.nb1 - 1) / g.nb1 + T1 - 1) / T1 * T1;                        // whole passes of the block
    // (one 512-slot tile per gather block up to GRP rows per virtual shard on one handle as on several: a cloud of 4 x odd or 2 x odd particles -
    // 33 .. 64 rows per shard - had two-tile blocks on one handle until round 6, i.e. moment rows summed in another order than its sharded runs')
    g.nbg = (int)std::max<long long>(1, std::min<long long>((g.nv + 511) / 512, g.direct ? GRP : 256));
    g.perg = ((g.nv + g.nbg - 1) / g.nbg + 255) / 256 * 256;
    // (a virtual shard of at most 256 particles: one gather block of ONE 512-slot tile all the same - the block a segment worker is, so that
    // clouds of a few thousand particles resample inside their segments too)
    if (g.direct && g.perg < 512) g.perg = 512;
    if ((long long)g.V * g.nb1 > 1024) return false;
    *out = g;
    return true;
}
// One handle whose particle count has no divisor among 8 / 4 / 2 that leaves it the direct geometry (100 001 particles: one virtual
// shard of 196 rows - engine 1's stage at twice the time): virtual shards of ceil(n / V) particles, the last one shorter (vchunk and
// k2_scan clamp at n; a block beyond the end holds no particle and publishes zero rows).  Only where no sharded run of the same
// cloud shares the canonical order anyway (such a cloud runs on engine 1 today, which has another order).
static bool make_geo2_uneven(const smcmi_handle *h, Geo2 *out) {
    if (h->d > 10 || getenv("SMCMI_E2_REDUCED")) return false;
    for (int V : {8, 4, 2}) {
        Geo2 g{};
        g.N = h->cfg.n_parts; g.n = h->n;
        if (g.n != g.N) return false;
        g.V = V; g.Vl = V; g.v0 = 0; g.nv = (g.n + V - 1) / V;
        if ((long long)(V - 1) * g.nv >= g.n) continue;                              // (no empty virtual shard)
        g.wide = 0; g.t2 = 512;
        g.nb2 = (int)((g.nv + g.t2 - 1) / g.t2);
        if (!(g.nb2 <= GRP && (long long)g.nb2 * V <= 2 * (256 - V2_MAXV))) continue;
        g.direct = 1; g.inker = 1;
        g.nb1 = g.nb2;
        g.per1 = ((g.nv + g.nb1 - 1) / g.nb1 + T1 - 1) / T1 * T1;
        g.nbg = (int)std::max<long long>(1, std::min<long long>((g.nv + 511) / 512, GRP));      // (one tile per gather block, like make_geo2's: in-place selection beyond 32 rows)
        g.perg = ((g.nv + g.nbg - 1) / g.nbg + 255) / 256 * 256;
        if (g.perg < 512) g.perg = 512;
        *out = g;
        return true;
    }
    return false;
}

// The geometry a handle gets (ensure_eng2 builds it, eng2_eligible asks whether engine 2 serves it: ONE rule for both).  `single`: a run of
// smcmi_run on one handle - not a communicator of one rank, whose handle keeps the geometry its larger worlds have.  The uneven cut replaces
// only a geometry that would send the handle to engine 1; a cloud forced onto engine 2 (SMCMI_ENGINE=2) keeps the canonical cut a sharded
// run of the same cloud has, so the file's contract - results do not depend on the number of handles - holds for it.
static bool handle_geo2(const smcmi_handle *h, int world, int rank, bool single, Geo2 *out) {
    static const int eng = getenv("SMCMI_ENGINE") ? atoi(getenv("SMCMI_ENGINE")) : 0;
    if (!make_geo2(h, world, rank, single, out)) return false;
    if (single && world == 1 && !out->wide && !out->direct && eng != 2) { Geo2 gu; if (make_geo2_uneven(h, &gu)) *out = gu; }
    return true;
}
static int ensure_eng2(smcmi_handle *h, int world, int rank, bool single) {
    Geo2 g;
    if (!handle_geo2(h, world, rank, single, &g)) return set_err(SMCMI_ERR_UNSUPPORTED, "engine 2: unsupported shard geometry");
    if (h->e2 && h->e2->world == world && h->e2->g.direct == g.direct && h->e2->g.inker == g.inker && h->e2->g.nb2 == g.nb2 && h->e2->g.v0 == g.v0 && h->e2->g.t2 == g.t2 && h->e2->g.nb1 == g.nb1 && h->e2->g.wide == g.wide && h->e2->g.V == g.V && h->e2->g.nv == g.nv) return 0;
    if (h->e2) { free_eng2(h->e2); h->e2 = nullptr; }
    Eng2 *e = new Eng2();
    e->g = g; e->world = world;
    const int npf = pad2(h->npairs + 2), npp = pad2(h->npairs);
    const size_t n1 = (size_t)g.Vl * g.nb1, n2 = (size_t)g.Vl * g.nb2, ng = (size_t)g.Vl * g.nbg;
    if (dmalloc(&e->d_ctl, 1) || dmalloc(&e->rows_mut, n2 * RMUT) || dmalloc(&e->rows_cm, n1 * npf) || dmalloc(&e->csum, n1) ||
        dmalloc(&e->csum_full, (size_t)g.V * g.nb1) || dmalloc(&e->rows_gm, ng * npp) || dmalloc(&e->rows_pass[0], n1 * 2 * KC) ||
        dmalloc(&e->rows_pass[1], n1 * 2 * KC) || dmalloc(&e->vt_mut, (size_t)g.V * RMUT) || dmalloc(&e->vt_cm, (size_t)g.V * npf) ||
        dmalloc(&e->vt_gm, (size_t)g.V * npp) || dmalloc(&e->vt_pass, (size_t)g.V * 2 * KC) || dmalloc(&e->d_ranges, 2 * V2_MAXV + 2) || dmalloc(&e->d_ranges_all, (size_t)V2_MAXV * (2 * V2_MAXV + 2)) || dmalloc(&e->d_pre, 1) || dmalloc(&e->d_tick, 2 * V2_MAXV * TICK2_STRIDE)) {
        free_eng2(e);
        return SMCMI_ERR_HIP;
    }
    if (g.direct || (g.inker && !g.wide && g.t2 == T3 && g.nb1 == g.nb2)) {       // engine 3 (stage3.hpp) can serve this geometry: tickets, records, time-out words, per-launch stage counts
        const size_t gw = k3_table_words(g.Vl * g.nb2);
        if (dmalloc(&e->d_tick3, 2 * SEG3_TICKS) || dmalloc(&e->d_rec3, REC3_WORDS) || dmalloc(&e->d_to3, 2) || dmalloc(&e->d_done3, SEG3_MAX_LAUNCHES) ||
            dmalloc(&e->d_gran3, gw) || dmalloc(&e->d_sel3, 1)) {
            free_eng2(e);
            return SMCMI_ERR_HIP;
        }
        // (the exit note of a segment: host-mapped; without it the host copies Ctl2 and syncs as for every other launch)
        if (hipHostMalloc(&e->h_note3, 64 + sizeof(Ctl2), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&e->d_note3, e->h_note3, 0) == hipSuccess) {
            memset(e->h_note3, 0, 64 + sizeof(Ctl2));
        } else {
            if (e->h_note3) hipHostFree(e->h_note3);
            e->h_note3 = e->d_note3 = nullptr;
            (void)hipGetLastError();
        }
        HIP_TRY(hipMemsetAsync(e->d_tick3, 0, 2 * SEG3_TICKS * sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(e->d_rec3, 0xFF, REC3_WORDS * sizeof(unsigned long long), h->stream));
        HIP_TRY(hipMemsetAsync(e->d_gran3, 0xFF, gw * sizeof(unsigned long long), h->stream));
        HIP_TRY(hipMemsetAsync(e->d_done3, 0, SEG3_MAX_LAUNCHES * sizeof(int), h->stream));
    }
    HIP_TRY(hipMemsetAsync(e->d_pre, 0, sizeof(Prop2Glob), h->stream));
    HIP_TRY(hipMemsetAsync(e->d_tick, 0, 2 * V2_MAXV * TICK2_STRIDE * sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(e->rows_mut, 0, n2 * RMUT * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(e->rows_cm, 0, n1 * npf * sizeof(double), h->stream));        // (the pad columns stay zero)
    HIP_TRY(hipMemsetAsync(e->rows_gm, 0, ng * npp * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(e->csum, 0, n1 * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(e->csum_full, 0, (size_t)g.V * g.nb1 * sizeof(double), h->stream));
    h->e2 = e;
    return 0;
}

// Engine 2 serves n_para <= 10: one handle while its cloud is small enough for the direct geometry (every block totals the rows
// itself: the latency-bound regime engine 2 was built for), and every multi-handle run (one all-gather of V rows per hand-over,
// results independent of the number of handles).  A single handle with a larger cloud keeps engine 1: its kernels fill the chip
// there and one-block set-up launches are cheap next to them (engine 2's reduced geometry measured 10-15 % behind at N >= 1e6).
// SMCMI_ENGINE=1 / =2 force one engine wherever it can run (development, tests).
// Two chunks per segment worker (one handle of 126 977 .. 253 952 particles) pay where a stage is hand-overs and serial work, not likelihood
// evaluations: α = 1, one block, one MH step, a likelihood that is a handful of flops per datum.  (Measured in round 5: the 10-dim Gaussian at
// 250 000 particles 47.5 against engine 1's 63 µs per stage; config 4 - CAPM, three MH steps - 30.6 against 29.6 ms per run: MH-bound runs stay on engine 1.)
static bool two_chunk_run(const smcmi_handle *h, const smcmi_run_config *rc) {
    auto cheap = [](int fam) { return fam == SMCMI_LIK_GAUSS_ISO || fam == SMCMI_LIK_LINREG || fam == SMCMI_LIK_NONE; };
    return rc && rc->alpha == 1.0 && h->d <= 10 && rc->n_blocks == 1 && rc->n_mh_steps == 1 && cheap(h->h_model.lik[0].family) && cheap(h->h_model.lik[1].family);
}
static bool eng2_eligible(const smcmi_handle *h, int world, bool single, const smcmi_run_config *rc) {
    static const int eng = getenv("SMCMI_ENGINE") ? atoi(getenv("SMCMI_ENGINE")) : 0;
    if (eng == 1 || h->d > 16) return false;
    Geo2 g;
    if (!handle_geo2(h, world, 0, single, &g)) return false;
    // (one handle with more than 256 - V blocks: only the two-chunk segments make engine 2's geometry worth it there)
    if (single && world == 1 && g.direct && (long long)g.nb2 * g.V > 256 - V2_MAXV && !two_chunk_run(h, rc) && eng != 2) return false;
    if (g.wide) {                     // n_para 11 .. 16: the same two-launch stage around the generic mutation body (SMCMI_ENGINE=1: engine 1's stage)
        return true;
    }
    return eng == 2 || world > 1 || !single || g.direct;      // (a communicator of one rank is a sharded run: the measurement vehicle for one rank's share)
}

// the row totals of K1 / K2 are taken by the last block of each virtual shard instead of a k2_reduce launch, while the mutation
// kernel has at most ~4 blocks per CU (the ticket costs every block two barriers and an atomic: 48.1 vs 50.8 µs per stage at
// 125 000 particles per handle, but 52.7 vs 46.7 ms per run at 10⁶ on one handle: beyond 2048 blocks k2_reduce launches total the rows)
static bool fused_tails(const Eng2 *e) {
    // (several handles: up to 2048 blocks - there the tails break even with the launches they replace, 138.5 vs 139.6 µs per stage at
    // 500 000 particles per handle, and they are what lets the peer mailbox replace the all-gathers)
    if (e->g.wide) return true;                  // (the wide kernels' rows are always totalled by the last block of a virtual shard)
    return !e->g.direct && (long long)e->g.Vl * e->g.nb2 <= 2048;
}
#define SMCMI_D_SWITCH(d, CALL)                                                                                                              \
    switch (d) {                                                                                                                          \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;                \
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; case 9: CALL(9); break; case 10: CALL(10); break;              \
    case 11: CALL(11); break; case 12: CALL(12); break; case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break;      \
    default: CALL(16); break;                                                                                                             \
    }

#include "mailbox.hpp"
#include "prof2.hpp"

// Engine 3 serves a single handle in the direct geometry whose blocks are all resident at one per CU; the first use runs the residency
// self-test (k3_census) and a failure - or SMCMI_ENGINE3=0 - leaves the handle on engine 2's launches for good.
static int seg3_time_out_words(smcmi_handle *h, double ms) {
    // (in stream order in front of the launches that read the words; the source is a member of the handle: no sync)
    h->e2->h_to3[0] = 0ull; h->e2->h_to3[1] = (unsigned long long)(ms * 1e5);          // 100 MHz wall clock
    HIP_TRY(hipMemcpyAsync(h->e2->d_to3, h->e2->h_to3, sizeof(h->e2->h_to3), hipMemcpyHostToDevice, h->stream));
    return 0;
}
static int seg3_ready(smcmi_handle *h, bool *ok, bool two_ok = false) {
    Eng2 *e = h->e2;
    *ok = false;
    static const int off = getenv("SMCMI_ENGINE3") ? (atoi(getenv("SMCMI_ENGINE3")) == 0) : 0;
    const Geo2 &g = e->g;
    int n_cu = 0;
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->cfg.device));
    // workers + one gatherer per virtual shard, one CU each; a cloud with more 512-particle blocks than that gives every worker two of them
    // (one handle, a run two_chunk_run admits: `two_ok`)
    const int ch = (g.Vl * g.nb2 + g.Vl <= n_cu) ? 1 : 2;
    if (ch == 2 && !two_ok) return 0;
    const int grid = g.Vl * ((g.nb2 + ch - 1) / ch) + g.Vl;
    if (e->seg_ch != ch) { e->seg_ch = ch; if (e->e3_state > 0) e->e3_state = 0; }      // (another grid: the residency self-test again)
    // (a gatherer totals at most two canonical groups of rows: stage3.hpp gather_vshard)
    if (off || !(g.direct || g.inker) || g.wide || !e->d_rec3 || g.nb1 != g.nb2 || g.nb2 > 2 * GRP || g.per1 != T3 || g.t2 != T3 || grid > n_cu || h->cfg.max_stages >= 65536) return 0;
    if (e->e3_state < 0) return 0;
    if (e->e3_state == 0) {
        int *d_ok = nullptr;
        HIP_TRY(hipMalloc((void **)&d_ok, sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_ok, 0, sizeof(int), h->stream));
        if (int rc = seg3_time_out_words(h, 50.0)) return rc;
        const size_t lds = 96 * 1024;                       // more than half a CU's LDS: one block per CU, the placement the segment kernel must survive
        HIP_TRY(hipFuncSetAttribute((const void *)k3_census, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k3_census<<<grid, T3, lds, h->stream>>>(e->d_tick3, e->d_rec3 + REC3_WORDS - 1, 0xC0FFEEu, e->d_to3, d_ok);
        int okc = 0;
        unsigned long long fl[2] = {1, 0};
        HIP_TRY(hipMemcpyAsync(&okc, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(fl, e->d_to3, sizeof(fl), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        hipFree(d_ok);
        e->e3_state = (okc == grid && fl[0] == 0) ? 1 : -1;
        HIP_TRY(hipMemsetAsync(e->d_tick3, 0, 2 * SEG3_TICKS * sizeof(int), h->stream));
        if (getenv("SMCMI_TRACE")) fprintf(stderr, "[smcmi3] residency self-test: %d of %d blocks, time-out flag %llu -> engine 3 %s\n", okc, grid, fl[0], e->e3_state > 0 ? "on" : "off");
    }
    *ok = e->e3_state > 0;
    return 0;
}

static int run2_impl(ShardGroup &g, const smcmi_run_config *rc, smcmi_result *res) {
    smcmi_handle *h0 = g.hs[0];
    const int nf = h0->h_model.n_free, d = h0->d;
    if (rc->n_blocks < 1 || rc->n_blocks > nf || ((nf + rc->n_blocks - 1) / rc->n_blocks) * (rc->n_blocks - 1) >= nf)
        return set_err(SMCMI_ERR_ARG, "n_blocks incompatible with the number of free parameters");
    if (rc->n_phi < 2 || rc->n_mh_steps < 1) return set_err(SMCMI_ERR_ARG, "bad n_phi / n_mh_steps");
    if (rc->resampling_method != SMCMI_RESAMPLE_SYSTEMATIC && rc->resampling_method != SMCMI_RESAMPLE_MULTINOMIAL)
        return set_err(SMCMI_ERR_ARG, "Invalid resampler in SMC. Options are systematic or multinomial");
    const bool adaptive = !rc->use_fixed_schedule;
    const bool multi = g.world > 1;
    const bool cont = rc->continue_run != 0;
    std::vector<double> sched(rc->n_phi);
    for (int k = 0; k < rc->n_phi; ++k) sched[k] = pow((double)k / (double)(rc->n_phi - 1), rc->lambda);
    // Fixed schedules (the reference's default, src/smc_main.jl:139,386-387): the energy shift of a stage's incremental weights lags the cloud's
    // largest energy by one mutation (stage2.hpp Begin2::e_seen), so that inside a persistent segment a stage is ONE hand-over (stage3.hpp
    // k3_rides).  The rule belongs to the run, not to the engine: launches, segments and any number of handles apply it alike and leave the
    // same bits.  A shift is a common factor of all weights - W, ESS and log-MDD do not depend on it beyond rounding - but a lagged one does not
    // bound the weights by 1: should a stage's sums overflow (the cloud's largest energy grew by more than ~350 / (ϕ_n - ϕ_{n-1}) in one
    // mutation), the run goes on from that stage with the exact shift (below, at the batch's sync).  SMCMI_SHIFT_LAG=0: exact shifts from the
    // start; =<k >= 3> (development): stage k's lagged shift is made to overflow.
    static const int lag_env = getenv("SMCMI_SHIFT_LAG") ? atoi(getenv("SMCMI_SHIFT_LAG")) : 1;
    bool shift_lag = !adaptive && lag_env != 0;
    // ---- per-handle set-up: run parameters and the stage-1 state in DevState (as engine 1), then imported into Ctl2
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        if (!adaptive && rc->n_phi > h->cfg.max_stages) return set_err(SMCMI_ERR_CAPACITY, "max_stages < n_phi");
        const int rank = multi ? (g.rccl ? h->rank : shard_rank(h)) : 0;
        if (int e = ensure_eng2(h, g.world, rank, g.hs.size() == 1 && !g.rccl)) return e;
        if (multi && ensure_shard_buffers(h)) return SMCMI_ERR_HIP;
        // (a fresh run rebuilds the loop state from scratch: only a continuation needs what the device holds - one 70 KB copy and a host
        // round trip less at the start of every run)
        if ((cont && pull_state(h)) || upload_sched(h, sched.data(), rc->n_phi)) return SMCMI_ERR_HIP;
        DevState &s = h->h_st;
        RunParams rp{};
        rp.n_parts = h->cfg.n_parts; rp.n_blocks = rc->n_blocks; rp.n_mh_steps = rc->n_mh_steps; rp.n_phi = rc->n_phi;
        rp.resampling_method = rc->resampling_method; rp.use_fixed_schedule = rc->use_fixed_schedule;
        rp.threshold = rc->threshold_ratio * (double)h->cfg.n_parts;
        rp.alpha = rc->alpha; rp.target = rc->target; rp.tempering_target = rc->tempering_target;
        rp.pw = rc->tempered_update_prior_weight; rp.logp_old = rc->log_prob_old_data;
        rp.max_stages = h->cfg.max_stages; rp.store_history = h->cfg.store_history;
        rp.stall_on_exhaust = 1;
        rp.phi_rtol = rc->phi_rtol > 0.0 ? rc->phi_rtol : (rc->phi_rtol < 0.0 ? 0.0 : DEFAULT_PHI_RTOL);
        rp.stop_stage = rc->stop_after_stage > 0 ? rc->stop_after_stage : 0;
        rp.shift_lag = shift_lag ? std::max(lag_env, 1) : 0;
        if (cont) {
            if (s.stage < 1 || s.stage >= h->cfg.max_stages) return set_err(SMCMI_ERR_STATE, "no loop state to continue from");
            if (s.phi_n >= 1.0) return set_err(SMCMI_ERR_STATE, "the run to continue has already reached phi = 1");
            if (s.stage != h0->h_st.stage) return set_err(SMCMI_ERR_STATE, "shards hold different loop states");
            s.rp = rp; s.done = 0; s.err = 0; s.skip_fold = 1; s.do_resample = 0;
        } else {
            const int cur = s.cur;
            memset(&s, 0, sizeof(DevState));
        s.e_seen = __builtin_nan("");
            s.rp = rp; s.cur = cur;
            s.stage = 1; s.j = 2;                                   // i = 1, j = 2 (smc_main.jl:198-199)
            s.c = rc->c; s.accept = rc->target;                     // initialize_cloud_settings!, initialization.jl:196-211
            s.ess_prev = rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts;
        }
        if (push_state(h)) return SMCMI_ERR_HIP;
        if (!cont) {
            HIP_TRY(hipMemsetAsync(h->rec.resampled, 0, sizeof(int) * h->cfg.max_stages, h->stream));
            if (h->cfg.store_history) {
                HIP_TRY(hipMemsetAsync(h->d_hist_w, 0, sizeof(double) * h->n, h->stream));
                launch_copy_f64(h->d_hist_W, h->cl.buf[0] + (long long)(h->R - 1) * h->n, h->n, h->stream);
            }
        }
        // (the run's first records ride on the import kernel: four 8-byte host copies and a stream sync less per run)
        k2_state<<<1, 64, 0, h->stream>>>(h->d_st, h->e2->d_ctl, 1, h->rec, cont ? 0 : 1, rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts, rc->c, rc->target);
        HIP_TRY(hipMemsetAsync(h->e2->d_tick, 0, 2 * V2_MAXV * TICK2_STRIDE * sizeof(int), h->stream));
        // random numbers drawn ahead: while K1 leaves most CUs idle (small clouds = the direct geometry with 512-thread mutation blocks)
        Eng2 *e = h->e2;
        e->rng_ahead = false; e->z_ahead = 0; e->n_steps = rc->n_mh_steps; e->n_blocks = rc->n_blocks;
        if (e->g.inker && e->g.t2 == 512 && e->g.Vl * e->g.nb1 <= 160) {
            const size_t need = (size_t)h->n * (size_t)(h->d + 2) * (size_t)rc->n_mh_steps * (size_t)rc->n_blocks;
            if (need > h->zbuf_cap) {
                if (h->d_zbuf) { hipFree(h->d_zbuf); h->d_zbuf = nullptr; h->zbuf_cap = 0; }
                if (dmalloc(&h->d_zbuf, need)) return SMCMI_ERR_HIP;
                h->zbuf_cap = need;
            }
            e->rng_ahead = true;
        }
    }
    if (getenv("SMCMI_PROF2") && !h0->e2->d_prof) {
        // (the buffer's layout: stage2.hpp PROF2_*)
        if (dmalloc(&h0->e2->d_prof, PROF2_WORDS)) return SMCMI_ERR_HIP;
        HIP_TRY(hipMemset(h0->e2->d_prof, 0, PROF2_WORDS * sizeof(long long)));
        HIP_TRY(hipDeviceSynchronize());
        h0->e2->prof_stage = atoi(getenv("SMCMI_PROF2"));
    }
    const Geo2 g0 = h0->e2->g;
    const int npf = pad2(h0->npairs + 2), np = pad2(h0->npairs);
    const bool direct = g0.direct != 0, inker = g0.inker != 0;
    // ---- peer mailbox instead of the two all-gathers of a stage: RCCL driver when smcmi_comm_init mapped and tested it everywhere;
    // in-process groups only on request (their kernels share one GPU: a consumer spinning on every CU could starve its producers)
    bool mbox = false;
    // (read at every run: a caller that has validated - or lost confidence in - the transport can switch it between runs)
    const int want = getenv("SMCMI_MAILBOX") ? atoi(getenv("SMCMI_MAILBOX")) : -1;                 // -1: default; 2: also with one rank (tests)
    for (auto *h : g.hs) h->mbox_used = false;
    // large shards (stage2b.hpp): 256-particle mutation blocks, the stage's serial work in helper blocks that take their hand-over from the mailbox
    const bool big = !inker && !g0.wide && d <= 10;
    // (a single handle on that geometry - SMCMI_ENGINE=2, the 1-rank measurement vehicle - posts to itself: the same two-launch stage a rank runs)
    const bool self_mb = big && !multi && want != 0;
    if ((multi || (g.rccl && want == 2) || self_mb) && fused_tails(h0->e2) && npf <= MB_LD) {       // (a mailbox row carries at most MB_LD sums: n_para <= 10)
        if (g.rccl) { if (want != 0) { if (int e = mbox_setup_remote(g)) return e; } mbox = h0->mbox_ok && want != 0; }
        else if (want == 1 || self_mb) { if (int e = mbox_setup_group(g)) return e; mbox = true; }
    }
    static const int big_helpers = getenv("SMCMI_E2_HELPERS") ? atoi(getenv("SMCMI_E2_HELPERS")) : 1;      // development: 0 = k2_begin / k2_prepare as launches
    const bool bighelp = big && mbox && big_helpers != 0;
    if (bighelp) {
        // the first proposals' random numbers are drawn by blocks of K1 that follow the correction blocks onto the CUs and run under the helper
        // block's serial work: as many proposals per particle as fit that window (SMCMI_RNG_AHEAD_PART draws, default 250 000: engine 1's measure)
        static const long long part = getenv("SMCMI_RNG_AHEAD_PART") ? atoll(getenv("SMCMI_RNG_AHEAD_PART")) : 250000;
        for (auto *h : g.hs) {
            Eng2 *e = h->e2;
            const int za = (int)std::min<long long>((long long)rc->n_mh_steps * rc->n_blocks, part / std::max<long long>(1, h->n));
            e->z_ahead = 0;
            if (za < 1) continue;
            HIP_TRY(hipSetDevice(h->cfg.device));
            const size_t need = (size_t)h->n * (size_t)(h->d + 2) * (size_t)za;
            if (need > h->zbuf_cap) {
                if (h->d_zbuf) { hipFree(h->d_zbuf); h->d_zbuf = nullptr; h->zbuf_cap = 0; }
                if (dmalloc(&h->d_zbuf, need)) return SMCMI_ERR_HIP;
                h->zbuf_cap = need;
            }
            e->rng_ahead = true; e->z_ahead = za;
        }
    }
    unsigned mb_cnt[MB_KINDS] = {0, 0};               // counter (-> tag, parity) of the post the next consumer of that kind reads
    unsigned mb_next[MB_KINDS] = {0, 0};              // counters are never reused: a resumed stage posts under fresh tags
    bool mb_live[MB_KINDS] = {false, false};          // the latest rows of that kind were posted to the mailboxes
    std::map<int, unsigned> mb_cm_at, mb_mut_at;      // stage -> counter of its latest K1 / K2 launch (to rewind after a stall)
    if (mbox) {
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            h->mbox_used = true;
            h->mbox_epoch += 1;
            HIP_TRY(hipMemsetAsync(h->d_mbox, 0xFF, sizeof(unsigned long long) * MB_WORDS, h->stream));
            if (int e = mbox_reset_flag(h, h->stream)) return e;
        }
        if (int e = g.barrier()) return e;            // nobody posts before every table is cleared
    }
    auto mb_rows = [&](smcmi_handle *h, int kind, const double *vt, int m) {
        Rows2 r{vt, g0.V, 1, m};
        r.mb = h->d_mbox + mbox_table(kind, mb_cnt[kind]);
        r.to = h->d_mbox + MB_WORDS;
        r.tag = mbox_tag(h->mbox_epoch, mb_cnt[kind]);
        return r;
    };
    auto mb_tail = [&](smcmi_handle *h, int kind, double *vt_slice) {
        Eng2 *e = h->e2;
        Tail2 t{};
        if (!fused_tails(e)) return t;
        t.tick = e->d_tick + kind * V2_MAXV * TICK2_STRIDE; t.vt = vt_slice;
        if (mbox) {
            t.peers = h->d_peers; t.world = g.world; t.gv0 = e->g.v0;
            t.table = mbox_table(kind, mb_cnt[kind]); t.tag = mbox_tag(h->mbox_epoch, mb_cnt[kind]);
        }
        return t;
    };
    // ---- row-set plumbing
    auto view = [&](smcmi_handle *, const double *rows, const double *vt, int nr, int m) {
        return direct ? Rows2{rows, g0.Vl, nr, m} : Rows2{vt, g0.V, 1, m};
    };
    // total this handle's rows per virtual shard and (several handles) all-gather the V x m totals
    auto publish = [&](double *Eng2::*rows, double *Eng2::*vt, int nr, int m, int max_idx, int pair = 0, bool fused = false) -> int {
        if (direct) return 0;
        if (!(fused && fused_tails(h0->e2)))             // (fused: the producing kernel's last blocks wrote the totals)
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            Eng2 *e = h->e2;
            k2_reduce<<<e->g.Vl, RT, 0, h->stream>>>(e->*rows, nr, m, max_idx, e->*vt + (size_t)e->g.v0 * m, pair);
        }
        if (!multi || (fused && mbox)) return 0;        // (mailbox: the tails posted the totals to every handle)
        return g.allgather([=](smcmi_handle *h) { return (const double *)(h->e2->*vt + (size_t)h->e2->g.v0 * m); },
                           [=](smcmi_handle *h) { return h->e2->*vt; }, (size_t)g0.Vl * m);
    };
    auto publish_pass = [&](int slot) -> int {
        if (direct) return 0;
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            Eng2 *e = h->e2;
            k2_reduce<<<e->g.Vl, RT, 0, h->stream>>>(e->rows_pass[slot], e->g.nb1, 2 * KC, -1, e->vt_pass + (size_t)e->g.v0 * 2 * KC, 0);
        }
        if (!multi) return 0;
        return g.allgather([=](smcmi_handle *h) { return (const double *)(h->e2->vt_pass + (size_t)h->e2->g.v0 * 2 * KC); },
                           [=](smcmi_handle *h) { return h->e2->vt_pass; }, (size_t)g0.Vl * 2 * KC);
    };
    auto mut_rows = [&](smcmi_handle *h) { return mb_live[1] ? mb_rows(h, 1, h->e2->vt_mut, RMUT) : view(h, h->e2->rows_mut, h->e2->vt_mut, g0.nb2, RMUT); };
    auto cm_rows = [&](smcmi_handle *h) { return mb_live[0] ? mb_rows(h, 0, h->e2->vt_cm, npf) : view(h, h->e2->rows_cm, h->e2->vt_cm, g0.nb1, npf); };
    auto gm_rows = [&](smcmi_handle *h) { return view(h, h->e2->rows_gm, h->e2->vt_gm, g0.nbg, np); };
    // energy maximum of the initial cloud in the mutation-row layout (stage 2's energy shift)
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        k_energy_max<<<g0.Vl * g0.nb2, TB, 0, h->stream>>>(h->cl, h->d_st, h->e2->rows_mut + RMAX_IDX, RMUT, 0);   // (a maximum: the blocks need not be the rows' particles)
    }
    // (mutation rows of 256-thread blocks are paired: the canonical row stands for 512 particles, whatever the block size)
    if (int e = publish(&Eng2::rows_mut, &Eng2::vt_mut, g0.nb2, RMUT, RMAX_IDX, 0)) return e;

    // ---- engine 3: runs of stages that neither resample nor need a certificate pass become one persistent launch each
    bool e3 = false;
    if (!multi && !g.rccl && g.hs.size() == 1) { if (int e = seg3_ready(h0, &e3, two_chunk_run(h0, rc))) return e; }
    // several handles: the segments span them when the peer mailbox is up (the gatherers post their shard totals into every handle's
    // tables, stage3.hpp Seg3Args::peers) and every handle's grid passed its residency self-test - all ranks must take the same decision
    static const int e3_env = getenv("SMCMI_ENGINE3") ? atoi(getenv("SMCMI_ENGINE3")) : 1;       // 0 off, 1 default, 2 one handle only, 3 also for in-process groups of any size
    const int e3_multi = e3_env == 2 ? 0 : (e3_env == 3 ? 2 : 1);
    // (handles of ONE process share the device's few hardware queues: beyond two of them a handle's persistent launch can sit in a queue
    // in front of the launch it waits for - the in-process group driver, a test vehicle, keeps to launches there; =2 forces segments)
    const bool seg_sys = (multi || g.rccl) && mbox;      // (a one-rank communicator with SMCMI_MAILBOX=2: the measurement vehicle for one rank's share)
    if (seg_sys && e3_multi && (g.hs.size() <= 2 || e3_multi == 2)) {
        double bad = 0.0;
        for (auto *h : g.hs) {
            bool ok = false;
            HIP_TRY(hipSetDevice(h->cfg.device));
            if (int e = seg3_ready(h, &ok)) return e;
            if (!ok) bad += 1.0;
        }
        for (auto *h : g.hs) { HIP_TRY(hipSetDevice(h->cfg.device)); HIP_TRY(hipMemcpyAsync(h->d_comm, &bad, sizeof(double), hipMemcpyHostToDevice, h->stream)); HIP_TRY(hipStreamSynchronize(h->stream)); }
        if (int e = g.allreduce([](smcmi_handle *h) { return h->d_comm; }, 1)) return e;
        HIP_TRY(hipSetDevice(h0->cfg.device));
        HIP_TRY(hipMemcpyAsync(&bad, h0->d_comm, sizeof(double), hipMemcpyDeviceToHost, h0->stream));
        HIP_TRY(hipStreamSynchronize(h0->stream));
        e3 = bad == 0.0;
    }
    std::vector<hipEvent_t> evs3;
    int seg_launches = 0;
    // a stage that must resample does so inside the segment (stage3.hpp SELECTION): one handle, its own tables
    static const int sel_in_env = getenv("SMCMI_SEG_SELECT") ? atoi(getenv("SMCMI_SEG_SELECT")) : 1;      // development: 0 = the segment leaves, selection as launches
    // (the workers' blocks must be the selection kernels' blocks: one 512-slot tile per moment row - not so when a cloud is cut into 2 or 4
    // long virtual shards of more than 32 rows, whose gather blocks take two tiles each)
    // Several handles (sharded segments): the same, with what the handles exchange - chunk sums, the cum column, the ancestors' rows - in the
    // mailbox allocation every peer has mapped (stage3.hpp Sel3Args, mailbox.hpp mbox_sel_words); the worker's scratch holds up to 1 024 chunk
    // sums / chunk ends there (n_para >= 3 at that size)
    const int sel_ncg = (int)((g0.N + SEL_GCH - 1) / SEL_GCH), sel_nch = g0.V * g0.nb1;
    const bool sel_lds = sel_ncg <= 1024 && 16 + 2 * sel_nch + 256 <= (d + 2) * T3 && 16 + sel_ncg + 2 * SEL_GCH <= (d + 2) * T3;      // (the worker's scratch: chunk sums, then chunk ends + staged cum values)
    const bool sel_one = !seg_sys && g.hs.size() == 1 && h0->d_cum != nullptr && sel_lds;
    const bool sel_sys = seg_sys && mbox_sel_words(h0) > 0 && sel_lds && h0->e2->seg_ch == 1;
    // (two chunks per worker - one handle of up to 253 952 particles -: k3_select_two, the chunk in registers through Sel3Args::transit)
    const bool sel_inside = e3 && (sel_one || sel_sys) && d <= 10 && sel_in_env != 0 && g0.nbg == g0.nb2 && g0.perg == T3;
    if (sel_inside)
      for (auto *hh : g.hs) {
        smcmi_handle *h0 = hh;                                     // (shadows: one Sel3Args per handle)
        HIP_TRY(hipSetDevice(h0->cfg.device));
        Eng2 *e = h0->e2;
        const size_t nblk = (size_t)e->g.Vl * e->g.nb2;
        Sel3Args &sl = e->h_sel3;                                  // (a member: the copy needs no sync)
        sl = Sel3Args{};
        sl.method = rc->resampling_method; sl.cum = h0->d_cum; sl.anc = h0->d_anc;
        if (k3_sel_cols(d, rc->alpha == 1.0) == 0 || e->seg_ch == 2) {      // (mixture proposals beyond n_para 7: the particle in transit does not fit the kernel's LDS; two chunks: the one in registers)
            if (!e->d_transit3 && dmalloc(&e->d_transit3, nblk * (size_t)(d + 5) * T3)) return SMCMI_ERR_HIP;
            sl.transit = e->d_transit3;
        }
        sl.g_sel = e->d_gran3 + nblk * (72 + RMUT) * 2 + (size_t)V2_MAXV * (72 + RMUT) * 2; sl.gt_sel = sl.g_sel + nblk * 2 * 2;
        sl.g_gm = sl.gt_sel + (size_t)V2_MAXV * 2 * 2; sl.gt_gm = sl.g_gm + nblk * 72 * 2;
        if (sel_sys) {
            sl.peers = h0->d_peers; sl.mine = h0->d_mbox; sl.world = g.world; sl.chunk0 = e->g.v0 * e->g.nb1; sl.n_loc = h0->n;
            sl.off_cs = MB_SEL_OFF; sl.off_sel = sl.off_cs + MB_SEL_CS_WORDS; sl.off_gm = sl.off_sel + MB_SEL_T_WORDS;
            sl.off_cum = MB_SEL_OFF + MB_SEL_TABLE_WORDS; sl.off_rows = sl.off_cum + h0->cfg.n_parts;
            sl.cum = nullptr;
            // (tags restart with every run of sharded segments: the tables the peers post into start cleared, like the segments' own below)
            HIP_TRY(hipMemsetAsync(h0->d_mbox + MB_SEL_OFF, 0xFF, sizeof(unsigned long long) * MB_SEL_TABLE_WORDS, h0->stream));
        }
        HIP_TRY(hipMemcpyAsync(e->d_sel3, &sl, sizeof(sl), hipMemcpyHostToDevice, h0->stream));
      }
    // (the note outlives a run and sequence numbers start over - sharded segments reset them, 65 535 launches wrap them: a note left by an
    // earlier run must never equal the number a launch of this run is waited for under.  Every run ends with its stream drained, so nothing
    // is in flight that could still write the word)
    if (h0->e2->h_note3) *(volatile int *)h0->e2->h_note3 = -1;
    int force_sel = -1;          // the stage a segment left because it must resample: enqueued with its selection in front of the next segment
    bool status_pending = false; // ... and its status (code 6) is still set: the segment that enters at that stage's mutation clears it
    int last_note_seq = -1;      // sequence number of the latest segment launch that leaves a note; -1: the stream's last launch is not such a segment
    struct SegRange { int a, b; bool enter; };
    std::vector<SegRange> seg_ranges;              // stages each segment launch was enqueued for (error diagnosis)
    if (e3) {
        // (across ranks the first hand-over of a segment also absorbs the skew between the ranks' hosts: a longer bound)
        static const double to_env = getenv("SMCMI_SEG_TIMEOUT_MS") ? atof(getenv("SMCMI_SEG_TIMEOUT_MS")) : 0.0;
        const double to_ms = to_env > 0.0 ? to_env : (seg_sys ? 1000.0 : 200.0);
        for (auto *h : g.hs) { HIP_TRY(hipSetDevice(h->cfg.device)); if (int e = seg3_time_out_words(h, to_ms)) return e; }
        if (seg_sys) {
            // one launch sequence for all handles - the tags (sequence << 16 | stage) must agree - restarted for every run on cleared tables:
            // this handle's rows, and the totals tables every handle posts into
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                Eng2 *e = h->e2;
                e->seg_seq = 0;
                HIP_TRY(hipMemsetAsync(e->d_gran3, 0xFF, k3_table_words(e->g.Vl * e->g.nb2) * sizeof(unsigned long long), h->stream));
                HIP_TRY(hipMemsetAsync(h->d_mbox + MB_SEG_OFF, 0xFF, sizeof(unsigned long long) * 2 * MB_SEG_COPY_WORDS, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
            }
            if (int e = g.barrier()) return e;           // nobody posts before every table is cleared
        }
    }
    const bool profile = rc->use_graph == 2;
    std::vector<hipEvent_t> evs;
    std::vector<int> ev_stage;
    const int dbg = 0;           // (the mutation kernels' ablation bits: retired as a switch in round 6)
    // large shards with helper blocks: the stage whose decision + proposal K1's helper leaves in Prop2Glob (no k2_prepare launch for it), and
    // the stage whose begin the mutation launch in front of it ran (no k2_begin launch for it) with that spec_expected
    int prepared_stage = -1, begun_stage = -1, begun_spec = 0;
    int drawn_stage = -1;        // the stage whose first proposals' random numbers the latest K1 launch drew into zbuf
    // ---- pieces of a stage
    // helper: the launch's extra block takes the hand-over of its own rows and leaves decision + proposal in Prop2Glob (large shards, no selection)
    auto enq_K1 = [&](int n, int begin_done, int spec_expected, bool helper = false) -> int {
        last_note_seq = -1;
        std::vector<Rows2> mrs;
        for (auto *h : g.hs) mrs.push_back(mut_rows(h));                 // (the mutation rows this launch consumes)
        if (mbox) { mb_cnt[0] = ++mb_next[0]; mb_live[0] = true; mb_cm_at[n] = mb_cnt[0]; }   // its own rows go out under a fresh correction tag
        for (size_t k = 0; k < g.hs.size(); ++k) {
            smcmi_handle *h = g.hs[k];
            HIP_TRY(hipSetDevice(h->cfg.device));
            const Rows2 mr = mrs[k];
            const Tail2 tail = mb_tail(h, 0, h->e2->vt_cm + (size_t)h->e2->g.v0 * npf);
            Prep2Args pb{};
            if (helper && bighelp) {
                pb.enable = 1; pb.nb = rc->n_blocks; pb.nf = nf; pb.seed = h->cfg.seed; pb.cmrows = cm_rows(h); pb.rec = h->rec;
                pb.md = h->d_model; pb.out = h->e2->d_pre;
            }
#define SMCMI_CALL(D) launch_k2_correct<D>(h, n, begin_done, spec_expected, mr, tail, pb)
            SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
        }
        if (helper && bighelp) prepared_stage = n;
        if (h0->e2->rng_ahead) drawn_stage = n;
        return publish(&Eng2::rows_cm, &Eng2::vt_cm, g0.nb1, npf, -1, 0, true);
    };
    auto enq_select = [&](int n) -> int {
        last_note_seq = -1;
        if (!multi) {
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                Eng2 *e = h->e2;
                const Rows2 cr = cm_rows(h);
                k2_scan<<<g0.V * g0.nb1, TS, 0, h->stream>>>(e->d_ctl, h->d_st, e->g, n, cr, h->d_wt, e->csum, h->d_cum);
#define SMCMI_CALL(D) launch_k2_gather<D>(h, n, cr, h->d_cum, rc->resampling_method, nullptr, 0, -1)
                SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
            }
            return publish(&Eng2::rows_gm, &Eng2::vt_gm, g0.nbg, np, -1);
        }
        // several handles.  Systematic resampling: only the chunk sums are all-gathered; every handle scans ITS weights
        // into its own cum column (offsets and total from the gathered sums: the values a scan over the whole cloud gives), finds
        // for every handle r which of its rows r's slots can descend from (k2_owner_ranges), the G x 2G table of those ranges is
        // all-gathered (the one host read of a resample stage: the exchange below needs its sizes on the host) and only those
        // rows - with their cum values - travel.  No all-gather of the weight column, no scan of N weights on every handle.
        // Multinomial resampling (and SMCMI_RESAMPLE_EXCHANGE=allgather): every slot can descend from any row - all-gather of the
        // weights and of the shard clouds.
        const size_t nloc = (size_t)h0->n;
        if (int e = g.allgather([](smcmi_handle *h) { return (const double *)h->e2->csum; }, [](smcmi_handle *h) { return h->e2->csum_full; },
                                (size_t)g0.Vl * g0.nb1)) return e;
        static const char *xchg = getenv("SMCMI_RESAMPLE_EXCHANGE");
        const bool a2a = !(xchg && !strcmp(xchg, "allgather")) && rc->resampling_method == SMCMI_RESAMPLE_SYSTEMATIC && !(g.hostc && !h0->hostc.alltoallv);
        bool rs = true;
        std::vector<long long> ranges(2 * (size_t)g.world, -1);            // per needer: the global rows its slots can descend from
        if (a2a) {
            const int tw = 2 * g.world + 2;                               // a handle's table: (lo, hi) per needer, the "stage resamples" flag, pad
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                Eng2 *e = h->e2;
                const int c0 = e->g.v0 * e->g.nb1;
                k2_scan<<<g0.Vl * g0.nb1, TS, 0, h->stream>>>(e->d_ctl, h->d_st, e->g, n, cm_rows(h), h->d_wt, e->csum_full, h->d_cum, c0, c0 + g0.Vl * g0.nb1, h->cfg.gid0);
                k2_owner_ranges<<<1, 64, 0, h->stream>>>(e->d_ctl, h->d_st, n, cm_rows(h), h->d_cum, h->cfg.n_parts, h->n, h->cfg.gid0, g.world, h->cfg.seed,
                                                        e->d_ranges, e->csum_full, c0);
            }
            if (int e = g.allgather([](smcmi_handle *h) { return (const double *)h->e2->d_ranges; }, [](smcmi_handle *h) { return (double *)h->e2->d_ranges_all; }, (size_t)tw)) return e;
            std::vector<long long> all((size_t)g.world * tw);
            HIP_TRY(hipSetDevice(h0->cfg.device));
            HIP_TRY(hipMemcpyAsync(all.data(), h0->e2->d_ranges_all, sizeof(long long) * all.size(), hipMemcpyDeviceToHost, h0->stream));
            HIP_TRY(hipStreamSynchronize(h0->stream));
            rs = all[2 * (size_t)g.world] == 1;            // 0: the device decided not to resample after all (or the stage is a no-op)
            if (rs) {
                for (int s = 0; s < g.world; ++s)
                    for (int r = 0; r < g.world; ++r) {
                        const long long lo = all[(size_t)s * tw + 2 * r], hi = all[(size_t)s * tw + 2 * r + 1];
                        if (lo < 0) continue;
                        if (ranges[2 * r] < 0 || lo < ranges[2 * r]) ranges[2 * r] = lo;
                        if (hi > ranges[2 * r + 1]) ranges[2 * r + 1] = hi;
                    }
                for (int r = 0; r < g.world; ++r)
                    if (ranges[2 * r] < 0) return set_err(SMCMI_ERR_STATE, "sharded resampling: no handle offers ancestors for a handle's slots");
                if (int e = g.exchange_rows(ranges, true)) return e;
            }
        } else {
            if (int e = g.allgather([](smcmi_handle *h) { return (const double *)h->d_wt; }, [](smcmi_handle *h) { return h->d_full_w; }, nloc)) return e;
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                Eng2 *e = h->e2;
                k2_scan<<<g0.V * g0.nb1, TS, 0, h->stream>>>(e->d_ctl, h->d_st, e->g, n, cm_rows(h), h->d_full_w, e->csum_full, h->d_cum_full);
            }
            if (int e = g.allgather([](smcmi_handle *h) { return (const double *)h->cl.buf[0]; }, [](smcmi_handle *h) { return h->d_full_cloud; },
                                    nloc * h0->R)) return e;
        }
        if (rs)
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                const Rows2 cr = cm_rows(h);
                const int me = g.rccl ? h->rank : shard_rank(h);
                const long long s_lo = a2a ? ranges[2 * me] : 0, s_hi = a2a ? ranges[2 * me + 1] : -1;
#define SMCMI_CALL(D) launch_k2_gather<D>(h, n, cr, h->d_cum_full, rc->resampling_method, h->d_full_cloud, s_lo, s_hi)
                SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
            }
        return publish(&Eng2::rows_gm, &Eng2::vt_gm, g0.nbg, np, -1);
    };
    // next_begin: -1 none; 0 / 1: the launch's helper block runs stage n + 1's begin with that spec_expected (large shards with the mailbox)
    auto enq_K2 = [&](int n, int sel_enqueued, int next_begin = -1) -> int {
        last_note_seq = -1;
        std::vector<Rows2> crs;
        for (auto *h : g.hs) crs.push_back(cm_rows(h));                  // (the correction rows this launch consumes)
        if (mbox) { mb_cnt[1] = ++mb_next[1]; mb_live[1] = true; mb_mut_at[n] = mb_cnt[1]; }
        for (size_t hk = 0; hk < g.hs.size(); ++hk) {
            smcmi_handle *h = g.hs[hk];
            HIP_TRY(hipSetDevice(h->cfg.device));
            Eng2 *e = h->e2;
            Mut2Args ma{};
            ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.n = n; ma.sel_enqueued = sel_enqueued; ma.adaptive = adaptive ? 1 : 0;
            ma.cmrows = crs[hk]; ma.gmrows = gm_rows(h); ma.wt = h->d_wt; ma.rows_mut = e->rows_mut;
            ma.zbuf = e->rng_ahead ? h->d_zbuf : nullptr;
            ma.tail = mb_tail(h, 1, e->vt_mut + (size_t)e->g.v0 * RMUT);
            ma.pre = inker ? nullptr : e->d_pre;
            ma.lik[0] = h->h_model.lik[0]; ma.lik[1] = h->h_model.lik[1];
            ma.n_steps = rc->n_mh_steps; ma.store_history = h->cfg.store_history; ma.has_other = h->h_model.has_other_priors;
            ma.alpha = rc->alpha; ma.n_parts = (double)h->cfg.n_parts;
            ma.hist_W = h->d_hist_W; ma.hist_ld = h->n; ma.rec = h->rec; ma.debug = dbg;
            ma.prof = (e->d_prof && n == e->prof_stage) ? e->d_prof + PROF2_K2 : nullptr;
            if (!inker && prepared_stage != n) {                // decision + proposal once, by one block (unless K1's helper block left them)
                Mut2Args mp = ma;
                mp.pre = nullptr;
#define SMCMI_CALL(D) launch_k2_prepare<D>(h, mp, rc->n_blocks)
                SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (profile && h == h0) { hipEventCreate(&e0); hipEventCreate(&e1); evs.push_back(e0); evs.push_back(e1); ev_stage.push_back(n); hipEventRecord(e0, h->stream); }
            if (!inker && !g0.wide) {
                // 256-particle blocks (stage2b.hpp); the drawn-ahead numbers only where this stage's K1 carried the drawing blocks
                Beg2Args bb{};
                if (bighelp && next_begin >= 0) { bb.enable = 1; bb.spec_expected = next_begin; bb.mrows = mb_rows(h, 1, e->vt_mut, RMUT); bb.sched = h->d_sched; bb.rec = h->rec; }
                ma.zbuf = (e->rng_ahead && e->z_ahead > 0 && drawn_stage == n) ? h->d_zbuf : nullptr;
                ma.z_ahead = ma.zbuf ? e->z_ahead : 0;
#define SMCMI_CALL(D) launch_k2b_mutate<D>(h, ma, bb, rc->n_blocks, rc->alpha == 1.0)
                SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
            } else {
#define SMCMI_CALL(D) launch_k2_mutate<D>(h, ma, rc->n_blocks, rc->alpha == 1.0)
            SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
            }
            if (e1) hipEventRecord(e1, h->stream);
        }
        if (bighelp && next_begin >= 0) { begun_stage = n + 1; begun_spec = next_begin; }
        return publish(&Eng2::rows_mut, &Eng2::vt_mut, g0.nb2, RMUT, RMAX_IDX, 0, true);
    };
    // enter_mut: stage n_first's correction (and selection, if sel) were enqueued as launches - the segment enters at its mutation
    auto enq_K3 = [&](int n_first, int n_last, bool enter_mut = false, bool sel = false) -> int {             // one persistent launch for stages n_first .. n_last (stage3.hpp)
        std::vector<Rows2> mrs, crs;
        for (auto *h : g.hs) { mrs.push_back(mut_rows(h)); crs.push_back(cm_rows(h)); }      // (what the segment's first stage consumes)
        if (seg_sys) {
            // the rows these stages leave reach the launches behind the segment as the plain V x m table (Seg3Args::vt_mut_out), not through the mailbox
            mb_live[1] = false;
            for (int q = n_first; q <= n_last; ++q) { mb_mut_at.erase(q); if (!(enter_mut && q == n_first)) mb_cm_at.erase(q); }
        }
        for (size_t hk = 0; hk < g.hs.size(); ++hk) {
        smcmi_handle *h = g.hs[hk];
        Eng2 *e = h->e2;
        HIP_TRY(hipSetDevice(h->cfg.device));
        if (++e->seg_seq >= 0xFFFFu) {                               // tags are (launch << 16 | stage): start over on clean records
            if (seg_sys) return set_err(SMCMI_ERR_CAPACITY, "sharded segments: launch sequence exhausted (65 535 segment launches on one set of handles)");
            HIP_TRY(hipMemsetAsync(e->d_rec3, 0xFF, REC3_WORDS * sizeof(unsigned long long), h->stream));
            HIP_TRY(hipMemsetAsync(e->d_gran3, 0xFF, k3_table_words(e->g.Vl * e->g.nb2) * sizeof(unsigned long long), h->stream));
            e->seg_seq = 1;
            if (e->h_note3) { HIP_TRY(hipStreamSynchronize(h->stream)); *(volatile int *)e->h_note3 = -1; }      // (once in 65 535 launches)
        }
        Mut2Args ma{};
        ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.n = n_first; ma.sel_enqueued = sel ? 1 : 0; ma.adaptive = adaptive ? 1 : 0;
        ma.rows_mut = e->rows_mut; ma.zbuf = nullptr; ma.pre = nullptr;
        ma.cmrows = crs[hk]; ma.gmrows = gm_rows(h); ma.wt = h->d_wt;
        ma.lik[0] = h->h_model.lik[0]; ma.lik[1] = h->h_model.lik[1];
        ma.n_steps = rc->n_mh_steps; ma.store_history = h->cfg.store_history; ma.has_other = h->h_model.has_other_priors;
        ma.alpha = rc->alpha; ma.n_parts = (double)h->cfg.n_parts;
        ma.hist_W = h->d_hist_W; ma.hist_ld = h->n; ma.rec = h->rec; ma.debug = dbg;
        Seg3Args sa{};
        sa.n_first = n_first; sa.n_last = n_last; sa.enter_mut = enter_mut ? 1 : 0; sa.mrows = mrs[hk]; sa.sched = h->d_sched;
        sa.clear_status = (enter_mut && n_first == force_sel && status_pending) ? 1 : 0;
        if (sa.clear_status && hk + 1 == g.hs.size()) status_pending = false;
        const size_t nblk = (size_t)e->g.Vl * e->g.nb2;
        sa.g_cm = e->d_gran3; sa.g_mut = sa.g_cm + nblk * 72 * 2; sa.gt_cm = sa.g_mut + nblk * RMUT * 2; sa.gt_mut = sa.gt_cm + (size_t)V2_MAXV * 72 * 2;
        sa.sel = sel_inside ? e->d_sel3 : nullptr;
        if (seg_sys) {                                               // the totals tables every handle posts into: inside the mailbox allocation
            sa.peers = h->d_peers; sa.world = g.world;
            sa.off_cm = MB_SEG_OFF; sa.off_mut = MB_SEG_OFF + MB_SEG_KIND_WORDS;
            sa.gt_cm = h->d_mbox + sa.off_cm; sa.gt_mut = h->d_mbox + sa.off_mut;
            sa.vt_mut_out = e->vt_mut;
        }
        sa.rec = e->d_rec3;
        sa.note = (!seg_sys && g.hs.size() == 1) ? (int *)e->d_note3 : nullptr; sa.note_seq = (int)e->seg_seq;
        if (sa.note) { last_note_seq = (int)e->seg_seq; }
        sa.tag_base = e->seg_seq << 16; sa.to = e->d_to3; sa.hist_w = h->d_hist_w; sa.hist_ld = h->n;
        sa.done_out = (h == h0 && seg_launches < SEG3_MAX_LAUNCHES) ? e->d_done3 + seg_launches : nullptr;
        sa.prof = (h == h0 && e->d_prof && n_first <= e->prof_stage && e->prof_stage <= n_last) ? e->d_prof + PROF2_SEG0 : nullptr;
        sa.prof_stage = e->prof_stage;
        sa.gprof = sa.prof ? e->d_prof : nullptr;            // (the wall-clock stamps of every block: absolute PROF2_* offsets)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile && sa.done_out) { hipEventCreate(&e0); hipEventCreate(&e1); evs3.push_back(e0); evs3.push_back(e1); hipEventRecord(e0, h->stream); }
        // (the stage counters of all launches of a run are cleared once, in front of its first segment: a fill per launch was 5 µs each)
        if (sa.done_out && seg_launches == 0) HIP_TRY(hipMemsetAsync(e->d_done3, 0, SEG3_MAX_LAUNCHES * sizeof(int), h->stream));
#define SMCMI_CALL(D) launch_k3_segment<D>(h, ma, sa, rc->n_blocks, rc->alpha == 1.0, shift_lag)
        SMCMI_D_SWITCH(d, SMCMI_CALL)
#undef SMCMI_CALL
        HIP_TRY(hipGetLastError());                  // (a rejected launch would otherwise surface as a bogus capacity / time-out error)
        if (e1) hipEventRecord(e1, h->stream);
        }
        ++seg_launches;
        seg_ranges.push_back({n_first, n_last, enter_mut});
        return 0;
    };
    auto enq_passes = [&](int n, int p0, int P) -> int {           // passes p0 .. P-1, then the closing decision
        for (int p = p0; p < P; ++p) {
        last_note_seq = -1;
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                Eng2 *e = h->e2;
                const Rows2 prev = view(h, e->rows_pass[(p + 1) & 1], e->vt_pass, g0.nb1, 2 * KC);
                k2_pass<<<g0.Vl * g0.nb1, T1, 0, h->stream>>>(h->cl, h->d_st, e->d_ctl, e->g, n, p, prev, h->d_sched, e->rows_pass[p & 1]);
            }
            if (int e = publish_pass(p & 1)) return e;
        }
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            Eng2 *e = h->e2;
            const Rows2 prev = view(h, e->rows_pass[(P + 1) & 1], e->vt_pass, g0.nb1, 2 * KC);
            k2_finish<<<1, T1, 0, h->stream>>>(h->d_st, e->d_ctl, n, P, prev, h->d_sched);
        }
        return 0;
    };
    auto enq_begin = [&](int n, int spec_expected = 0) -> int {
        last_note_seq = -1;
        if (begun_stage == n && begun_spec == spec_expected) return 0;      // (the helper block of the mutation launch in front ran it)
        begun_stage = -1;
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            k2_begin<<<1, T1, 0, h->stream>>>(h->d_st, h->e2->d_ctl, n, mut_rows(h), h->d_sched, h->rec, spec_expected);
        }
        return 0;
    };
    // a stage up to (not including) its mutation: what runs as launches in front of a segment that enters at the mutation
    auto enq_stage_front = [&](int n, bool cert, int P, bool sel) -> int {
        if (cert) {
            if (int e = enq_begin(n)) return e;
            if (int e = enq_passes(n, 0, P)) return e;
            if (int e = enq_K1(n, 1, 0)) return e;
        } else if (!inker) {
            if (int e = enq_begin(n, adaptive ? 1 : 0)) return e;
            if (int e = enq_K1(n, 1, 0)) return e;
        } else if (int e = enq_K1(n, 0, adaptive ? 1 : 0)) return e;
        if (sel) { if (int e = enq_select(n)) return e; }
        return 0;
    };
    // a whole stage; cert: certificate passes instead of a predicted ϕ_n (adaptive schedules only)
    // next_begin: what the mutation launch's helper block runs for stage n + 1 (-1: nothing - the batch ends here, or its mode is not known)
    auto enq_stage = [&](int n, bool cert, int P, bool sel, int next_begin = -1) -> int {
        if (cert) {
            if (int e = enq_begin(n)) return e;
            if (int e = enq_passes(n, 0, P)) return e;
            if (int e = enq_K1(n, 1, 0, !sel)) return e;
        } else if (!inker) {               // many blocks per CU: the stage-begin logic once, by one block
            if (int e = enq_begin(n, adaptive ? 1 : 0)) return e;
            if (int e = enq_K1(n, 1, 0, !sel)) return e;
        } else if (int e = enq_K1(n, 0, adaptive ? 1 : 0)) return e;
        if (sel) { if (int e = enq_select(n)) return e; }
        return enq_K2(n, sel ? 1 : 0, next_begin);
    };
    auto read_ctl = [&](Ctl2 *c) -> int {
        HIP_TRY(hipSetDevice(h0->cfg.device));
        if (last_note_seq >= 0 && h0->e2->h_note3) {
            // the batch ends with a segment: its block 0 leaves Ctl2 and its sequence number in host-mapped memory when it is done with them
            // (stage3.hpp k3_leave_note) - no copy, no wait for the stream to drain; a stream that drains without the note (a launch that
            // was rejected) falls through to the copy
            volatile int *note = (volatile int *)h0->e2->h_note3;
            const int want = last_note_seq;
            last_note_seq = -1;
            for (;;) {
                if (note[0] == want) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    memcpy(c, (const char *)h0->e2->h_note3 + 64, sizeof(Ctl2));
                    return 0;
                }
                if (hipStreamQuery(h0->stream) != hipErrorNotReady) break;
            }
            (void)hipGetLastError();
        }
        HIP_TRY(hipMemcpyAsync(c, h0->e2->d_ctl, sizeof(Ctl2), hipMemcpyDeviceToHost, h0->stream));
        HIP_TRY(hipStreamSynchronize(h0->stream));
        if (g.hs.size() > 1)                                          // in-process groups: the other streams are done when the collectives' syncs are
            for (auto *h : g.hs) { HIP_TRY(hipSetDevice(h->cfg.device)); HIP_TRY(hipStreamSynchronize(h->stream)); }
        return 0;
    };
    auto clear_status = [&]() -> int {
        for (auto *h : g.hs) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            HIP_TRY(hipMemsetAsync(&h->e2->d_ctl->status, 0, 2 * sizeof(int), h->stream));      // code, stage
        }
        return 0;
    };

    const int base = cont ? h0->h_st.stage - 1 : 0;           // stages completed before this call
    const int max_iter = (adaptive ? h0->cfg.max_stages : rc->n_phi - 1) - base;
    // stages per host sync (a stalled stage wastes the rest of its batch: two idle launches per stage; with engine 3 a batch end also cuts
    // a segment in two - a write-back and a reload of the cloud - while an idle segment launch costs next to nothing: longer batches)
    const int sync_every = rc->sync_every > 0 ? rc->sync_every : (e3 ? 96 : 32);
    // (default batches double while nothing stalls - a sync in the middle of a run idles the GPU for ~50 µs and cuts a segment in two - and
    // fall back after a stall; the first batch stays short: only its sync tells how many stages are left)
    int cur_sync = sync_every;
    // (with predictions switched off every stage is four launches and a one-stage segment: a stage that runs out of solver passes idles all of
    // them behind it - short batches there, doubling while nothing stalls: 100 000 particles, n_para 1, 29 adaptive stages 7.3 -> 2.7 ms)
    int cert_sync = 8;
    const int solver_passes = rc->solver_passes >= 1 ? rc->solver_passes : DEFAULT_SOLVER_PASSES;
    const int first_passes = std::max(solver_passes, FIRST_SOLVER_PASSES);
    const double N_tot = (double)h0->cfg.n_parts, thr = rc->threshold_ratio * N_tot;
    static const int sel_mode = getenv("SMCMI_NO_SELECT_PREDICT") ? atoi(getenv("SMCMI_NO_SELECT_PREDICT")) : 0;   // development only
    const bool predict_select = adaptive && sel_mode != 1;
    // predicted ϕ_n needs the solver's objective to be the correction's ESS (no prior weight, quirk Q4) and a tolerance to verify against
    const bool spec_ok = adaptive && !getenv("SMCMI_NO_PREDICTOR") && rc->tempered_update_prior_weight == 0.0 && !(rc->phi_rtol < 0.0);
    bool spec_on = spec_ok;
    int last_spec_stall = -100, spec_strikes = 0, last_solver_stall = -100;
    int dyn_P = solver_passes;
    if (rc->solver_passes < 1 && rc->tempering_target < 0.95) dyn_P = 2;
    double pred_ess = cont ? h0->h_st.ess_prev : (rc->initial_ess > 0.0 ? rc->initial_ess : N_tot);
    int pred_rl = cont ? h0->h_st.resampled_last : 0;
    int stall_stage = -1, stall_p = 0, stages_left_est = 1 << 30;
    int launched = 0;
    res->solver_stalls = 0; res->select_stalls = 0; res->spec_stalls = 0; res->shift_fallback_stage = 0;
    Ctl2 c{};
    const auto t0 = std::chrono::steady_clock::now();
    bool finished = false;
    while (!finished) {
        const int room = max_iter - launched;
        const int batch = adaptive ? std::min(std::min(spec_on || !spec_ok ? cur_sync : std::min(cur_sync, cert_sync), std::max(stages_left_est, 4)), room) : room;
        bool stalled = false;
        int seg_a = -1, seg_b = -1;                      // pending segment of engine 3
        bool seg_enter = false, seg_sel = false;         // ... which enters at the mutation of its first stage (corrected / resampled by launches)
        auto flush_seg = [&]() -> int {
            if (seg_a < 0) return 0;
            const int a = seg_a, b2 = seg_b;
            seg_a = seg_b = -1;
            return enq_K3(a, b2, seg_enter, seg_sel);
        };
        for (int b = 0; b < batch; ++b) {
            const int n = base + launched + 2;
            bool sel = true;
            if (predict_select) {
                // ESS this stage will end at (helpers.jl:14-20), with a margin: a wrong "resample" guess only costs two idle launches
                const double ess_bar = rc->tempering_target * (pred_rl ? N_tot : pred_ess);
                const bool rs = ess_bar < thr * (1.0 + 1e-6);
                sel = rs && sel_mode != 2;
                pred_ess = ess_bar; pred_rl = rs ? 1 : 0;
            }
            // stages that follow a resample or have no mutation rows yet (first stage of a run / a continuation) get certificate
            // passes; so does everything once predictions have stopped verifying
            // (resample stages run on the predicted, verified ϕ_n like every other stage since round 4: 10.16 -> 10.00 ms on config 2)
            const bool cert = adaptive && (!spec_on || launched < 2);
            // engine 3 takes every stage that is expected to need neither (fixed schedules: nobody can tell which stage resamples -
            // the segment leaves at the first one that must, code 6, and the host runs that stage through the launches) ...
            if (n == force_sel) { sel = true; if (predict_select) pred_rl = 1; }        // (a segment left at this stage: it must resample)
            if (e3 && !cert && (!sel || !adaptive || sel_inside) && n != force_sel) {
                if (seg_a < 0) { seg_a = n; seg_enter = false; seg_sel = false; }
                seg_b = n;
                ++launched;
                continue;
            }
            if (int e = flush_seg()) return e;
            // ... and the MUTATION of the others: their solver passes, correction and selection run as launches, then a new segment
            // enters at the mutation (what K2 would do) and goes on with the stages behind it
            if (e3) {
                if (int e = enq_stage_front(n, cert, launched < 2 ? first_passes : dyn_P, sel)) return e;
                seg_a = seg_b = n; seg_enter = true; seg_sel = sel;
            } else {
                // large shards: the mutation launch's helper block runs the NEXT stage's begin - in the mode that stage will be enqueued in
                // (the same rules one stage ahead; the last stage of a batch leaves it to a launch: the sync in between may change the mode)
                int next_begin = -1;
                if (bighelp && b + 1 < batch) {
                    const bool cert2 = adaptive && (!spec_on || launched + 1 < 2);
                    next_begin = cert2 ? 0 : (adaptive ? 1 : 0);
                }
                if (int e = enq_stage(n, cert, launched < 2 ? first_passes : dyn_P, sel, next_begin)) return e;
            }
            ++launched;
        }
        if (int e = flush_seg()) return e;
        if (int e = read_ctl(&c)) return e;
        static const int trace = getenv("SMCMI_TRACE") ? atoi(getenv("SMCMI_TRACE")) : 0;                       // development only
        if (trace) {
            const Post2 &tp = c.ps[0].stage >= c.ps[1].stage ? c.ps[0] : c.ps[1];
            fprintf(stderr, "[smcmi2] sync: launched %d  status (code %d, stage %d, err %d)  post (stage %d, phi %.6g, ess %.6g, rs %d)  begin (stage %d, phi %.6g, final %d)  segments %d\n",
                    launched, c.status.code, c.status.stage, c.status.err, tp.stage, tp.phi_n, tp.ess, tp.do_resample, c.bg.stage, c.bg.phi_n, c.bg.final, seg_launches);
        }
        while (c.status.code == 2 || c.status.code == 3 || c.status.code == 4 || c.status.code == 6) {
            const int sn = c.status.stage, code = c.status.code;
            stalled = true;
            // mailbox: a resumed stage posts under fresh tags into tables a slower handle may still be polling for the stalled
            // stage's - every handle must have left the stalled batch first
            if (mbox) {
                if (int e = g.barrier()) return e;
                // the posts that were actually made: stage sn - 1's mutation rows; stage sn's correction rows if only its selection is missing
                const auto it = mb_mut_at.find(sn - 1);
                mb_live[1] = it != mb_mut_at.end();
                if (mb_live[1]) mb_cnt[1] = it->second;
                if (code == 3) { mb_cnt[0] = mb_cm_at[sn]; mb_live[0] = true; }
            }
            for (int &s : ev_stage) if (s >= sn) s = -1;           // the stalled stage's mutation launch and everything behind it were no-ops
            prepared_stage = begun_stage = -1;                     // (nothing a helper block was enqueued for stands: the resumed stage runs on launches)
            const bool resume_in_segment = code == 6 && e3 && !(adaptive && !spec_on);
            if (!resume_in_segment) { if (int e = clear_status()) return e; }
            if (resume_in_segment) {
                // a segment of engine 3 left at this stage (it must resample): nothing of the stage is committed.  The next batch starts
                // with it - correction and selection as launches, then a segment that enters at its mutation and goes on - instead of the
                // whole stage as launches behind a second host sync (fixed schedules, whose resample stages nobody can foresee, pay this at
                // every one of them: 9.2 -> 8.6 ms for 300 fixed stages at N = 1e5)
                force_sel = sn;
                status_pending = true;                             // (the entering segment clears the status: no fill launch in front of the stage)
                res->select_stalls += 1;
                launched = sn - 2 - base;                          // (stage sn itself is the next one to enqueue)
                c.status.code = 0;
                break;
            } else if (code == 6) {
                // ... with certificate passes (predictions switched off): the full path runs it
                const bool cert6 = adaptive && !spec_on;
                if (int e = enq_stage(sn, cert6, first_passes, true)) return e;
                if (cert6) { stall_stage = sn; stall_p = first_passes; }
                res->select_stalls += 1;
            } else if (code == 4) {
                // predicted ϕ_n unusable or not verified: nothing of the stage is committed; run it through the certificate path
                if (int e = enq_stage(sn, true, first_passes, true)) return e;
                stall_stage = sn; stall_p = first_passes;
                res->spec_stalls += 1;
                if (sn - last_spec_stall <= 4) { if (++spec_strikes >= 2) spec_on = false; }
                else spec_strikes = 0;
                last_spec_stall = sn;
            } else if (code == 2) {
                const int had = (sn == stall_stage) ? stall_p : (sn - base <= 3 ? first_passes : dyn_P);
                const int more = 8;
                // (a bracketing search halves its interval at least every pass: 1100 passes exhaust the exponent range of a double - whatever
                // keeps a stage asking for more is not a search any more, and the host must not feed it for ever)
                if (had > 1200) return set_err(SMCMI_ERR_BRACKET, "adaptive tempering solver: the search for phi_n does not terminate (the ESS objective is not a number?)");
                if (int e = enq_passes(sn, had, had + more)) return e;
                if (int e = enq_K1(sn, 1, 0)) return e;
                if (int e = enq_select(sn)) return e;
                if (int e = enq_K2(sn, 1)) return e;
                stall_stage = sn; stall_p = had + more;
                res->solver_stalls += 1;
                if (sn - last_solver_stall <= 4 && dyn_P < 4) ++dyn_P;
                last_solver_stall = sn;
            } else {
                // the stage must resample but its selection kernels were not enqueued: run the rest of it
                if (int e = enq_select(sn)) return e;
                if (int e = enq_K2(sn, 1)) return e;
                res->select_stalls += 1;
            }
            launched = sn - 1 - base;
            if (int e = read_ctl(&c)) return e;
        }
        if (c.status.code == 9 && c.status.err == SMCMI_ERR_NAN_ESS && shift_lag && c.status.stage >= 2) {
            // the sums of stage sn are not numbers under the LAGGED energy shift: nothing of the stage is committed (the state in memory is the
            // one after stage sn - 1's mutation, as for every stage that stalls).  From here on the run shifts by the cloud's current maximum,
            // which bounds every weight by 1 - if the sums still are not numbers, check_nan_ess's error stands (helpers.jl:270-305)
            const int sn = c.status.stage;
            shift_lag = false;
            res->shift_fallback_stage = sn;
            if (mbox) { if (int e = g.barrier()) return e; const auto it = mb_mut_at.find(sn - 1); mb_live[1] = it != mb_mut_at.end(); if (mb_live[1]) mb_cnt[1] = it->second; }
            for (auto *h : g.hs) {
                HIP_TRY(hipSetDevice(h->cfg.device));
                h->h_lag0 = 0;
                HIP_TRY(hipMemcpyAsync(&h->d_st->rp.shift_lag, &h->h_lag0, sizeof(int), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(hipMemsetAsync(&h->e2->d_ctl->status, 0, 4 * sizeof(int), h->stream));      // code, stage, err, pad
            }
            for (int &s : ev_stage) if (s >= sn) s = -1;
            prepared_stage = begun_stage = -1;
            force_sel = -1; status_pending = false;
            launched = sn - 2 - base;
            if (getenv("SMCMI_TRACE")) fprintf(stderr, "[smcmi2] stage %d: sums overflowed under the lagged energy shift - exact shifts from here on\n", sn);
            continue;
        }
        if (c.status.code == 1 || c.status.code == 5 || c.status.code == 9) break;
        const Post2 &p = c.ps[0].stage >= c.ps[1].stage ? c.ps[0] : c.ps[1];
        if (p.phi_n >= 1.0 || launched >= max_iter) {
            // the last enqueued stage reached ϕ = 1 (or the capacity is used up): the next stage's begin folds the last acceptance
            // rate and closes the run
            if (int e = enq_begin(p.stage + 1)) return e;
            if (int e = read_ctl(&c)) return e;
            finished = true;
            break;
        }
        if (c.bg.stage == p.stage && c.bg.phi_n > c.bg.phi_prev && p.phi_n < 1.0) {
            const double left = (1.0 - p.phi_n) / (c.bg.phi_n - c.bg.phi_prev);
            stages_left_est = left < 1e6 ? (int)left + 1 : 1 << 30;
        }
        if (predict_select) { pred_ess = p.ess; pred_rl = p.do_resample; }
        if (rc->sync_every <= 0) cur_sync = stalled ? sync_every : std::min(2 * cur_sync, 4 * sync_every);
        if (!spec_on && spec_ok) cert_sync = stalled ? 8 : std::min(2 * cert_sync, sync_every);
    }
    // (a batch that ended with a segment's exit note was read before the launch had finished: everything behind this line reads what it left)
    HIP_TRY(hipSetDevice(h0->cfg.device));
    HIP_TRY(hipStreamSynchronize(h0->stream));
    if (h0->e2->d_prof) { if (int e = prof2_report(h0, g0, seg_launches)) return e; }
    for (auto *h : g.hs) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        k2_state<<<1, 64, 0, h->stream>>>(h->d_st, h->e2->d_ctl, 0);
        if (pull_state(h)) return SMCMI_ERR_HIP;
        h->last_n_stages = h->h_st.stage;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const DevState &s = h0->h_st;
    res->kernel_ms_mutate = 0.0; res->n_mutate_launches = 0;
    if (profile && !evs.empty()) {
        // event pairs bracket dispatch + kernel: calibrate the fixed part around an empty kernel of the same grid (smcmi_run)
        HIP_TRY(hipSetDevice(h0->cfg.device));
        hipEvent_t c0, c1;
        hipEventCreate(&c0); hipEventCreate(&c1);
        double acc_ms = 0.0;
        int got = 0;
        for (int r = 0; r < 64; ++r) {
            k_fill<<<(unsigned)((h0->n + 255) / 256), 256, 0, h0->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)
            hipEventRecord(c0, h0->stream);
            k_fill<<<(unsigned)((h0->n + 255) / 256), 256, 0, h0->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)
            hipEventRecord(c1, h0->stream);
            hipStreamSynchronize(h0->stream);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c0, c1) == hipSuccess) { acc_ms += ms; ++got; }
        }
        hipEventDestroy(c0); hipEventDestroy(c1);
        const double over = got ? std::max(0.0, acc_ms / got - 0.0025) : 0.0;
        for (size_t k = 0; k + 1 < evs.size(); k += 2) {
            float ms = 0.f;
            if (ev_stage[k / 2] >= 0 && ev_stage[k / 2] <= s.stage && hipEventElapsedTime(&ms, evs[k], evs[k + 1]) == hipSuccess) {
                res->kernel_ms_mutate += std::max(0.0, (double)ms - over);
                res->n_mutate_launches += 1;
            }
        }
    }
    for (hipEvent_t e : evs) hipEventDestroy(e);
    // segments of engine 3: launches, the stages they completed and (profile mode) their HIP-event time
    res->n_segments = seg_launches; res->segment_stages = 0; res->kernel_ms_segments = 0.0;
    // (workers + gatherers; no gatherer where the workers take each other's rows: launch2.hpp launch_k3_seg)
    res->segment_blocks = seg_launches > 0 ? g0.Vl * ((g0.nb2 + h0->e2->seg_ch - 1) / h0->e2->seg_ch) + ((g0.nb2 <= 2 && !seg_sys) ? 0 : g0.Vl) : 0;
    res->segment_state = h0->e2->e3_state; res->segment_timeouts = h0->seg_timeouts;
    if (seg_launches > 0) {
        const int nl = std::min(seg_launches, SEG3_MAX_LAUNCHES);
        std::vector<int> done(nl);
        HIP_TRY(hipSetDevice(h0->cfg.device));
        HIP_TRY(hipMemcpy(done.data(), h0->e2->d_done3, sizeof(int) * nl, hipMemcpyDeviceToHost));
        for (int k = 0; k < nl; ++k) res->segment_stages += done[k];
        int timed_stages = 0;
        for (size_t k = 0; k + 1 < evs3.size(); k += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, evs3[k], evs3[k + 1]) == hipSuccess) { res->kernel_ms_segments += (double)ms; timed_stages += done[k / 2]; }
        }
        if (!evs3.empty()) res->segment_stages = timed_stages;        // (profile mode: the stages behind kernel_ms_segments)
    }
    for (hipEvent_t e : evs3) hipEventDestroy(e);
    res->n_stages = s.stage; res->resamples = s.resamples; res->logmdd = s.logz; res->c = s.c; res->accept = s.accept;
    res->seconds = std::chrono::duration<double>(t1 - t0).count();
    res->solver_passes = s.solver_passes;
    res->paused = (s.done == 5) ? 1 : 0;
    if (mbox) {
        // a hand-over that timed out poisoned the sums with NaN: report THAT, not the NaN-ESS message the poisoned sums lead to
        unsigned long long timed_out = 0;
        for (auto *h : g.hs) {                           // (any handle of an in-process group may be the one whose wait ran out)
            unsigned long long t = 0;
            HIP_TRY(hipSetDevice(h->cfg.device));
            HIP_TRY(hipMemcpy(&t, h->d_mbox + MB_WORDS, sizeof(t), hipMemcpyDeviceToHost));
            timed_out |= t;
        }
        if (timed_out) {
            for (auto *h : g.hs) { h->mbox_ok = false; h->mbox_tried = true; }        // later runs of these handles use the all-gathers
            return set_err(SMCMI_ERR_TIMEOUT, "peer mailbox (flag " + std::to_string(timed_out) + "): a rank's per-stage sums did not arrive within the time-out (SMCMI_MAILBOX_TIMEOUT_MS); "
                                              "the run is void - repeat it (this handle now uses the all-gathers; SMCMI_MAILBOX=0 does so from the start)");
        }
    }
    if (s.err == SMCMI_ERR_TIMEOUT) {
        for (auto *h : g.hs) h->e2->e3_state = -1;       // these handles keep to engine 2's launches from now on
        return set_err(SMCMI_ERR_TIMEOUT, "engine 3: a hand-over inside a persistent stage segment timed out (SMCMI_SEG_TIMEOUT_MS); the run is void - "
                                          "repeat it (this handle now runs every stage as launches; SMCMI_ENGINE3=0 does so from the start)");
    }
    if (s.err == SMCMI_ERR_NAN_ESS) {
        // the failing correction's unnormalised weights: the scratch column K1 wrote - unless the stage ran inside a segment (registers only)
        const int sn = c.status.stage;
        bool in_seg = false;
        for (const SegRange &r : seg_ranges) in_seg |= (sn >= r.a && sn <= r.b && !(r.enter && sn == r.a));
        if (in_seg && sn >= 2 && sn <= h0->cfg.max_stages) {
            double ph[2] = {0.0, 0.0};
            HIP_TRY(hipSetDevice(h0->cfg.device));
            HIP_TRY(hipMemcpy(ph, h0->rec.phi + (sn - 2), sizeof(ph), hipMemcpyDeviceToHost));
            return nan_ess_error_from_cloud(h0, ph[1], ph[0], rc->tempered_update_prior_weight, rc->log_prob_old_data);
        }
        return nan_ess_error(h0, h0->d_wt);
    }
    if (s.err) return err_from_state(s.err);
    if (!(c.status.code == 1 || c.status.code == 5)) return set_err(SMCMI_ERR_CAPACITY, "max_stages exceeded before the tempering schedule reached 1");
    return 0;
}
