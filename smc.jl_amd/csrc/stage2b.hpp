// stage2b.hpp - engine 2 on LARGE shards (more than one 512-particle block per CU: > 131 072 particles per handle, n_para <= 10).
//
// What a rank of a 2- or 4-GPU run of config 3 holds (500 000 / 250 000 particles) ran through four launches per stage - k2_begin (one
// block), K1, k2_prepare (one block), K2 = k2_mutate<D, a, 256, true> at 168 VGPRs, three wavefronts per SIMD: 977 blocks on 768 slots,
// two rounds where engine 1's k_mutate_reg (127 VGPRs, 1 024 slots) needs one, and the block that totalled a shard's 489 paired rows
// took ~10 µs doing so at the end of every launch - 88 / 116 µs per stage against 131 µs for the WHOLE 10^6
// cloud on one GPU.  Here the stage is two launches whose blocks never wait, with one helper block each that does:
//
//   K1  = k2_correct<D, true>: correction blocks (rows totalled per virtual shard by Tail2, posted into every handle's mailbox)
//         + ONE helper block (k2_prepare_block, stage2.hpp): takes the V x m correction totals from the mailbox, decides, builds the
//           proposal, leaves it in Prop2Glob; the stage's bookkeeping after that
//         + blocks that draw the first proposal's random numbers (Rng2) on the CUs the correction blocks have left - they run under the
//           helper's serial work, where the chip is otherwise idle
//   K2b = k2b_mutate<D, a> below: the register-resident mutation of k2_mutate (k2_mh_steps: same arithmetic, same rows, same bits) with
//         nothing else in the instantiation - 4 wavefronts per SIMD - reading Prop2Glob and the drawn-ahead numbers
//         + ONE helper block: takes the V x 34 mutation totals from the mailbox and runs the NEXT stage's begin (acceptance fold, energy
//           shift, phi predictor / schedule) into Begin2, so the next K1 starts on a decided phi_n
//
// Only the two helper blocks ever poll: handles that share a GPU (in-process groups, several processes on one box: the tests) cannot
// starve each other the way blocks that all spin could.  Without the mailbox (all-gather fall-back) the helpers are off and k2_begin /
// k2_prepare run as launches around the same K2b.  Reference: the particle loop of src/smc_main.jl:472-476 behind :377-469.
#pragma once
#include "stage2.hpp"

namespace smcmi {

struct Beg2Args {
    int enable;                   // the launch carries the helper block (block Vl * nb2) that runs stage n + 1's begin
    int spec_expected;            // ... on the predicted phi (adaptive schedules outside certificate stages)
    Rows2 mrows;                  // this launch's mutation totals as the mailbox delivers them
    const double *sched;
    Records rec;
};

template <int T>
__device__ inline void k2b_begin_block(int n_next, DevState *st, Ctl2 *ctl, const Beg2Args &bb) {
    __shared__ Post2 s_po;
    __shared__ Begin2 s_bg;
    __shared__ double s_vt[V2_MAXV * RMUT], s_tot[RMUT], s_sw[64];
    __shared__ int s_act;
    begin2_block<T, true>(n_next, st, ctl, bb.mrows, bb.spec_expected, bb.sched, bb.rec, &s_po, &s_bg, s_vt, s_tot, s_sw, &s_act, nullptr, false, 1);
}

constexpr int T2B = 512;          // threads = particles of a block: one canonical mutation row (block_reduce_es2<8>), two blocks per CU
// One 512-particle block of the mutation (src/mutation.jl:56-138, helpers.jl:87-164): the `ma.pre` path of k2_mutate and nothing else -
// 128 VGPRs (alpha = 1), i.e. 4 wavefronts per SIMD.
template <int D, bool ALPHA1>
__global__ void __launch_bounds__(T2B, ALPHA1 ? 2 : 1) k2b_mutate(CloudPtrs cl, DevState *st, Ctl2 *ctl, const ModelDev *md, Geo2 g, Mut2Args ma,
                                                                   Beg2Args bb, int nb, int nf) {
SMCMI_FP_CONTRACT
    constexpr int T = T2B;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, n = ma.n;
    if ((int)blockIdx.x >= g.Vl * g.nb2) {
        if (bb.enable) k2b_begin_block<T>(n + 1, st, ctl, bb);
        return;
    }
    const Mut2Lds<D> L(sm);
    K2_STAMP(ma.prof, 0);
    // (development, SMCMI_PROF2: every block's start / end on the 100 MHz wall clock -> the residency census run2_impl prints)
    if (ma.prof != nullptr && tid == 0 && blockIdx.x < PROF2_BLOCKS) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        ma.prof[(PROF2_CENSUS - PROF2_K2) + 3 * blockIdx.x] = wall_clock64();
        ma.prof[(PROF2_CENSUS - PROF2_K2) + 3 * blockIdx.x + 2] = (long long)(((xcc & 0xFu) << 16) | (hw & 0xFF00u));        // (XCC | SE, SH, CU: the CU the block sits on)
    }
    const double nrm_N = ma.n_parts;
    const int nrm_hist = ma.store_history;
    const LikDev &ld0 = ma.lik[0], &ld1 = ma.lik[1];
    double *red = L.red, *l_dat = L.l_dat, *Lraw = L.Lraw, *logdet_s = L.logdet_s;
    double *mub_raw = L.mub_raw, *sdd_raw = L.sdd_raw, *sdn_raw = L.sdn_raw;
    int *bptr_s = L.bptr_s, *loff_s = L.loff_s, *ball_raw = L.ball_raw;
    // ---- the particle (speculatively from buffer 0: only resample stages read the gathered cloud in buffer 1) is requested before anything
    // is waited for: its columns and the stage's decision arrive under one round trip
    long long beg, end;
    vchunk(g, blockIdx.x / g.nb2, blockIdx.x % g.nb2, T, beg, end);
    const long long i = beg + tid;
    const bool live = i < end;
    const long long il = live ? i : (end > beg ? end - 1 : 0);           // unconditional loads (clamped row)
    const unsigned long long pid = (unsigned long long)(ma.gid0 + i);
    double like, lprior, like_prev, accept = 0.0;
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = col(cl, 0, k)[il];
    like = col(cl, 0, D)[il]; lprior = col(cl, 0, D + 1)[il]; like_prev = col(cl, 0, D + 2)[il];
    const double wt_i = ma.wt[il];
    // ---- decision and proposal (k2_prepare / K1's helper block): model constants + the proposal's arrays into LDS, one barrier
    const Prop2Glob *G = ma.pre;
    const int pstage = G->stage, pgo = G->go;
    const int rs = G->rs;
    const double nrm_sumw = G->s1, phi_n = G->phi_n, e_center = G->e_center;
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
    K2_STAMP(ma.prof, 1);
    if (rs) {                       // the resampled cloud is in buffer 1 (k2_gather)
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = col(cl, 1, k)[il];
        like = col(cl, 1, D)[il]; lprior = col(cl, 1, D + 1)[il]; like_prev = col(cl, 1, D + 2)[il];
    }
    double step_prob = 0.0, uc = 0.0, z[D];        // (every proposal's numbers are loaded or drawn inside the MH loop: nothing is carried into it)
#pragma unroll
    for (int e = 0; e < D; ++e) z[e] = 0.0;
    ModelView mv{D, L.m_fix, L.m_fam, L.m_lo, L.m_hi, L.m_a, L.m_b, L.m_k};
    LikView lv[2];
    k2_stage_lik<T>(ld0, ld1, L.l_par, l_dat, lv);
    __syncthreads();
    // (development stamps, SMCMI_PROF2=<stage>: [1, 2] Prop2Glob + particle + likelihood data in, [8, 9] MH steps, [9, 10] stores + row, [10, 11] tail)
    for (int q = 2; q <= 8; ++q) K2_STAMP(ma.prof, q);
    double w_part = 0.0;
    if (live) {
        w_part = rs ? 1.0 : (wt_i * nrm_N) / nrm_sumw;                      // W·N then /ΣW̃, two roundings like the reference (particle.jl:362-366)
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
    k2_mh_steps<D, ALPHA1, T, false, true>(L, mixbuf, mixpos, mixzt, ma, g.n, lv, mv, nb, nf, live, i, pid, (unsigned)n, phi_n, x, like, lprior, like_prev,
                                           accept, step_prob, uc, z);
    K2_STAMP(ma.prof, 9);
    double acc_val = 0.0;
    if (live) {
        // (a particle that accepted nothing still holds what buffer 0 holds - not after a resample: it came from buffer 1)
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
    // ---- this block's row for the next stage's begin; the last block of a virtual shard totals the shard's rows - all groups of 64 rows
    // in ONE batch of loads (RMUT columns: seven groups fit the block) - and posts them
    k2_mut_row<T>(ma.rows_mut + (long long)blockIdx.x * RMUT, ma.adaptive != 0, like, like_prev, w_part, acc_val, e_center, live, rs != 0, l_dat, red,
                  ma.tail.tick != nullptr);
    K2_STAMP(ma.prof, 10);
    tail_reduce<T, RMUT>(ma.tail, ma.rows_mut, (int)blockIdx.x / g.nb2, g.nb2, RMUT, RMAX_IDX, 0);
    K2_STAMP(ma.prof, 11);
    if (ma.prof != nullptr && tid == 0 && blockIdx.x < PROF2_BLOCKS) ma.prof[(PROF2_CENSUS - PROF2_K2) + 3 * blockIdx.x + 1] = wall_clock64();
}

}  // namespace smcmi
