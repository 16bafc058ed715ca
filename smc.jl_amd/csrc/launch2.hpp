// launch2.hpp - host launchers of the engine 2 kernels that are templates over n_para (stage2.hpp).  Each n_para is compiled in its
// own translation unit (inst2.hip with -DSMCMI_INST_D=<d>), so the sixty-odd instantiations build in parallel; every other
// translation unit sees the launchers as extern templates and instantiates none of the kernels.
#pragma once
#include "handle.hpp"

// pb.enable: the launch carries the helper block of stage2b.hpp (decision + proposal from this launch's own totals; large shards with the mailbox)
template <int D>
void launch_k2_correct(smcmi_handle *h, int n, int begin_done, int spec_expected, const Rows2 &mrows, const Tail2 &tail, const Prep2Args &pb) {
    Eng2 *e = h->e2;
    Rng2 ra{};
    unsigned grid = (unsigned)(e->g.Vl * e->g.nb1);
    const bool helper = pb.enable && tail.tick && D <= 10;
    if (e->rng_ahead) {              // extra blocks, one per mutation block, draw the stage's random numbers on the idle CUs
        ra.zbuf = h->d_zbuf; ra.n_steps = e->n_steps; ra.nb = e->n_blocks; ra.nf = h->h_model.n_free; ra.seed = h->cfg.seed; ra.gid0 = h->cfg.gid0;
        ra.t_lim = e->z_ahead;
        // (large shards: the drawing blocks follow the correction blocks onto the CUs - one 512-thread block each - and run under the helper's serial work)
        if (e->g.inker) grid += (unsigned)std::max(1, std::min(e->g.Vl * e->g.nb2, 256 - (int)grid));
        else grid += (unsigned)std::max<long long>(1, std::min<long long>((e->g.n + T1 - 1) / T1, 256));
    }
    if (helper) grid += 1;
    const size_t lds = helper ? k2_lds_bytes(D) : 0;
    Prep2Args pa = pb;
    pa.enable = helper ? 1 : 0;
    if (tail.tick) k2_correct<D, true><<<grid, T1, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, e->g, n, begin_done, spec_expected, mrows, h->d_sched, h->rec,
                                                           e->rows_cm, e->csum, h->d_wt, h->d_hist_w, h->n, ra, tail, pa, (e->d_prof && n == e->prof_stage) ? e->d_prof : nullptr);
    else k2_correct<D, false><<<grid, T1, 0, h->stream>>>(h->cl, h->d_st, e->d_ctl, e->g, n, begin_done, spec_expected, mrows, h->d_sched, h->rec,
                                                           e->rows_cm, e->csum, h->d_wt, h->d_hist_w, h->n, ra, tail, pa, (e->d_prof && n == e->prof_stage) ? e->d_prof : nullptr);
}
template <int D>
void launch_k2_gather(smcmi_handle *h, int n, const Rows2 &cmrows, const double *cum, int method, const double *full, long long s_lo, long long s_hi) {
    Eng2 *e = h->e2;
    if constexpr (D <= 10)
        k2_gather<D><<<e->g.Vl * e->g.nbg, TS, 0, h->stream>>>(h->cl, e->d_ctl, h->d_st, e->g, n, cmrows, cum, method, h->cfg.seed, h->cfg.gid0, h->d_anc, full,
                                                              h->n, e->rows_gm, s_lo, s_hi);
    else
        k2_gather_wide<D><<<e->g.Vl * e->g.nbg, TB, 0, h->stream>>>(h->cl, e->d_ctl, h->d_st, e->g, n, cmrows, cum, method, h->cfg.seed, h->cfg.gid0, h->d_anc, full,
                                                                   h->n, e->rows_gm, s_lo, s_hi);
}
template <int D>
void launch_k2_mutate(smcmi_handle *h, const Mut2Args &ma, int nb, bool alpha1) {
    Eng2 *e = h->e2;
    if constexpr (D > 10) {          // the generic mutation body behind K2's prologue (stage2.hpp k2w_mutate); e->g.wide = lanes per particle
        const unsigned gw = (unsigned)(e->g.Vl * e->g.nb2);
        if (!e->wide_attr_set) {     // (more than the default 64 KB of dynamic LDS per block: the per-particle vectors of 256 particles)
            hipFuncSetAttribute((const void *)k2w_mutate<D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2w_lds_bytes(D, 1));
            if constexpr (D == 13) hipFuncSetAttribute((const void *)k2w_mutate<13, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2w_lds_bytes(13, 4));
            e->wide_attr_set = true;
        }
        if constexpr (D == 13) {
            if (e->g.wide == 4) { k2w_mutate<13, 4><<<gw, 256, k2w_lds_bytes(13, 4), h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free); return; }
        }
        k2w_mutate<D, 1><<<gw, 256, k2w_lds_bytes(D, 1), h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free);
        return;
    } else {
    const size_t lds = k2_lds_bytes(D);
    const unsigned grid = (unsigned)(e->g.Vl * e->g.nb2);
    // (every block runs the prologue: the direct geometry and small shards of several handles; large shards are k2b_mutate's - launch_k2b_mutate)
    if (!ma.tail.tick) {             // the direct geometry (config 2): no hand-over code in the instantiation
        if (alpha1) k2_mutate<D, true, 512, false><<<grid, 512, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free);
        else k2_mutate<D, false, 512, false><<<grid, 512, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free);
    } else {
        if (alpha1) k2_mutate<D, true, 512, true><<<grid, 512, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free);
        else k2_mutate<D, false, 512, true><<<grid, 512, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, nb, h->h_model.n_free);
    }
    }
}
// the mutation launch of large shards (stage2b.hpp), compiled in translation units of its own (inst2b.hip, Makefile BIGFLAGS)
template <int D>
void launch_k2b_mutate(smcmi_handle *h, const Mut2Args &ma, const Beg2Args &bb, int nb, bool alpha1) {
    Eng2 *e = h->e2;
    if constexpr (D <= 10) {
        const size_t lds = k2_lds_bytes_body(D);          // (the prologue's scratch behind it is never touched: 13 instead of 22 KB per block)
        const unsigned grid = (unsigned)(e->g.Vl * e->g.nb2) + (bb.enable ? 1u : 0u);
        if (alpha1) k2b_mutate<D, true><<<grid, T2B, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, bb, nb, h->h_model.n_free);
        else k2b_mutate<D, false><<<grid, T2B, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, bb, nb, h->h_model.n_free);
    }
}
template <int D>
void launch_k2_prepare(smcmi_handle *h, const Mut2Args &mp, int nb) {
    Eng2 *e = h->e2;
    if constexpr (D <= 10) k2_prepare<D><<<1, 256, k2_lds_bytes(D), h->stream>>>(h->d_st, e->d_ctl, h->d_model, mp, nb, h->h_model.n_free, e->d_pre);
}
// one instantiation per (n_para, α = 1?, riding?): the proposal kinds are compiled with different flags (Makefile SEGFLAGS / SEGFLAGS_MIX).
// RIDE: fixed schedules under RunParams::shift_lag on one handle - a stage's correction row rides the mutation row in front of it (stage3.hpp k3_rides)
// CH = 2: two 512-particle chunks per worker (α = 1, one handle of up to 253 952 particles: run2.hpp seg3_ready), translation units of their own (inst3c / inst3cr)
// SYS: several handles (Seg3Args::peers set): translation units of their own as well (inst3s / inst3ms / inst3rs / inst3mrs)
template <int D, bool A1, bool RIDE, int CH = 1, bool SYS = false>
void launch_k3_seg(smcmi_handle *h, const Mut2Args &ma, const Seg3Args &sa, int nb) {
    Eng2 *e = h->e2;
    if constexpr (D <= 10 && CH == 2) {
        static_assert(A1, "two chunks per worker: the α = 1 kernel only");
        const size_t lds2 = k3_lds_bytes(D, D + 6);             // (the parked chunk: D + 6 columns where the one-chunk kernel keeps the particle in transit)
        const unsigned grid2 = (unsigned)(e->g.Vl * ((e->g.nb2 + 1) / 2) + e->g.Vl);
        constexpr int bit2 = RIDE ? 32 : 16;
        if (!(e->seg_attr_set & bit2)) {
            hipFuncSetAttribute((const void *)k3_segment<D, true, RIDE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((lds2 + 1023) / 1024 * 1024));
            e->seg_attr_set |= bit2;
        }
        k3_segment<D, true, RIDE, 2><<<grid2, T3, lds2, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, sa, nb, h->h_model.n_free);
    } else if constexpr (D <= 10) {
    // (a gatherer stages its shard's rows in the dynamic LDS: beyond GRP rows per virtual shard - several handles with 32 769 .. 65 536
    // particles per virtual shard - the small-n_para and the large mixture kernels' allocation would not hold them)
    const size_t lds = std::max(k3_lds_bytes(D, k3_sel_cols(D, A1)), e->g.nb2 > GRP ? k3_gather_lds_bytes(D) : (size_t)0);
    // workers + one gatherer per virtual shard.  One handle whose virtual shards are one or two blocks (stage3.hpp rows_direct / rows_two): the
    // workers take each other's rows themselves and nobody reads a gatherer's totals - none is launched (a gatherer that nothing waits for has
    // no flow control: one stage behind, it would poll for a tag its rows have already left and raise the waits' abort word)
    const bool gatherers = !(e->g.nb2 <= 2 && sa.peers == nullptr);
    const unsigned grid = (unsigned)(e->g.Vl * e->g.nb2 + (gatherers ? e->g.Vl : 0));
    constexpr int attr_bit = ((A1 ? 1 : 2) << (RIDE ? 2 : 0)) << (SYS ? 6 : 0);
    if (!(e->seg_attr_set & attr_bit)) {       // (per handle = per device: a function attribute belongs to the device's copy of the kernel)
        // (opt in to more than the default 64 KB per block: the kernel's static arrays come on top of `lds`; a CU has 160 KB)
        hipFuncSetAttribute((const void *)k3_segment<D, A1, RIDE, 1, SYS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((std::max(k3_lds_bytes(D, k3_sel_cols(D, A1)), k3_gather_lds_bytes(D)) + 1023) / 1024 * 1024));
        e->seg_attr_set |= attr_bit;
    }
    k3_segment<D, A1, RIDE, 1, SYS><<<grid, T3, lds, h->stream>>>(h->cl, h->d_st, e->d_ctl, h->d_model, e->g, ma, sa, nb, h->h_model.n_free);
    }
}
template <int D>
inline void launch_k3_segment(smcmi_handle *h, const Mut2Args &ma, const Seg3Args &sa, int nb, bool alpha1, bool ride) {
    if (h->e2->seg_ch == 2) {             // (seg3_ready grants two chunks to α = 1 runs only)
        if (ride) launch_k3_seg<D, true, true, 2>(h, ma, sa, nb); else launch_k3_seg<D, true, false, 2>(h, ma, sa, nb);
        return;
    }
    if (sa.peers != nullptr) {            // several handles
        if (alpha1) { if (ride) launch_k3_seg<D, true, true, 1, true>(h, ma, sa, nb); else launch_k3_seg<D, true, false, 1, true>(h, ma, sa, nb); }
        else { if (ride) launch_k3_seg<D, false, true, 1, true>(h, ma, sa, nb); else launch_k3_seg<D, false, false, 1, true>(h, ma, sa, nb); }
        return;
    }
    if (alpha1) { if (ride) launch_k3_seg<D, true, true>(h, ma, sa, nb); else launch_k3_seg<D, true, false>(h, ma, sa, nb); }
    else { if (ride) launch_k3_seg<D, false, true>(h, ma, sa, nb); else launch_k3_seg<D, false, false>(h, ma, sa, nb); }
}

#define SMCMI_LAUNCH2_INSTANCES(X, D)                                                                                              \
    X template void launch_k2_correct<D>(smcmi_handle *, int, int, int, const Rows2 &, const Tail2 &, const Prep2Args &);            \
    X template void launch_k2_gather<D>(smcmi_handle *, int, const Rows2 &, const double *, int, const double *, long long, long long); \
    X template void launch_k2_mutate<D>(smcmi_handle *, const Mut2Args &, int, bool);                                                \
    X template void launch_k2_prepare<D>(smcmi_handle *, const Mut2Args &, int);
// the persistent segment kernel lives in translation units of its own (inst3.hip, one per n_para and proposal kind) with their own flags
// (Makefile SEGFLAGS / SEGFLAGS_MIX).  History of the α = 1 variant at n_para 10: default flags 256 VGPRs / 216 B of scratch per lane, 36.4 µs per
// stage; hoisted computations sunk back into the stage loop (-sink-insts-to-avoid-spills) 72 B, 35.6 µs; no machine LICM instead 229 VGPRs,
// no spill, 35.2 µs (both together: 36.6).  The mixture variant needs both (251 VGPRs, no spill, 42.5 µs; LICM off alone: 77 spilled
// registers, 45.3 µs).  K1 / K2 and the generic mutation kernels lose 1-2 % to the sinking and are indifferent to the LICM switch.
#define SMCMI_LAUNCH3_INSTANCES(X, D) X template void launch_k3_seg<D, true, false>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, false, false>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, true, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, false, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, true, false, 2>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, true, true, 2>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, true, false, 1, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, false, false, 1, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, true, true, 1, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int); \
                                      X template void launch_k3_seg<D, false, true, 1, true>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int);
#define SMCMI_LAUNCH3_ONE(D, A, R, C, S) template void launch_k3_seg<D, A, R, C, S>(smcmi_handle *, const Mut2Args &, const Seg3Args &, int);
#define SMCMI_LAUNCH2B_INSTANCES(X, D) X template void launch_k2b_mutate<D>(smcmi_handle *, const Mut2Args &, const Beg2Args &, int, bool);
#define SMCMI_LAUNCH_ALL_D(M, X)                                                                                                          \
    M(X, 1) M(X, 2) M(X, 3) M(X, 4) M(X, 5) M(X, 6) M(X, 7) M(X, 8) M(X, 9) M(X, 10) M(X, 11) M(X, 12) M(X, 13) M(X, 14) M(X, 15) M(X, 16)
// (the large-shard mutation kernel and the segment kernel exist for n_para <= 10: beyond that the launchers are empty and instantiated where they are called)
#define SMCMI_LAUNCH_B_D(M, X) M(X, 1) M(X, 2) M(X, 3) M(X, 4) M(X, 5) M(X, 6) M(X, 7) M(X, 8) M(X, 9) M(X, 10)
#if defined(SMCMI_INST_D)
SMCMI_LAUNCH2_INSTANCES(, SMCMI_INST_D)
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH3_INSTANCES, extern)
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH2B_INSTANCES, extern)
#elif defined(SMCMI_INST3_D)
SMCMI_LAUNCH_ALL_D(SMCMI_LAUNCH2_INSTANCES, extern)
SMCMI_LAUNCH3_ONE(SMCMI_INST3_D, (SMCMI_INST3_A != 0), (SMCMI_INST3_R != 0), SMCMI_INST3_C, (SMCMI_INST3_S != 0))
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH2B_INSTANCES, extern)
#elif defined(SMCMI_INST2B_D)
SMCMI_LAUNCH_ALL_D(SMCMI_LAUNCH2_INSTANCES, extern)
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH3_INSTANCES, extern)
SMCMI_LAUNCH2B_INSTANCES(, SMCMI_INST2B_D)
#else
SMCMI_LAUNCH_ALL_D(SMCMI_LAUNCH2_INSTANCES, extern)
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH3_INSTANCES, extern)
SMCMI_LAUNCH_B_D(SMCMI_LAUNCH2B_INSTANCES, extern)
#endif
