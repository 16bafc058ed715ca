// handle.hpp - the opaque handle of include/smcmi.h (one particle shard on one GPU) and engine 2's per-handle buffers.
// Shared by the translation units of libsmcmi.so: smcmi.hip (C ABI, drivers, engine 1) and inst2.hip (the per-dimension
// instantiations of the engine 2 / engine 3 kernels, compiled in parallel - one object per n_para).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smcmi.h"
#include "devstate.hpp"
#include "kernels.hpp"
#include "stage2.hpp"
#include "stage2b.hpp"
#include "stage3.hpp"

using namespace smcmi;
struct CallbackBuffers;            // pinned staging buffers of the host-likelihood path (callback.hpp)

struct Eng2 {
    Geo2 g{};
    Ctl2 *d_ctl = nullptr;
    double *rows_mut = nullptr, *rows_cm = nullptr, *csum = nullptr, *csum_full = nullptr, *rows_gm = nullptr, *rows_pass[2] = {nullptr, nullptr};
    double *vt_mut = nullptr, *vt_cm = nullptr, *vt_gm = nullptr, *vt_pass = nullptr;
    long long *d_ranges = nullptr, *d_ranges_all = nullptr;     // a resample stage: the rows every handle needs of this one; all handles' tables
    Prop2Glob *d_pre = nullptr;      // decision + proposal of the current stage (k2_prepare; large clouds / several handles)
    int *d_tick = nullptr;           // ticket counters of the fused row totals (Tail2): [0, V) correction rows, [V, 2V) mutation rows
    long long *d_prof = nullptr;     // development only (SMCMI_PROF2=<stage>): [0,64) K1 stamps, [64,128) K2 stamps of that stage
    int prof_stage = 0;
    int world = 0;
    // engine 3 (stage3.hpp): tickets, records, time-out flag words, per-launch stage counts (profiling)
    int *d_tick3 = nullptr;
    unsigned long long *d_rec3 = nullptr, *d_to3 = nullptr, *d_gran3 = nullptr;    // (d_gran3: rows and shard totals as granules)
    int *d_done3 = nullptr;
    smcmi::Sel3Args *d_sel3 = nullptr;    // what a segment's in-place selection needs (stage3.hpp)
    double *d_transit3 = nullptr;         // ... where its workers park their particles when the kernel's LDS has no room (Sel3Args::transit)
    // host copies of small per-run uploads (members, not locals: the asynchronous copy may read them after the call that issued it has returned)
    smcmi::Sel3Args h_sel3{};
    unsigned long long h_to3[2] = {0ull, 0ull};
    unsigned seg_seq = 0;
    // a segment's exit note in host-mapped memory (one handle): [0] = the launch sequence number of the segment that has left, behind it a copy of
    // Ctl2 - the host learns of a batch's end (or of a stage that must resample) without a device-to-host copy and a stream sync
    void *h_note3 = nullptr, *d_note3 = nullptr;
    int seg_ch = 1;                  // 512-particle chunks per segment worker (2: one handle of 126 977 .. 253 952 particles, α = 1, a cheap likelihood)
    int e3_state = 0;                // 0 untested, 1 usable (residency self-test passed), -1 off for this handle
    int seg_attr_set = 0;         /* bits 0 / 1: α = 1 / mixture variant, bits 2 / 3: their riding instantiations, bits 4 / 5: two chunks per worker */       // k3_segment's dynamic-LDS opt-in done on this handle's device
    bool wide_attr_set = false;      // k2w_mutate's dynamic-LDS opt-in done on this handle's device
    bool rng_ahead = false;          // K1 carries blocks that draw the mutation's random numbers into the handle's zbuf
    int n_steps = 1, n_blocks = 1;
    int z_ahead = 0;                 // large shards: proposals per particle the drawing blocks of K1 leave in zbuf (0 with rng_ahead: all of them)
};

static const int ESUM_RED_ROWS = 128;         // rows left by the first level of the energy-sum reduction when there are very many blocks
struct smcmi_handle {
    smcmi_config cfg{};
    int d = 0, R = 0, npairs = 0;
    long long n = 0;                 // local particles
    hipStream_t stream = nullptr;
    CloudPtrs cl{};
    DevState *d_st = nullptr;
    DevState h_st{};
    ModelDev *d_model = nullptr;
    ModelDev h_model{};
    bool have_params = false, have_lik = false;
    double *d_data[2] = {nullptr, nullptr}, *d_aux[2] = {nullptr, nullptr};
    Records rec{};
    double *d_sched = nullptr;
    int sched_len = 0;
    // scratch
    int nb_e = 0, nb_m = 0, nb_mr = 0, nb_mut = 0, nb_mut_ls4 = 0, mut_T = 0, nb_reg = 0, reg_T = 0;
    size_t mut_lds = 0, mom_lds = 0, reg_lds_base = 0, prep_lds = 0;
    double *d_prep_rows = nullptr;    // PREP_G group rows of the prepare launch's two-level totals, the ticket behind them (d_prep_tick)
    int *d_prep_tick = nullptr;
    double *d_part_ess[2] = {nullptr, nullptr}, *d_part_fin = nullptr, *d_part_cm = nullptr, *d_wt = nullptr, *d_chunk_off = nullptr, *d_cum = nullptr;
    long long *d_anc = nullptr;
    double *d_part_mom = nullptr, *d_totals = nullptr, *d_acc_part = nullptr, *d_esum_part = nullptr, *d_esum_red = nullptr, *d_emax_part = nullptr, *d_zbuf = nullptr, *d_comm = nullptr, *d_offsets = nullptr;
    long long comm_cap = 0;
    double *d_hist_w = nullptr, *d_hist_W = nullptr;
    std::vector<double> lik_host_data[2], lik_host_aux[2];   // host copies (lgss_kalman only): is the old vintage a prefix of the new one?
    // peer mailbox of sharded engine-2 runs (stage2.hpp Mailbox): this handle's table, the peers' tables as mapped here
    unsigned long long *d_mbox = nullptr;
    unsigned long long **d_peers = nullptr;
    std::vector<unsigned long long *> h_peers;
    std::vector<void *> ipc_opened;           // peers' tables opened through hipIpcOpenMemHandle (closed with the handle)
    bool mbox_ok = false;                     // RCCL driver: every rank mapped every table and the self-test passed everywhere
    bool mbox_tried = false;
    bool mbox_used = false;                   // the last engine-2 run of this handle handed its sums over through the mailbox
    unsigned mbox_epoch = 0;
    double *d_snap = nullptr;         // single-handle runs that may use engine 3: the cloud the run started from and, behind it, its DevState (repeat after a segment time-out)
    int seg_timeouts = 0;             // runs repeated as launches after a segment time-out
    int h_lag0 = 0;                   // source of an asynchronous copy (run2.hpp: RunParams::shift_lag switched off in mid-run)
    double *d_mix = nullptr;          // register mutation kernel, α < 1: dense mixture matrices per block (k_mix_prepare)
    int *d_mixpos = nullptr;
    // host-callback split
    double *d_prop = nullptr, *d_prop_lp = nullptr, *d_prop_q = nullptr, *d_lik_new = nullptr, *d_lik_old = nullptr;
    int *d_acc_count = nullptr, *d_flag = nullptr;
    double *d_cum_full = nullptr, *d_part_full = nullptr, *d_off_full = nullptr;
    int nb_full = 0;
    // sharded driver (sharded.hpp)
    void *nccl = nullptr;
    int rank = 0, world = 1;
    smcmi_host_comm hostc{};         // host-mediated communicator (smcmi_comm_init_host) and its staging buffers
    bool has_hostc = false;
    std::vector<double> hc_send, hc_recv;
    double *d_tot_ess = nullptr, *d_tot_fin = nullptr, *d_tot_mom = nullptr, *d_tot_acc = nullptr, *d_full_w = nullptr, *d_full_cloud = nullptr;
    int last_n_stages = 1;
    int launch_nb = 1;
    size_t zbuf_cap = 0;         // doubles allocated in d_zbuf (random numbers drawn ahead of the mutation, kernels.hpp RngAhead)
    bool spec_stage = false;     // the enqueued stage takes the predicted ϕ_n without a certificate pass (W̃ goes to d_wt)
    bool fused_cm = false;       // the enqueued stage ran k_correct_moments: the mutation kernel normalises the weights
    bool rng_ahead = false;      // the enqueued stage's k_prepare_mutation fills d_zbuf and the mutation kernel reads it
    int z_ahead = 0;             //   ... for the first z_ahead proposals (MH step x block) of every particle
    bool run_adaptive = false;   // the enqueued stage belongs to an adaptive-schedule run (mutation leaves energy sums)
    int noop_grid = 4096;          // grid cap of the selection kernels inside smcmi_run (they are no-ops on most stages)
    bool launch_alpha1 = false;
    long long *d_prof = nullptr;   // development only (SMCMI_PROF2: engine 1's phase stamps)
    // host-mapped progress words of a fixed-schedule run enqueued without selection kernels (smcmi_run): [0] stage index the device
    // has begun, [1] != 0: a stage stalled because it resamples after all.  h_note is the host view, d_note the device alias.
    volatile int *h_note = nullptr;
    int *d_note = nullptr;
    bool note_on = false;          // the stage being enqueued posts to the words
    int graph_sig = 0;
    Eng2 *e2 = nullptr;            // engine 2 (stage2.hpp / run2.hpp): rows, virtual-shard totals, Ctl2
    // host likelihoods (callback.hpp)
    smcmi_lik_callback cb[2] = {nullptr, nullptr};
    void *cb_ud[2] = {nullptr, nullptr};
    CallbackBuffers *cbuf = nullptr;
    long long cb_calls = 0, cb_evals = 0;
    bool cb_energy = false;        // the run's accept launches leave energy power sums / maxima (adaptive single-handle closure runs: predictor rings, shifted weights)
};
