// run1.hpp - engine 1's driver: the stage as a fixed kernel sequence (enqueue_stage) and smcmi_run, the entry point every one-handle run comes
// through (it hands runs that qualify to engines 2 / 3: run2.hpp).  Included by smcmi.hip behind the stage primitives it enqueues.
#pragma once

// ------------------------------------------------------------------------------------------------ whole loop
// One stage = a fixed kernel sequence (no host decision inside): see the header of kernels.hpp.
// p0 > 0 resumes a stage whose solver ran out of passes after p0 of them (st->done == 2 stall, see solver_prologue): the
// search continues with passes p0 .. p0 + solver_passes - 1 exactly as if the original list had been that much longer.
// no_select: the host expects no resampling in this stage: k_post_correct and k_resample_gather are not launched, the moments
// kernel takes the decision itself and stalls the run (done = 3) if selection is needed after all.
// tail_only: resume such a stage from k_post_correct on (the correction is already done).
static void enqueue_stage(smcmi_handle *h, bool adaptive, int solver_passes, int method, int n_blocks, double alpha, int acc_nb,
                          hipEvent_t ev0, hipEvent_t ev1, int p0 = 0, bool no_select = false, bool tail_only = false, bool spec = false,
                          bool skip_begin = false, bool host_mut = false) {
    const long long n = h->n;
    hipStream_t s = h->stream;
    // spec: predict -> correct -> verify (kernels.hpp k_stage_begin): no certificate pass is enqueued at all - 4 launches
    const int P = (adaptive && !spec) ? p0 + solver_passes : 0;
    static const int no_pred = getenv("SMCMI_NO_PREDICTOR") ? atoi(getenv("SMCMI_NO_PREDICTOR")) : 0;   // development only
    h->run_adaptive = adaptive;
    const int fin_slot = P == 0 ? 0 : (P & 1);
    // no selection expected and the register kernels apply: the correction pass gathers the moments too, k_prepare_mutation
    // takes the post-correction decision, the mutation kernel normalises the weights (5 launches per stage)
    const bool cm = no_select && !tail_only && can_fuse_cm(h);
    h->fused_cm = cm;
    h->spec_stage = spec && cm;
    if (!tail_only) {
    if (p0 == 0 && !skip_begin) {
        // (a host-callback mutation without cb_energy - sharded closure runs, fixed schedules - leaves neither energy sums nor energy maxima:
        // plain schedule walk, unshifted weights)
        const bool hm_plain = host_mut && !h->cb_energy;
        const double *es = (adaptive && !no_pred && !hm_plain) ? h->d_esum_part : nullptr;
        int es_nb = acc_nb, em_nb = acc_nb;
        const double *em = hm_plain ? nullptr : h->d_emax_part;
        // tens of thousands of rows are not for one block: blocks 1..ESUM_RED_ROWS of the same launch total a chunk each (k_stage_begin)
        PrepRed rr{};
        unsigned grid = 1;
        // (reducer blocks: ~96 rows each - 40 blocks at N = 1e6, the full 128 from 3e6 on; block 0 adds one group row per reducer)
        if (es && acc_nb > 2048 && em) { rr.rows = h->d_esum_red; rr.tick = h->d_prep_tick + 1; grid = 1u + (unsigned)std::min(ESUM_RED_ROWS, std::max(16, acc_nb / 96)); }
        else if (es && acc_nb > 2048) {
            k_reduce_rows<<<ESUM_RED_ROWS, TB, 0, s>>>(h->d_esum_part, acc_nb, ES, h->d_esum_red, nullptr, nullptr);
            es = h->d_esum_red; es_nb = ESUM_RED_ROWS;
        }
        k_stage_begin<<<grid, BT, 0, s>>>(h->d_st, h->d_sched, h->d_acc_part, es ? es_nb : acc_nb, h->rec, es, h->d_prof ? h->d_prof + 9 : nullptr,
                                          h->spec_stage ? 1 : 0, em, em_nb, rr, h->note_on ? h->d_note : nullptr);
    }
    if (adaptive && !h->spec_stage) enqueue_solver(h, P, p0);
    if (cm) launch_correct_moments(h, P);
    else k_pass<1, true><<<h->nb_e, TB, 0, s>>>(h->cl, h->d_st, h->d_sched, h->d_part_ess[(P + 1) & 1], h->d_part_fin, h->nb_e, P, h->d_hist_w, n);
    }
    int nbm = 0;
    if (cm) {
        launch_prepare_in_run(h, h->d_part_cm, h->nb_e, 3, fin_slot);
        if (ev0) hipEventRecord(ev0, s);
        launch_mutate(h, n_blocks, 0, alpha);
        if (ev1) hipEventRecord(ev1, s);
        return;
    }
    if (no_select && !tail_only && can_fuse_post(h)) nbm = launch_moments(h, h->d_hist_W, 0, fin_slot);
    else {
        k_post_correct<<<h->nb_e, TB, 0, s>>>(h->d_st, h->d_part_fin, h->nb_e, nullptr, h->rec, fin_slot, h->cl, h->d_cum);
        k_resample_gather<<<(unsigned)std::min<long long>((n + TB - 1) / TB, 4 * h->noop_grid), TB, 0, s>>>(h->cl, h->d_st, h->d_cum, n, 0, h->cfg.n_parts, method,
                                                                     h->cfg.seed, 0u, nullptr, h->d_anc, nullptr, 0);
        nbm = launch_moments(h, h->d_hist_W, 0);
    }
    launch_prepare_in_run(h, h->d_part_mom, nbm, 1);
    if (host_mut) return;                   // the mutation runs through the host callback (callback.hpp)
    if (ev0) hipEventRecord(ev0, s);
    launch_mutate(h, n_blocks, 0, alpha);
    if (ev1) hipEventRecord(ev1, s);
}

// a (host closure, device family) pair in a tempered update: the callback path would score the old likelihood as 0, the device path
// would call a device family that does not exist - refuse instead of sampling the wrong posterior
static int check_lik_pair(const smcmi_handle *h) {
    const int f1 = h->h_model.lik[1].family;
    const bool old_dev = f1 != SMCMI_LIK_NONE && f1 != SMCMI_LIK_HOST_CALLBACK, old_cb = h->cb[1] != nullptr;
    if ((h->cb[0] && old_dev) || (!h->cb[0] && old_cb))
        return set_err(SMCMI_ERR_UNSUPPORTED, "the new and the old likelihood must both be device families or both host callbacks");
    return 0;
}
struct ShardGroup;
static int run_callback(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res);
static bool eng2_eligible(const smcmi_handle *h, int world, bool single, const smcmi_run_config *rc);
static int run2_single(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res);

extern "C" int smcmi_run(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res) {
    if (int e = need_model(h, 2)) return e;
    if (!rc || !res) return set_err(SMCMI_ERR_ARG, "null argument");
    if (h->cfg.n_local != h->cfg.n_parts) return set_err(SMCMI_ERR_UNSUPPORTED, "smcmi_run drives a single shard; use the shard-level calls for multi-GPU");
    res->n_segments = 0; res->segment_stages = 0; res->kernel_ms_segments = 0.0;
    res->segment_blocks = 0; res->segment_state = 0; res->segment_timeouts = 0; res->shift_fallback_stage = 0;
    if (int e = check_lik_pair(h)) return e;
    if (h->cb[0]) return run_callback(h, rc, res);                    // user likelihood on the host (callback.hpp)
    if (eng2_eligible(h, 1, true, rc)) return run2_single(h, rc, res);          // n_para <= 10: the two-launch stage (stage2.hpp)
    const int nf = h->h_model.n_free;
    if (rc->n_blocks < 1 || rc->n_blocks > nf || ((nf + rc->n_blocks - 1) / rc->n_blocks) * (rc->n_blocks - 1) >= nf)
        return set_err(SMCMI_ERR_ARG, "n_blocks incompatible with the number of free parameters");
    if (rc->n_phi < 2 || rc->n_mh_steps < 1) return set_err(SMCMI_ERR_ARG, "bad n_phi / n_mh_steps");
    if (rc->resampling_method != SMCMI_RESAMPLE_SYSTEMATIC && rc->resampling_method != SMCMI_RESAMPLE_MULTINOMIAL)
        return set_err(SMCMI_ERR_ARG, "Invalid resampler in SMC. Options are systematic or multinomial");
    const bool adaptive = !rc->use_fixed_schedule;
    if (!adaptive && rc->n_phi > h->cfg.max_stages) return set_err(SMCMI_ERR_CAPACITY, "max_stages < n_phi");
    if (pull_state(h)) return SMCMI_ERR_HIP;
    // proposed fixed schedule ((k-1)/(n_Φ-1))^λ, smc_main.jl:348-352
    std::vector<double> sched(rc->n_phi);
    for (int k = 0; k < rc->n_phi; ++k) sched[k] = pow((double)k / (double)(rc->n_phi - 1), rc->lambda);
    if (upload_sched(h, sched.data(), rc->n_phi)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    const int cur = s.cur;
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
    // continue_run (continue_intermediate, smc_main.jl:334-335,355-361): keep the loop scalars, records and history the handle
    // holds (left by a paused run, or put there by smcmi_set_loop_state / _set_stage_records / _set_history)
    const bool cont = rc->continue_run != 0;
    if (cont) {
        if (s.stage < 1 || s.stage >= h->cfg.max_stages) return set_err(SMCMI_ERR_STATE, "no loop state to continue from");
        if (s.phi_n >= 1.0) return set_err(SMCMI_ERR_STATE, "the run to continue has already reached phi = 1");
        s.rp = rp; s.done = 0; s.err = 0; s.skip_fold = 1; s.do_resample = 0;
    } else {
        memset(&s, 0, sizeof(DevState));
        s.e_seen = __builtin_nan("");
        s.rp = rp; s.cur = cur;
        s.stage = 1; s.j = 2;                                   // i = 1, j = 2 (smc_main.jl:198-199)
        s.c = rc->c; s.accept = rc->target;                     // initialize_cloud_settings!, initialization.jl:196-211
        s.ess_prev = rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts;   // tempered update: ESS of the old cloud (initialization.jl:199-200)
    }
    const int base = cont ? s.stage - 1 : 0;                // stages completed before this call
    if (push_state(h)) return SMCMI_ERR_HIP;
    // the arrival counters of the two-level totals (PrepRed) start every run at zero: a launch whose wait timed out (SMCMI_ERR_TIMEOUT,
    // the run is void) may have left late arrivals behind - in stream order they precede this fill
    HIP_TRY(hipMemsetAsync(h->d_prep_tick, 0, 8 * sizeof(double), h->stream));
    // stage-1 records and history columns (w[:,1] = 0, W[:,1] = weights; smc_main.jl:363-366)
    if (!cont) {
        const double v0[4] = {0.0, rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts, rc->c, rc->target};
        HIP_TRY(hipMemcpyAsync(h->rec.phi, &v0[0], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.ess, &v0[1], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.c, &v0[2], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.accept, &v0[3], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(h->rec.resampled, 0, sizeof(int) * h->cfg.max_stages, h->stream));
        if (h->cfg.store_history) {
            HIP_TRY(hipMemsetAsync(h->d_hist_w, 0, sizeof(double) * h->n, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_hist_W, h->cl.buf[cur] + (long long)(h->R - 1) * h->n, sizeof(double) * h->n,
                                   hipMemcpyDeviceToDevice, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (getenv("SMCMI_PROF2") && !h->d_prof) { if (dmalloc(&h->d_prof, 32)) return SMCMI_ERR_HIP; }
    if (int e = ensure_zbuf(h, rc->n_mh_steps, rc->n_blocks)) return e;
    const int solver_passes = rc->solver_passes >= 1 ? rc->solver_passes : DEFAULT_SOLVER_PASSES;
    const int first_passes = std::max(solver_passes, FIRST_SOLVER_PASSES);
    const int sync_every = rc->sync_every > 0 ? rc->sync_every : 32;     // (16 until round 4: 36.75 vs 36.45 ms per run at N = 1e6)
    const int acc_nb = mut_blocks(h);
    // largest energy of the initial cloud, in the layout the mutation epilogue uses afterwards (stage 1's energy shift)
    k_energy_max<<<acc_nb, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_emax_part);
    const bool profile = rc->use_graph == 2;    // 2 = direct launches with HIP events around the mutation kernel
    std::vector<hipEvent_t> evs;
    std::vector<int> ev_iter;      // profile mode: iteration each event pair belongs to, -1 once known to have bracketed a no-op
    // Resampling is predictable on an adaptive schedule (every stage ends at ESS = target x the previous ESS, or x N after a
    // resample), so the host enqueues the selection kernels only where it expects a resample; the device checks the expectation
    // (k_moments_reg) and stalls the run if it was wrong.  SMCMI_NO_SELECT_PREDICT=1 (development) keeps the full list everywhere,
    // =2 deliberately predicts "never" to exercise the stall path.
    static const int sel_mode = getenv("SMCMI_NO_SELECT_PREDICT") ? atoi(getenv("SMCMI_NO_SELECT_PREDICT")) : 0;
    // (Fixed schedules: extrapolating the ESS decay was tried and dropped - CAPM-like posteriors collapse within two or three
    // stages, 19 of 20 resamples stalled, and the per-batch sync it needs makes short stages host-bound.)
    const bool predict_select = adaptive && can_fuse_post(h) && sel_mode != 1;
    // Fixed schedules: the host cannot foresee which stages resample, so EVERY stage is enqueued without the selection kernels
    // (begin, correction + moments, prepare, mutation: four launches instead of seven) and a stage that resamples after all stalls
    // (done = 3) and is resumed through the full path, exactly as a mispredicted stage of an adaptive run.  What made this a loss
    // before was the drain - the whole schedule is enqueued at once, a stall left hundreds of idle launches behind it - and a sync
    // per batch makes short stages host-bound.  Instead the host stays `run_ahead` stages in front of the device WITHOUT a sync:
    // k_stage_begin posts its stage index and the stalling k_prepare_mutation a flag into host-mapped words (handle.hpp h_note)
    // which the enqueue loop polls; a drained stream (an error, a pause, phi = 1) also ends the wait.
    static const int fixed_sel = getenv("SMCMI_FIXED_NO_SELECT") ? atoi(getenv("SMCMI_FIXED_NO_SELECT")) : 1;        // development: 0 = the seven-launch stage
    const int run_ahead = 1;      // (config 4: 30.6 ms at 1, 30.8 at 2, 31.1 at 4 - fewer idle launches behind a stall; measured in round 4, the switch retired in round 6)
    bool fixed_ns = !adaptive && can_fuse_cm(h) && sel_mode != 1 && fixed_sel != 0;
    if (fixed_ns && !h->h_note) {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            h->h_note = (volatile int *)hp; h->d_note = (int *)dp;
        } else {
            if (hp) hipHostFree(hp);
            (void)hipGetLastError();
            fixed_ns = false;
        }
    }
    struct NoteGuard { smcmi_handle *h; ~NoteGuard() { h->note_on = false; } } note_guard{h};
    h->note_on = fixed_ns;
    if (fixed_ns) { h->h_note[0] = s.stage; h->h_note[1] = 0; }
    // (with a prior weight the correction's incremental weight differs from the solver's objective - quirk Q4 - so the ESS it
    // produces cannot verify a predicted root: those runs keep the certificate pass)
    const bool spec_ok = predict_select && can_fuse_cm(h) && !getenv("SMCMI_NO_PREDICTOR") &&
                         rc->tempered_update_prior_weight == 0.0 && rp.phi_rtol > 0.0;
    double pred_ess = cont ? s.ess_prev : (rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts);   // ESS after the last completed stage
    int pred_rl = cont ? s.resampled_last : 0;                                             // resampled_last_period
    const auto t0 = std::chrono::steady_clock::now();
    int launched = 0, done = 0;
    res->solver_stalls = 0; res->select_stalls = 0; res->spec_stalls = 0;
    const int max_iter = (adaptive ? h->cfg.max_stages : rc->n_phi - 1) - base;
    int stall_stage = -1, stall_p = 0;        // stage that last ran out of solver passes and how many it has had so far
    int dyn_P = solver_passes;                // passes enqueued per stage: raised when stalls are frequent (poorly predictable models)
    // Predict-correct-verify pays only while predictions verify: three failures, each within four stages of the one before
    // (heavy-tailed energies, steps too long for the 16-term model), switch the rest of the run to the certificate path,
    // where a miss costs an extra pass instead of a host round trip.
    bool spec_on = spec_ok;
    int last_spec_stall = -100, spec_strikes = 0, last_solver_stall = -100;
    if (rc->solver_passes < 1 && rc->tempering_target < 0.95) dyn_P = 2;   // larger steps: the 8-term model is good to ~1e-3 only, two passes are the norm
    int stages_left_est = 1 << 30;             // from the last sync: (1 - ϕ_n) / (ϕ_n - ϕ_{n-1}), an over-estimate while the steps grow
    while (launched < max_iter && !done) {
        // near the end of the run the batch shrinks to what is left, so that few no-op stages trail the one that reaches ϕ = 1
        int batch = adaptive ? std::min(std::min(sync_every, std::max(stages_left_est, 4)), max_iter - launched) : max_iter - launched;
        for (int b = 0; b < batch; ++b) {
            bool no_select = false;
            if (fixed_ns) {
                // iteration `launched` begins stage base + launched + 2: wait until the device has begun the stage run_ahead before it
                const int need = base + launched + 2 - run_ahead;
                bool leave = false;
                while (h->h_note[0] < need) {
                    if (h->h_note[1] != 0) break;
                    if (hipStreamQuery(h->stream) != hipErrorNotReady) { leave = h->h_note[0] < need; break; }   // nothing left in flight: look at the state
                }
                if (h->h_note[1] != 0) {
                    // Stage h_note[0] resamples after all (the begins behind it returned at once and posted nothing).  No sync: clear
                    // the stall behind the idle launches already in the stream, run the rest of that stage through the full path
                    // (tail_only: the correction is done) and go on enqueuing from the stage after it.
                    const int st_i = h->h_note[0];
                    h->h_note[1] = 0;
                    HIP_TRY(hipMemsetAsync(&h->d_st->done, 0, sizeof(int), h->stream));
                    for (int &it : ev_iter)
                        if (it >= st_i - 2 - base) it = -1;           // the stalled stage and everything behind it were no-ops
                    hipEvent_t r0 = nullptr, r1 = nullptr;
                    if (profile) { hipEventCreate(&r0); hipEventCreate(&r1); evs.push_back(r0); evs.push_back(r1); ev_iter.push_back(st_i - 2 - base); }
                    enqueue_stage(h, adaptive, 0, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, r0, r1, 0, false, true);
                    res->select_stalls += 1;
                    launched = st_i - 1 - base;
                    batch = max_iter - launched; b = -1;
                    continue;
                }
                if (leave) break;
                no_select = true;
            }
            if (predict_select) {
                // ESS this stage will end at (helpers.jl:14-20), with a margin: a wrong "resample" guess only costs two idle launches
                const double ess_bar = rc->tempering_target * (pred_rl ? (double)h->cfg.n_parts : pred_ess);
                const bool rs = ess_bar < rp.threshold * (1.0 + 1e-6);
                no_select = !rs || sel_mode == 2;
                pred_ess = ess_bar; pred_rl = rs ? 1 : 0;
            }
            const bool spec = spec_on && no_select && launched >= 2;
            {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (profile) { hipEventCreate(&e0); hipEventCreate(&e1); evs.push_back(e0); evs.push_back(e1); ev_iter.push_back(launched); }
                enqueue_stage(h, adaptive, launched < 2 ? first_passes : dyn_P, rc->resampling_method, rc->n_blocks, rc->alpha,
                              acc_nb, e0, e1, 0, no_select, false, spec);
            }
            ++launched;
        }
        // one copy per sync: the loop scalars from `stage` to `ess_prev` are contiguous in DevState (64 bytes) - the done flag, the
        // last stage's resample decision and its ESS used to be three copies (each a ~2.5 µs copy kernel plus a host round trip)
        DevState head;
        constexpr size_t head_off = offsetof(DevState, stage), head_len = offsetof(DevState, ess) - offsetof(DevState, stage);
        HIP_TRY(hipMemcpyAsync((char *)&head + head_off, (const char *)h->d_st + head_off, head_len, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        done = head.done;
        bool resumed = false;
        while (done == 2 || done == 3 || done == 4) {
            resumed = true;
            if (pull_state(h)) return SMCMI_ERR_HIP;
            if (fixed_ns) h->h_note[1] = 0;               // (the stream is drained: no post is in flight)
            const int st_i = s.stage;
            const int had = (st_i == stall_stage) ? stall_p : (st_i - base <= 3 ? first_passes : dyn_P);
            const int zero = 0;
            HIP_TRY(hipMemcpyAsync(&h->d_st->done, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
            for (int &it : ev_iter)
                if (it >= st_i - 2 - base) it = -1;           // the stalled stage and everything behind it were no-ops
            hipEvent_t r0 = nullptr, r1 = nullptr;
            if (profile) { hipEventCreate(&r0); hipEventCreate(&r1); evs.push_back(r0); evs.push_back(r1); ev_iter.push_back(st_i - 2 - base); }
            if (done == 4) {
                // A stage enqueued without a certificate pass had no usable prediction, or the ESS its correction produced did
                // not verify it: nothing of the stage has been committed (W̃ went to scratch).  Re-arm the solver with the plain
                // schedule walk and run the stage through the certificate-pass path.
                k_solver_rearm<<<1, 64, 0, h->stream>>>(h->d_st, h->d_sched);
                enqueue_stage(h, adaptive, first_passes, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, r0, r1, 0, false, false, false, true);
                stall_stage = st_i; stall_p = first_passes;
                res->spec_stalls += 1;
                if (st_i - last_spec_stall <= 4) { if (++spec_strikes >= 2) spec_on = false; }
                else spec_strikes = 0;
                last_spec_stall = st_i;
            } else if (done == 2) {
                // A stage exhausted its solver passes: it and everything enqueued behind it did nothing.  Clear the stall, give
                // that stage more passes (continuing the same search), and go on from the stage after it.
                const int more = 8;
                if (had > 1200) return set_err(SMCMI_ERR_BRACKET, "adaptive tempering solver: the search for phi_n does not terminate (the ESS objective is not a number?)");
                enqueue_stage(h, adaptive, more, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, r0, r1, had);
                stall_stage = st_i; stall_p = had + more;
                res->solver_stalls += 1;
                // a stall flushes the rest of its batch and costs a host round trip, an idle pass launch costs 3 µs: two stalls
                // within four stages -> enqueue one more pass per stage from here on
                if (st_i - last_solver_stall <= 4 && dyn_P < 4) ++dyn_P;
                last_solver_stall = st_i;
            } else {
                // A stage enqueued without selection kernels needs to resample after all: nothing past its correction has run.
                // Run the rest of that stage with the full path, then go on from the stage after it.
                enqueue_stage(h, adaptive, 0, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, r0, r1, had, false, true);
                res->select_stalls += 1;                   // selection stalls (diagnostic)
            }
            launched = st_i - 1 - base;
            HIP_TRY(hipMemcpyAsync(&done, &h->d_st->done, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        if (resumed) {               // a resumed stage ran after the copy above
            HIP_TRY(hipMemcpyAsync((char *)&head + head_off, (const char *)h->d_st + head_off, head_len, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        // The stage that reaches ϕ = 1 raises `done` only through the NEXT stage's k_stage_begin.  When it was the last one of its
        // batch (config 2: 256 stages = 16 batches of 16) nothing has raised it yet - do not enqueue a whole batch of no-ops (64
        // launches and a sync) to find out: the closing k_stage_begin below does the same bookkeeping.
        if (!done && head.phi_n >= 1.0) break;
        if (head.phi_n > head.phi_prev && head.phi_n < 1.0) {
            const double left = (1.0 - head.phi_n) / (head.phi_n - head.phi_prev);
            stages_left_est = left < 1e6 ? (int)left + 1 : 1 << 30;
        }
        if (predict_select) {
            // re-anchor the expectation on the device's ESS / flag after every sync
            s.resampled_last = head.do_resample;       // did the last stage resample
            s.ess_prev = head.ess_prev;
            pred_ess = s.ess_prev;
            pred_rl = s.resampled_last;
        }
        static const int trace = getenv("SMCMI_TRACE") ? atoi(getenv("SMCMI_TRACE")) : 0;   // development only
        if (trace) {
            static long long last_passes = 0;
            if (pull_state(h)) return SMCMI_ERR_HIP;
            fprintf(stderr, "[smcmi] stage %d phi %.12e dphi %.6e pred %.6e relerr %.2e passes %lld ess %.1f rs %d\n", s.stage, s.phi_n,
                    s.phi_n - s.phi_prev, s.pred_delta, (s.pred_delta - (s.phi_n - s.phi_prev)) / (s.phi_n - s.phi_prev),
                    s.solver_passes - last_passes, s.ess, s.do_resample);
            last_passes = s.solver_passes;
            if (h->d_prof) {
                long long pr[16];
                hipMemcpy(pr, h->d_prof, sizeof(pr), hipMemcpyDeviceToHost);
                long long pq[32];
                hipMemcpy(pq, h->d_prof, sizeof(pq), hipMemcpyDeviceToHost);
                fprintf(stderr, "[smcmi]    prepare phase ticks: %lld %lld %lld %lld %lld %lld\n", pq[26] - pq[25], pq[27] - pq[26], pq[28] - pq[27], pq[29] - pq[28], pq[30] - pq[29], pq[31] - pq[30]);
                fprintf(stderr, "[smcmi]    mutate phase ticks (block 0):");
                for (int q = 1; q <= 8; ++q) fprintf(stderr, " %lld", pq[q] - pq[q - 1]);
                fprintf(stderr, "  total %lld\n", pq[8] - pq[0]);
                fprintf(stderr, "[smcmi]    begin phase ticks: %lld %lld %lld %lld %lld\n", pr[10] - pr[9], pr[11] - pr[10], pr[12] - pr[11], pr[14] - pr[12], 0ll);
            }
            for (int q = 0; q < 2; ++q)
                fprintf(stderr, "[smcmi]    sol[%d] mode %d nv %d lo-phi %.3e hi-phi %.3e glo %.3e ghi %.3e\n", q, s.sol[q].mode, s.sol[q].n_valid,
                        s.sol[q].lo - s.phi_n, s.sol[q].hi - s.phi_n, s.sol[q].glo, s.sol[q].ghi);
        }
    }
    // fold the last mutation's acceptance rate and close the run
    k_stage_begin<<<1, BT, 0, h->stream>>>(h->d_st, h->d_sched, h->d_acc_part, acc_nb, h->rec);
    if (pull_state(h)) return SMCMI_ERR_HIP;
    const auto t1 = std::chrono::steady_clock::now();
    res->kernel_ms_mutate = 0.0; res->n_mutate_launches = 0;
    // An event pair brackets [previous kernel done -> this kernel done]: dispatch of the kernel included.  Calibrate that
    // fixed part with pairs around an empty kernel and subtract it, so the figure is the kernel's own duration (what
    // rocprofv3 --kernel-trace reports).
    double ev_overhead_ms = 0.0;
    if (profile && !evs.empty()) {
        hipEvent_t c0, c1;
        hipEventCreate(&c0); hipEventCreate(&c1);
        const int reps = 64;
        double acc_ms = 0.0;
        int got = 0;
        for (int r = 0; r < reps; ++r) {
            k_fill<<<(unsigned)((h->n + 255) / 256), 256, 0, h->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)     // predecessor of comparable size
            hipEventRecord(c0, h->stream);
            k_fill<<<(unsigned)((h->n + 255) / 256), 256, 0, h->stream>>>(nullptr, 0, 0.0);      // (an empty launch of the mutation kernel's grid: event-overhead calibration)
            hipEventRecord(c1, h->stream);
            hipStreamSynchronize(h->stream);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c0, c1) == hipSuccess) { acc_ms += ms; ++got; }
        }
        hipEventDestroy(c0); hipEventDestroy(c1);
        // an empty kernel of this grid itself lasts ~2.5 µs in a rocprofv3 kernel trace (wave launch + drain): leave that in
        if (got) ev_overhead_ms = std::max(0.0, acc_ms / got - 0.0025);
    }
    for (size_t k = 0; k + 1 < evs.size(); k += 2) {
        float ms = 0.f;
        if (ev_iter[k / 2] >= 0 && ev_iter[k / 2] < s.stage - 1 - base && hipEventElapsedTime(&ms, evs[k], evs[k + 1]) == hipSuccess) { res->kernel_ms_mutate += std::max(0.0, (double)ms - ev_overhead_ms); res->n_mutate_launches += 1; }
    }
    for (hipEvent_t e : evs) hipEventDestroy(e);
    res->n_stages = s.stage; res->resamples = s.resamples; res->logmdd = s.logz; res->c = s.c; res->accept = s.accept;
    res->seconds = std::chrono::duration<double>(t1 - t0).count();
    res->solver_passes = s.solver_passes;
    res->paused = (s.done == 5) ? 1 : 0;
    h->last_n_stages = s.stage;
    if (s.err == SMCMI_ERR_NAN_ESS) return nan_ess_error(h, h->spec_stage ? h->d_wt : h->cl.buf[0] + (long long)(h->R - 1) * h->n);
    if (s.err) return err_from_state(s.err);
    if (!s.done) return set_err(SMCMI_ERR_CAPACITY, "max_stages exceeded before the tempering schedule reached 1");
    return 0;
}

