// callback.hpp - smcmi_run for a USER likelihood on the host (included by smcmi.hip).
//
// The reference's entry point takes a closure: smc(loglikelihood::Function, parameters, data; ...) (src/smc_main.jl:118) evaluated
// per proposal inside mutation() (src/mutation.jl:93-121).  With smcmi_set_likelihood_callback the device loop keeps everything
// but that evaluation: ϕ solver, correction, selection, moments, proposal and the Metropolis-Hastings decision stay HIP kernels;
// per MH step x block the n x d proposals come to the host once (k_mutate<1>), the callback is invoked ONCE on the batch of
// proposals that passed the bounds check (the reference never calls the likelihood on a vector update! rejected, mutation.jl:93),
// synchronously on the thread that called smcmi_run (Julia @cfunction safety), and the log-likelihoods go back for the decision
// (k_mutate<2>).  Contract of the callback (include/smcmi.h): out[k] = log-likelihood of theta[k + n * j], j < d; -Inf allowed
// (mutation.jl:102-104); a non-zero return aborts the run with SMCMI_ERR_CALLBACK.
#pragma once

// Phases of the host-likelihood mutation, accumulated over a run (smcmi_callback_phases; wall clock of the calling thread, ms)
enum { CBP_FIRST = 0,      // propose kernel + the first chunk's way across PCIe (nothing to overlap it with)
       CBP_WAIT = 1,       // waiting for later chunks (0 when the callback is the slower side)
       CBP_PACK = 2,       // gathering the in-bounds proposals of a chunk that has out-of-bounds ones (0 when every proposal passed)
       CBP_CALL = 3,       // inside the user's callback
       CBP_SCATTER = 4,    // NaN -> -Inf pass / scatter of the packed results
       CBP_ENQUEUE = 5,    // enqueueing copies and kernels
       CBP_STAGE = 6,      // the stage's device part up to the proposal set-up, incl. the per-stage sync (run_callback)
       CBP_N = 8 };

struct CallbackBuffers {
    double *h_prop = nullptr, *h_lp = nullptr, *h_pack = nullptr, *h_lik[2] = {nullptr, nullptr}, *h_out = nullptr;
    long long *h_idx = nullptr;
    long long n = 0;
    int d = 0;
    // the chunk pipeline of host_mutation: proposals cross PCIe chunk by chunk on their own stream while the callback scores the chunk before
    hipStream_t s_down = nullptr, s_up = nullptr;
    hipEvent_t ev_prop = nullptr, ev_up = nullptr, ev_head = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    double phase_ms[CBP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static void free_callback_buffers(CallbackBuffers *b) {
    if (!b) return;
    void *ptrs[] = {b->h_prop, b->h_lp, b->h_pack, b->h_lik[0], b->h_lik[1], b->h_out, b->h_idx};
    for (void *p : ptrs)
        if (p) hipHostFree(p);
    for (hipEvent_t e : b->ev_chunk) hipEventDestroy(e);
    if (b->ev_prop) hipEventDestroy(b->ev_prop);
    if (b->ev_up) hipEventDestroy(b->ev_up);
    if (b->ev_head) hipEventDestroy(b->ev_head);
    if (b->s_down) hipStreamDestroy(b->s_down);
    if (b->s_up) hipStreamDestroy(b->s_up);
    delete b;
}
constexpr int CB_MAX_CHUNKS = 16;
static int ensure_callback_buffers(smcmi_handle *h) {
    if (h->cbuf && h->cbuf->n == h->n && h->cbuf->d == h->d) return 0;
    if (h->cbuf) { free_callback_buffers(h->cbuf); h->cbuf = nullptr; }
    CallbackBuffers *b = new CallbackBuffers();
    b->n = h->n; b->d = h->d;
    const size_t n = (size_t)h->n, d = (size_t)h->d;
    bool ok = hipHostMalloc((void **)&b->h_prop, n * (d + 1) * 8) == hipSuccess && hipHostMalloc((void **)&b->h_lp, n * 8) == hipSuccess &&
              hipHostMalloc((void **)&b->h_pack, n * d * 8) == hipSuccess && hipHostMalloc((void **)&b->h_lik[0], n * 8) == hipSuccess &&
              hipHostMalloc((void **)&b->h_lik[1], n * 8) == hipSuccess && hipHostMalloc((void **)&b->h_out, n * 8) == hipSuccess &&
              hipHostMalloc((void **)&b->h_idx, n * 8) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&b->s_down, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&b->s_up, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&b->ev_prop, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&b->ev_head, hipEventDisableTiming) == hipSuccess;
    for (int c = 0; ok && c < CB_MAX_CHUNKS; ++c) {
        hipEvent_t e = nullptr;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) b->ev_chunk.push_back(e);
    }
    if (!ok) {
        free_callback_buffers(b);
        return set_err(SMCMI_ERR_HIP, "hipHostMalloc / stream creation failed (callback staging buffers)");
    }
    h->cbuf = b;
    return 0;
}
// chunks a batch of n proposals crosses PCIe in: K = min(8, n / 12288) chunks of ceil(n / K) particles, the last one possibly shorter (a chunk
// costs an event wait and an invocation of the user's function: ~20 µs); SMCMI_CB_CHUNKS=<k> (1 = the whole batch at once: the documented
// opt-out for callbacks with per-batch state, include/smcmi.h, and the phase profile's serial reference)
static int callback_chunks(long long n) {
    static const int forced = getenv("SMCMI_CB_CHUNKS") ? atoi(getenv("SMCMI_CB_CHUNKS")) : 0;
    if (forced > 0) return std::min(forced, CB_MAX_CHUNKS);
    return (int)std::max<long long>(1, std::min<long long>(8, n / 12288));
}
static inline double cb_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Evaluate callback `which` on the m_rows x d column-major block `theta` (leading dimension m_rows) for the rows whose `gate` is finite
// (gate = the proposals' log-priors: -Inf = the bounds check failed, the reference never calls the likelihood there, mutation.jl:93);
// the others get -Inf.  When every row passes - the common case - the block goes to the user's function as it is and the results land in
// lik_out directly; otherwise the passing rows are packed first.  Synchronously on the calling thread.
static int eval_callback(smcmi_handle *h, int which, const double *theta, const double *gate, double *lik_out, long long m_rows = -1) {
    CallbackBuffers *b = h->cbuf;
    const long long n = m_rows >= 0 ? m_rows : h->n;
    const int d = h->d;
    double t0 = cb_now_ms();
    long long m = 0;
    if (gate) { for (long long i = 0; i < n; ++i) m += gate[i] != -HUGE_VAL; } else m = n;
    if (m == n) {
        const int rc = n > 0 ? h->cb[which](theta, (int64_t)n, (int64_t)d, lik_out, h->cb_ud[which]) : 0;
        const double t1 = cb_now_ms();
        b->phase_ms[CBP_CALL] += t1 - t0;
        if (rc != 0) return set_err(SMCMI_ERR_CALLBACK, "the likelihood callback returned " + std::to_string(rc));
        for (long long i = 0; i < n; ++i) { const double v = lik_out[i]; if (v != v) lik_out[i] = -HUGE_VAL; }      // NaN: the reference's `try ... catch` turns a failed evaluation into -Inf
        b->phase_ms[CBP_SCATTER] += cb_now_ms() - t1;
        h->cb_calls += 1; h->cb_evals += n;
        return 0;
    }
    m = 0;
    for (long long i = 0; i < n; ++i)
        if (gate[i] != -HUGE_VAL) b->h_idx[m++] = i;
    for (int j = 0; j < d; ++j) {
        const double *col = theta + (long long)j * n;
        double *dst = b->h_pack + (long long)j * m;
        for (long long k = 0; k < m; ++k) dst[k] = col[b->h_idx[k]];
    }
    double t1 = cb_now_ms();
    b->phase_ms[CBP_PACK] += t1 - t0;
    if (m > 0) {
        const int rc = h->cb[which](b->h_pack, (int64_t)m, (int64_t)d, b->h_out, h->cb_ud[which]);
        if (rc != 0) return set_err(SMCMI_ERR_CALLBACK, "the likelihood callback returned " + std::to_string(rc));
    }
    double t2 = cb_now_ms();
    b->phase_ms[CBP_CALL] += t2 - t1;
    for (long long i = 0; i < n; ++i) lik_out[i] = -HUGE_VAL;
    for (long long k = 0; k < m; ++k) {
        const double v = b->h_out[k];
        lik_out[b->h_idx[k]] = (v != v) ? -HUGE_VAL : v;
    }
    b->phase_ms[CBP_SCATTER] += cb_now_ms() - t2;
    h->cb_calls += 1; h->cb_evals += m;
    return 0;
}

// All MH steps x blocks of one stage's mutation with the host callback (src/mutation.jl:56-138); the proposal was set up by
// k_prepare_mutation of this stage.  Per step x block: propose (k_mutate<1>, proposals chunk-major: a chunk's block = its d proposal
// columns + their log-priors) -> the chunks cross PCIe back to back on a copy stream, each ONE linear copy -> the callback scores chunk c on
// the calling thread while chunk c + 1 is on its way -> the chunk's log-likelihoods go back on a third stream -> accept (k_mutate<2>)
// behind the last of them.  One invocation of the user's function per chunk (two with an old-data callback); the values, and so every
// bit of the run, do not depend on the chunking.
// host_propose_enqueue: the first half for (step, blk) - run_callback enqueues it for the first proposal BEFORE it has read the stage's
// verdict (a stage that did not go ahead leaves the propose kernel a no-op and the copies meaningless: they are dropped), so the
// stage's only host wait is the first chunk's arrival.
static int host_propose_enqueue(smcmi_handle *h, int step, int blk) {
    CallbackBuffers *b = h->cbuf;
    const long long n = h->n;
    const int d = h->d;
    const int K = callback_chunks(n);
    const long long mc = (n + K - 1) / K;
    const double t0 = cb_now_ms();
    MutArgs ma{};
    ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.proposals = h->d_prop; ma.prop_logprior = h->d_prop_lp;
    ma.prop_qdiff = h->d_prop_q; ma.acc_count = h->d_acc_count; ma.block = blk; ma.step = step;
    ma.prop_chunk = mc;
    k_mutate<1><<<h->nb_mut, h->mut_T, h->mut_lds, h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, 0);
    HIP_TRY(hipEventRecord(b->ev_prop, h->stream));
    HIP_TRY(hipStreamWaitEvent(b->s_down, b->ev_prop, 0));
    int c = 0;
    for (long long a = 0; a < n; a += mc, ++c) {
        const long long len = std::min(mc, n - a);
        HIP_TRY(hipMemcpyAsync(b->h_prop + a * (d + 1), h->d_prop + a * (d + 1), sizeof(double) * len * (d + 1), hipMemcpyDeviceToHost, b->s_down));
        HIP_TRY(hipEventRecord(b->ev_chunk[c], b->s_down));
    }
    b->phase_ms[CBP_ENQUEUE] += cb_now_ms() - t0;
    return 0;
}
static int host_mutation(smcmi_handle *h, const smcmi_run_config *rc, bool tempered, bool first_enqueued = false) {
    CallbackBuffers *b = h->cbuf;
    const long long n = h->n;
    const int d = h->d;
    const int K = callback_chunks(n);
    const long long mc = (n + K - 1) / K;
    if (ensure_split_buffers(h)) return SMCMI_ERR_HIP;
    for (int step = 0; step < rc->n_mh_steps; ++step)
        for (int blk = 0; blk < rc->n_blocks; ++blk) {
            if (!(first_enqueued && step == 0 && blk == 0)) { if (int e = host_propose_enqueue(h, step, blk)) return e; }
            int c = 0;
            for (long long a = 0; a < n; a += mc, ++c) {
                const long long len = std::min(mc, n - a);
                double t0 = cb_now_ms();
                HIP_TRY(hipEventSynchronize(b->ev_chunk[c]));
                b->phase_ms[c == 0 ? CBP_FIRST : CBP_WAIT] += cb_now_ms() - t0;
                const double *blk_theta = b->h_prop + a * (d + 1), *blk_lp = blk_theta + len * d;
                if (int e = eval_callback(h, 0, blk_theta, blk_lp, b->h_lik[0] + a, len)) return e;
                if (tempered) { if (int e = eval_callback(h, 1, blk_theta, blk_lp, b->h_lik[1] + a, len)) return e; }
                t0 = cb_now_ms();
                HIP_TRY(hipMemcpyAsync(h->d_lik_new + a, b->h_lik[0] + a, sizeof(double) * len, hipMemcpyHostToDevice, b->s_up));
                if (tempered) HIP_TRY(hipMemcpyAsync(h->d_lik_old + a, b->h_lik[1] + a, sizeof(double) * len, hipMemcpyHostToDevice, b->s_up));
                b->phase_ms[CBP_ENQUEUE] += cb_now_ms() - t0;
            }
            const double t0 = cb_now_ms();
            HIP_TRY(hipEventRecord(b->ev_up, b->s_up));
            HIP_TRY(hipStreamWaitEvent(h->stream, b->ev_up, 0));
            MutArgs ma{};
            ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.proposals = h->d_prop; ma.prop_logprior = h->d_prop_lp;
            ma.prop_qdiff = h->d_prop_q; ma.acc_count = h->d_acc_count; ma.block = blk; ma.step = step;
            ma.prop_chunk = mc;
            ma.lik_new = h->d_lik_new; ma.lik_old_new = tempered ? h->d_lik_old : nullptr;
            ma.last = (step == rc->n_mh_steps - 1 && blk == rc->n_blocks - 1) ? 1 : 0;
            if (h->cb_energy && ma.last) { ma.esum = h->d_esum_part; ma.emax = h->d_emax_part; }
            k_mutate<2><<<h->nb_mut, h->mut_T, h->mut_lds, h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, 0);
            // (the next propose overwrites d_prop / the pinned buffers: in stream order behind this accept; the copy streams are drained)
            b->phase_ms[CBP_ENQUEUE] += cb_now_ms() - t0;
        }
    return 0;
}

// loglh (and old_loglh) columns of the handle's cloud from the callbacks: initial clouds whose parameter columns were uploaded
// without likelihood values (`smcmi_eval_cloud_callback`), and initialize_likelihoods! (src/initialization.jl:153-186)
static int callback_fill_loglh(smcmi_handle *h, int which, int column) {
    if (int e = ensure_callback_buffers(h)) return e;
    CallbackBuffers *b = h->cbuf;
    const long long n = h->n;
    const int d = h->d;
    HIP_TRY(hipMemcpyAsync(b->h_prop, h->cl.buf[0], sizeof(double) * n * d, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b->h_lp, h->cl.buf[0] + (long long)(d + 1) * n, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));   // logprior: -Inf = out of bounds
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int e = eval_callback(h, which, b->h_prop, b->h_lp, b->h_lik[0])) return e;
    HIP_TRY(hipMemcpyAsync(h->cl.buf[0] + (long long)column * n, b->h_lik[0], sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// initial_draw! (src/initialization.jl:88-119) with a host likelihood: the device draws the priors on the build's RNG contract
// (attempt a of particle i = the draws k_init_prior makes on its a-th outer attempt), the callback scores them, particles without
// a finite log-likelihood are redrawn (one_draw's loop, :23-63) - the cloud a device family with the same values would start from.
static int callback_init_from_prior(smcmi_handle *h) {
    // (the Gamma-family draws keep their acceptance uniforms in tag k | 64: beyond 64 parameters that collides with parameter k's normal)
    if (h->d > 64) return set_err(SMCMI_ERR_UNSUPPORTED, "device prior draws serve n_para <= 64");
    if (int e = ensure_callback_buffers(h)) return e;
    if (ensure_split_buffers(h)) return SMCMI_ERR_HIP;
    CallbackBuffers *b = h->cbuf;
    const long long n = h->n;
    const int d = h->d;
    std::vector<int> attempt((size_t)n, 0);
    std::vector<double> gate((size_t)n);
    long long todo = n;
    for (int round = 0; todo > 0; ++round) {
        if (round > 100000) return set_err(SMCMI_ERR_STATE, "initial draw: no finite-likelihood draw found");
        HIP_TRY(hipMemcpyAsync(h->d_acc_count, attempt.data(), sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
        k_draw_prior<<<(unsigned)((n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_model, h->cfg.seed, h->cfg.gid0, h->d_acc_count);
        HIP_TRY(hipMemcpyAsync(b->h_prop, h->cl.buf[0], sizeof(double) * n * d, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(b->h_lp, h->cl.buf[0] + (long long)(d + 1) * n, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (long long i = 0; i < n; ++i) gate[i] = (attempt[i] >= 0 && b->h_lp[i] != -HUGE_VAL) ? 0.0 : -HUGE_VAL;     // evaluate the fresh, in-bounds draws only
        std::vector<double> keep(b->h_lik[1], b->h_lik[1] + n);           // log-likelihoods of the particles already done
        if (int e = eval_callback(h, 0, b->h_prop, gate.data(), b->h_lik[0])) return e;
        todo = 0;
        for (long long i = 0; i < n; ++i) {
            if (attempt[i] < 0) { b->h_lik[0][i] = keep[i]; continue; }
            const double ll = b->h_lik[0][i];
            if (ll == -HUGE_VAL || ll != ll) { attempt[i] += 1; ++todo; }
            else attempt[i] = -1;
        }
        memcpy(b->h_lik[1], b->h_lik[0], sizeof(double) * n);
    }
    HIP_TRY(hipMemcpyAsync(h->cl.buf[0] + (long long)d * n, b->h_lik[0], sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// The whole loop with host likelihoods: engine 1's full stage (src/smc_main.jl:377-508 in its kernel sequence) up to the proposal
// set-up, the callback mutation, one host sync per stage (the callback needs the proposals on the host anyway).
static int run_callback(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res) {
    const int nf = h->h_model.n_free;
    if (rc->n_blocks < 1 || rc->n_blocks > nf || ((nf + rc->n_blocks - 1) / rc->n_blocks) * (rc->n_blocks - 1) >= nf)
        return set_err(SMCMI_ERR_ARG, "n_blocks incompatible with the number of free parameters");
    if (rc->n_phi < 2 || rc->n_mh_steps < 1) return set_err(SMCMI_ERR_ARG, "bad n_phi / n_mh_steps");
    if (rc->resampling_method != SMCMI_RESAMPLE_SYSTEMATIC && rc->resampling_method != SMCMI_RESAMPLE_MULTINOMIAL)
        return set_err(SMCMI_ERR_ARG, "Invalid resampler in SMC. Options are systematic or multinomial");
    const bool adaptive = !rc->use_fixed_schedule;
    const bool tempered = h->cb[1] != nullptr;
    if (!adaptive && rc->n_phi > h->cfg.max_stages) return set_err(SMCMI_ERR_CAPACITY, "max_stages < n_phi");
    if (int e = ensure_callback_buffers(h)) return e;
    if (pull_state(h)) return SMCMI_ERR_HIP;
    std::vector<double> sched(rc->n_phi);
    for (int k = 0; k < rc->n_phi; ++k) sched[k] = pow((double)k / (double)(rc->n_phi - 1), rc->lambda);
    if (upload_sched(h, sched.data(), rc->n_phi)) return SMCMI_ERR_HIP;
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
    const bool cont = rc->continue_run != 0;
    if (cont) {
        if (s.stage < 1 || s.stage >= h->cfg.max_stages) return set_err(SMCMI_ERR_STATE, "no loop state to continue from");
        if (s.phi_n >= 1.0) return set_err(SMCMI_ERR_STATE, "the run to continue has already reached phi = 1");
        s.rp = rp; s.done = 0; s.err = 0; s.skip_fold = 1; s.do_resample = 0;
    } else {
        const int cur = s.cur;
        memset(&s, 0, sizeof(DevState));
        s.e_seen = __builtin_nan("");
        s.rp = rp; s.cur = cur;
        s.stage = 1; s.j = 2;
        s.c = rc->c; s.accept = rc->target;
        s.ess_prev = rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts;
    }
    const int base = cont ? s.stage - 1 : 0;
    if (push_state(h)) return SMCMI_ERR_HIP;
    if (!cont) {
        const double v0[4] = {0.0, rc->initial_ess > 0.0 ? rc->initial_ess : (double)h->cfg.n_parts, rc->c, rc->target};
        HIP_TRY(hipMemcpyAsync(h->rec.phi, &v0[0], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.ess, &v0[1], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.c, &v0[2], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->rec.accept, &v0[3], sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(h->rec.resampled, 0, sizeof(int) * h->cfg.max_stages, h->stream));
        if (h->cfg.store_history) {
            HIP_TRY(hipMemsetAsync(h->d_hist_w, 0, sizeof(double) * h->n, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_hist_W, h->cl.buf[0] + (long long)(h->R - 1) * h->n, sizeof(double) * h->n, hipMemcpyDeviceToDevice, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->rng_ahead = false;
    // adaptive schedules: the accept launches leave the energy power sums the phi predictor reads, so a stage's certificate search starts
    // from rings around the predicted root - one pass instead of six - and the energy maxima the shifted weights need
    h->cb_energy = adaptive && !getenv("SMCMI_NO_PREDICTOR");
    const int acc_nb = h->nb_mut;                       // the split kernels are the generic (LDS) mutation kernels
    k_energy_max<<<acc_nb, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_emax_part);
    const int first_passes = std::max(rc->solver_passes, FIRST_SOLVER_PASSES);
    const auto t0 = std::chrono::steady_clock::now();
    const int max_iter = (adaptive ? h->cfg.max_stages : rc->n_phi - 1) - base;
    h->cb_calls = 0; h->cb_evals = 0;
    for (double &p : h->cbuf->phase_ms) p = 0.0;
    res->solver_stalls = 0; res->select_stalls = 0; res->spec_stalls = 0;
    int launched = 0, done = 0, had = first_passes;
    DevState head;
    constexpr size_t head_off = offsetof(DevState, stage), head_len = offsetof(DevState, ess) - offsetof(DevState, stage);
    while (launched < max_iter && !done) {
        // the stage up to the proposal set-up (no mutation kernel: host_mutation below); no energy sums exist for a predictor
        const double ts0 = cb_now_ms();
        const int passes = (h->cb_energy && launched >= 2) ? std::max(1, rc->solver_passes) : first_passes;
        enqueue_stage(h, adaptive, passes, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, nullptr, nullptr, 0, false, false, false, false, true);
        had = passes;
        bool first_enqueued = false;
        for (;;) {
            HIP_TRY(hipMemcpyAsync((char *)&head + head_off, (const char *)h->d_st + head_off, head_len, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(h->cbuf->ev_head, h->stream));
            // the first proposal goes out before the verdict is read (a stage that stalled or ended leaves it a no-op)
            if (ensure_split_buffers(h)) return SMCMI_ERR_HIP;
            if (int e = host_propose_enqueue(h, 0, 0)) return e;
            first_enqueued = true;
            HIP_TRY(hipEventSynchronize(h->cbuf->ev_head));
            done = head.done;
            if (done != 2) break;
            first_enqueued = false;
            HIP_TRY(hipStreamSynchronize(h->cbuf->s_down));          // (the dropped proposal's copies: the pinned buffers are reused below)
            if (had > 1200) return set_err(SMCMI_ERR_BRACKET, "adaptive tempering solver: the search for phi_n does not terminate (the ESS objective is not a number?)");
            // the solver ran out of passes: continue the same search with more (smcmi_run)
            const int zero = 0;
            HIP_TRY(hipMemcpyAsync(&h->d_st->done, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
            enqueue_stage(h, adaptive, 8, rc->resampling_method, rc->n_blocks, rc->alpha, acc_nb, nullptr, nullptr, had, false, false, false, false, true);
            had += 8;
            res->solver_stalls += 1;
        }
        h->cbuf->phase_ms[CBP_STAGE] += cb_now_ms() - ts0;
        if (done) break;                                 // ϕ = 1 was reached by the previous stage (its begin raised the flag), a pause, or an error
        if (int e = host_mutation(h, rc, tempered, first_enqueued)) return e;
        ++launched;
    }
    k_stage_begin<<<1, BT, 0, h->stream>>>(h->d_st, h->d_sched, h->d_acc_part, acc_nb, h->rec);
    HIP_TRY(hipStreamSynchronize(h->cbuf->s_down));                  // (a proposal enqueued for a stage that turned out to be the end of the run)
    if (pull_state(h)) return SMCMI_ERR_HIP;
    const auto t1 = std::chrono::steady_clock::now();
    res->kernel_ms_mutate = 0.0; res->n_mutate_launches = 0;
    res->n_stages = s.stage; res->resamples = s.resamples; res->logmdd = s.logz; res->c = s.c; res->accept = s.accept;
    res->seconds = std::chrono::duration<double>(t1 - t0).count();
    res->solver_passes = s.solver_passes;
    res->paused = (s.done == 5) ? 1 : 0;
    h->last_n_stages = s.stage;
    if (s.err == SMCMI_ERR_NAN_ESS) return nan_ess_error(h, h->cl.buf[0] + (long long)(h->R - 1) * h->n);
    if (s.err) return err_from_state(s.err);
    if (!s.done) return set_err(SMCMI_ERR_CAPACITY, "max_stages exceeded before the tempering schedule reached 1");
    return 0;
}
