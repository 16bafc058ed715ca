// smcmi.hip - host side of libsmcmi.so: the C ABI of include/smcmi.h and the stage driver.
//
// The driver reproduces the loop order of the reference exactly (src/smc_main.jl:377-508):
//   ϕ selection -> correction -> ESS / NaN guard -> selection -> c update -> weighted moments -> random blocks
//   -> mutation -> acceptance rate,
// but as a fixed sequence of HIP kernels whose branches are resolved on the device (DevState), so the host never
// waits on the GPU inside a stage.  There is NO CPU fallback: every entry point fails with SMCMI_ERR_HIP when no
// gfx950 device is usable.
#include "handle.hpp"

static void free_eng2(Eng2 *e);
static void free_callback_buffers(CallbackBuffers *b);

static thread_local std::string g_err;
extern "C" const char *smcmi_last_error(void) { return g_err.c_str(); }
extern "C" int smcmi_version(void) { return 1; }

static int set_err(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return set_err(SMCMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)


static int push_state(smcmi_handle *h) {
    HIP_TRY(hipMemcpyAsync(h->d_st, &h->h_st, sizeof(DevState), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
static int pull_state(smcmi_handle *h) {
    HIP_TRY(hipMemcpyAsync(&h->h_st, h->d_st, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->h_st.e_shift = 0.0;      // the energy shift belongs to a running stage chain (k_stage_begin): stand-alone calls that push this copy back are unshifted
    return 0;
}
static int push_model(smcmi_handle *h) {
    HIP_TRY(hipMemcpyAsync(h->d_model, &h->h_model, sizeof(ModelDev), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
template <class T>
static int dmalloc(T **p, size_t count) {
    HIP_TRY(hipMalloc((void **)p, (count ? count : 1) * sizeof(T)));
    // development (SMCMI_POISON_ALLOC=1): fresh device memory reads as NaN / -1, so a read of something never written shows up
    // in every test instead of depending on what the allocator hands back
    static const int poison = getenv("SMCMI_POISON_ALLOC") ? atoi(getenv("SMCMI_POISON_ALLOC")) : 0;
    static int counter = 0;
    const int idx = counter++;
    if (poison) {
        HIP_TRY(hipMemset(*p, 0xFF, (count ? count : 1) * sizeof(T)));
        HIP_TRY(hipDeviceSynchronize());        // (the fill runs on the null stream: it must not land after the handle's first copies)
        if (poison > 1) fprintf(stderr, "[smcmi] poisoned allocation #%d (%zu bytes)\n", idx, (count ? count : 1) * sizeof(T));
    }
    return 0;
}
static int err_from_state(int code) {
    switch (code) {
    case 0: return 0;
    case SMCMI_ERR_NAN_ESS: return set_err(code, "No particles have non-zero weight (ESS is NaN)");
    case SMCMI_ERR_POSDEF: return set_err(code, "PosDefException: block proposal covariance is not positive definite");
    case SMCMI_ERR_CAPACITY: return set_err(code, "max_stages exceeded");
    case SMCMI_ERR_BRACKET: return set_err(code, "adaptive tempering solver: [phi_n1, phi_prop] does not bracket the ESS target (or the allotted passes ran out)");
    default: return set_err(code, "device error");
    }
}

// check_nan_ess (src/helpers.jl:270-305): the reference's diagnosis of "ESS is NaN", from the unnormalised weights W̃ the failing
// correction left (`wbuf`: the scratch column of engine 2 / of a predicted stage, else the cloud's weight column) and the stage's
// incremental weights where the history is stored.
static int nan_ess_message(long long n_parts, const std::vector<double> &inc, const std::vector<double> &w, bool ok) {
    std::string msg = "No particles have non-zero weight.";
    if (ok) {
        bool any_inf = false, any_nan = false;
        double s = 0.0, s2 = 0.0;
        for (double v : inc) { any_inf |= std::isinf(v); any_nan |= std::isnan(v); }
        for (double v : w) s += v;
        for (double v : w) { const double u = (double)n_parts * v / s; s2 += u * u; }
        if (any_inf) msg += " Some particles have approximately infinite log-likelihoods.";
        if (any_nan) msg += " Some particles have approximately NaN log-likelihoods.";
        if (s2 <= 2.220446049250313e-16) msg += " The squared sum of the normalized weights is at machine-error.";
        if (std::isnan(s2)) {
            msg += " The squared sum of the normalized weights is returning a NaN.";
            if (any_nan || !(s > 0.0)) msg += " Part of the reason is that one of the normalized weights is a NaN";
        }
    }
    return set_err(SMCMI_ERR_NAN_ESS, msg + " (ESS is NaN)");
}
static int nan_ess_error(smcmi_handle *h, const double *wbuf) {
    std::vector<double> w((size_t)h->n);
    const bool ok = hipMemcpy(w.data(), wbuf, sizeof(double) * h->n, hipMemcpyDeviceToHost) == hipSuccess;
    return nan_ess_message(h->cfg.n_parts, w, w, ok);
}
// The same diagnosis for a stage that failed INSIDE a persistent segment (engine 3): its unnormalised weights lived in registers only.
// The segment leaves the cloud as stage n - 1 completed it, so the failing correction (src/smc_main.jl:401-420) is repeated here, on
// the host, for the message alone.
static int nan_ess_error_from_cloud(smcmi_handle *h, double phi_n, double phi_prev, double pw, double logp_old) {
    const long long n = h->n;
    const int d = h->d;
    std::vector<double> lk((size_t)n), old((size_t)n), W((size_t)n), inc((size_t)n), w((size_t)n);
    const double *c = h->cl.buf[0];
    const bool ok = hipMemcpy(lk.data(), c + (long long)d * n, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(old.data(), c + (long long)(d + 2) * n, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(W.data(), c + (long long)(d + 4) * n, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok)
        for (long long i = 0; i < n; ++i) {
            const double dphi = phi_n - phi_prev;
            if (pw == 0.0) inc[i] = exp(-dphi * old[i] + dphi * lk[i]);
            else if (pw == 1.0) inc[i] = exp(dphi * lk[i]);
            else inc[i] = exp(-dphi * log(exp(old[i] - logp_old + log(1.0 - pw)) + pw) + dphi * lk[i]);
            w[i] = W[i] * inc[i];
        }
    return nan_ess_message(h->cfg.n_parts, inc, w, ok);
}

static int set_mutate_attrs(smcmi_handle *h);
static void smcmi_comm_release(smcmi_handle *h);

// ------------------------------------------------------------------------------------------------ lifetime
extern "C" int smcmi_create(const smcmi_config *cfg, smcmi_handle **out) {
    if (!cfg || !out) return set_err(SMCMI_ERR_ARG, "null argument");
    if (cfg->n_para < 1 || cfg->n_para > SMCMI_MAX_PARA) return set_err(SMCMI_ERR_ARG, "n_para out of range (1..SMCMI_MAX_PARA)");
    if (cfg->n_parts < 1) return set_err(SMCMI_ERR_ARG, "n_parts must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return set_err(SMCMI_ERR_HIP, "no HIP device available: libsmcmi has no CPU fallback (hipGetDeviceCount failed)");
    if (cfg->device < 0 || cfg->device >= ndev) return set_err(SMCMI_ERR_ARG, "bad device ordinal");
    HIP_TRY(hipSetDevice(cfg->device));
    smcmi_handle *h = new smcmi_handle();
    h->cfg = *cfg;
    if (h->cfg.n_local <= 0) h->cfg.n_local = cfg->n_parts;
    if (h->cfg.max_stages < 2) h->cfg.max_stages = 2;
    h->n = h->cfg.n_local;
    h->d = cfg->n_para;
    h->R = h->d + 5;
    h->npairs = (h->d + 1) * (h->d + 2) / 2;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const long long n = h->n;
    for (int b = 0; b < 2; ++b) {
        if (dmalloc(&h->cl.buf[b], (size_t)n * h->R)) return SMCMI_ERR_HIP;
        HIP_TRY(hipMemsetAsync(h->cl.buf[b], 0, (size_t)n * h->R * sizeof(double), h->stream));
    }
    h->cl.n = n;
    h->cl.R = h->R;
    if (dmalloc(&h->d_st, 1) || dmalloc(&h->d_model, 1)) return SMCMI_ERR_HIP;
    const int ms = h->cfg.max_stages;
    if (dmalloc(&h->rec.phi, ms) || dmalloc(&h->rec.ess, ms) || dmalloc(&h->rec.c, ms) || dmalloc(&h->rec.accept, ms) ||
        dmalloc(&h->rec.resampled, ms))
        return SMCMI_ERR_HIP;
    h->nb_e = (int)std::min<long long>(512, std::max<long long>(1, (n + 511) / 512));      // (512 rows, not 1024: less for the prepare launch to total - measured 2-4 % per run from 4e5 to 1e7 particles)
    h->nb_m = (int)std::min<long long>(256, std::max<long long>(1, (n + MT - 1) / MT));
    h->nb_mr = (int)std::min<long long>(512, std::max<long long>(std::min<long long>(64, (n + TB - 1) / TB), n / 1024));
    // mutation block size: largest of 256/128/64 threads whose per-thread LDS vectors fit 64 KiB
    for (int T : {256, 128, 64}) {
        h->mut_T = T;
        h->mut_lds = (size_t)(4 * h->d * T + T / 64) * sizeof(double) + (h->d <= 13 ? 64 + sizeof(MutStage) : 0);   // (+ staged proposal constants)
        if (h->mut_lds <= 64 * 1024) break;
    }
    h->nb_mut = (int)((n + h->mut_T - 1) / h->mut_T);
    h->nb_mut_ls4 = (int)((n + 63) / 64);              // lane-split mutation (lgss_kalman): 64 particles per 256-thread block
    // register-resident variant: only θ lives in per-thread LDS columns
    // (256-thread blocks: 782 of them at config 4's 200 000 particles sit on 1 024 slots, 14 CUs holding a fourth - measured in round 6 with 128- and
    // 64-thread blocks, which spread the wavefronts evenly: 60.7 / 61.0 / 61.1 µs per launch, no difference - the kernel is not bound by its slowest CU)
    h->reg_T = 256;
    h->nb_reg = (int)((n + h->reg_T - 1) / h->reg_T);
    h->mom_lds = (size_t)((h->d + 2) * (MT + 1)) * sizeof(double) + 2 * (size_t)h->npairs + 16;
    h->comm_cap = std::max<long long>(2 * KC, h->npairs) + 8;
    h->prep_lds = (size_t)(((h->npairs + 63) / 64) * 64 + 4 * h->d * h->d + 8) * sizeof(double);
    if (dmalloc(&h->d_part_ess[0], (size_t)h->nb_e * 2 * KC) || dmalloc(&h->d_part_ess[1], (size_t)h->nb_e * 2 * KC) || dmalloc(&h->d_part_fin, (size_t)h->nb_e * 2) || dmalloc(&h->d_part_cm, (size_t)h->nb_e * (h->npairs + 2)) || dmalloc(&h->d_prep_rows, (size_t)PREP_G * PT + 8) || dmalloc(&h->d_wt, n) ||
        dmalloc(&h->d_chunk_off, h->nb_e) || dmalloc(&h->d_cum, n) || dmalloc(&h->d_anc, n) ||
        dmalloc(&h->d_part_mom, (size_t)std::max(h->nb_m, h->nb_mr) * h->npairs) || dmalloc(&h->d_totals, h->npairs) ||
        dmalloc(&h->d_acc_part, std::max({h->nb_mut, h->nb_reg, h->nb_mut_ls4})) || dmalloc(&h->d_esum_part, (size_t)std::max({h->nb_mut, h->nb_reg, h->nb_mut_ls4}) * ES) || dmalloc(&h->d_esum_red, (size_t)ESUM_RED_ROWS * (ES + 1)) || dmalloc(&h->d_emax_part, std::max({h->nb_mut, h->nb_reg, h->nb_mut_ls4})) || dmalloc(&h->d_comm, h->comm_cap) || dmalloc(&h->d_offsets, n) ||
        dmalloc(&h->d_flag, 4) || dmalloc(&h->d_mix, (size_t)10 * (3 * 100 + 22)) || dmalloc(&h->d_mixpos, 100))
        return SMCMI_ERR_HIP;
    h->d_prep_tick = reinterpret_cast<int *>(h->d_prep_rows + (size_t)PREP_G * PT);
    HIP_TRY(hipMemset(h->d_prep_tick, 0, 8 * sizeof(double)));
    if (h->cfg.store_history) {
        if (dmalloc(&h->d_hist_w, (size_t)n * ms) || dmalloc(&h->d_hist_W, (size_t)n * ms)) return SMCMI_ERR_HIP;
    }
    memset(&h->h_st, 0, sizeof(DevState));
    h->h_st.rp.n_parts = cfg->n_parts;
    h->h_st.rp.phi_rtol = DEFAULT_PHI_RTOL;
    h->h_st.rp.max_stages = ms;
    h->h_st.stage = 1;
    h->h_st.c = 0.5;
    h->h_st.accept = 0.25;
    h->h_st.ess_prev = (double)cfg->n_parts;
    memset(&h->h_model, 0, sizeof(ModelDev));
    h->h_model.d = h->d;
    h->h_model.lik[0].family = SMCMI_LIK_NONE;
    h->h_model.lik[1].family = SMCMI_LIK_NONE;
    if (push_state(h) || push_model(h)) return SMCMI_ERR_HIP;
    HIP_TRY(hipFuncSetAttribute((const void *)k_moments, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mom_lds));
    HIP_TRY(hipFuncSetAttribute((const void *)k_prepare_mutation, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->prep_lds));
    HIP_TRY(hipFuncSetAttribute((const void *)k_mutate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mut_lds));
    HIP_TRY(hipFuncSetAttribute((const void *)k_mutate<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * mutate_wave_bytes_ls4(13) + 64 + sizeof(MutStage))));
    HIP_TRY(hipFuncSetAttribute((const void *)k_mutate<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mut_lds));
    HIP_TRY(hipFuncSetAttribute((const void *)k_mutate<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mut_lds));
    if (set_mutate_attrs(h)) return SMCMI_ERR_HIP;
    *out = h;
    return 0;
}

extern "C" int smcmi_destroy(smcmi_handle *h) {
    if (!h) return 0;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->nccl) smcmi_comm_release(h);
    if (h->e2) { free_eng2(h->e2); h->e2 = nullptr; }
    for (void *p : h->ipc_opened) hipIpcCloseMemHandle(p);
    h->ipc_opened.clear();
    if (h->d_mbox) { hipFree(h->d_mbox); h->d_mbox = nullptr; }
    if (h->d_peers) { hipFree(h->d_peers); h->d_peers = nullptr; }
    if (h->cbuf) { free_callback_buffers(h->cbuf); h->cbuf = nullptr; }
    if (h->h_note) { hipHostFree((void *)h->h_note); h->h_note = nullptr; h->d_note = nullptr; }
    void *ptrs[] = {h->cl.buf[0], h->cl.buf[1], h->d_st, h->d_model, h->d_data[0], h->d_data[1], h->d_aux[0], h->d_aux[1],
                    h->rec.phi, h->rec.ess, h->rec.c, h->rec.accept, h->rec.resampled, h->d_sched, h->d_part_ess[0], h->d_part_ess[1],
                    h->d_part_fin, h->d_part_cm, h->d_prep_rows, h->d_wt, h->d_chunk_off, h->d_cum, h->d_anc, h->d_part_mom, h->d_totals, h->d_acc_part, h->d_esum_part, h->d_esum_red, h->d_emax_part, h->d_zbuf,
                    h->d_comm, h->d_offsets, h->d_hist_w, h->d_hist_W, h->d_prop, h->d_prop_lp, h->d_prop_q,
                    h->d_lik_new, h->d_lik_old, h->d_acc_count, h->d_flag, h->d_cum_full, h->d_part_full, h->d_off_full,
                    h->d_tot_ess, h->d_tot_fin, h->d_tot_mom, h->d_tot_acc, h->d_full_w, h->d_full_cloud, h->d_prof, h->d_mix, h->d_mixpos, h->d_snap};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

// ------------------------------------------------------------------------------------------------ model
extern "C" int smcmi_set_parameters(smcmi_handle *h, const int32_t *fixed, const double *lo, const double *hi,
                                    const int32_t *prior_family, const double *prior_a, const double *prior_b) {
    if (!h || !lo || !hi || !prior_family || !prior_a || !prior_b) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    ModelDev &m = h->h_model;
    m.n_free = 0;
    m.has_other_priors = 0;
    for (int k = 0; k < h->d; ++k) {
        m.fixed[k] = fixed ? fixed[k] : 0;
        m.lo[k] = lo[k]; m.hi[k] = hi[k];
        m.prior_family[k] = prior_family[k];
        m.prior_a[k] = prior_a[k]; m.prior_b[k] = prior_b[k];
        m.prior_k[k] = m.fixed[k] ? 0.0 : prior_const_host(prior_family[k], prior_a[k], prior_b[k]);
        if (!m.fixed[k]) {
            if (prior_family[k] < SMCMI_PRIOR_NORMAL || prior_family[k] > SMCMI_PRIOR_ROOTINVGAMMA)
                return set_err(SMCMI_ERR_ARG, "unknown prior family");
            m.free_inds[m.n_free++] = k;
            if (prior_family[k] > SMCMI_PRIOR_UNIFORM) m.has_other_priors = 1;
        }
    }
    if (m.n_free == 0) return set_err(SMCMI_ERR_ARG, "All model parameters are fixed!");   // smc_main.jl:237
    h->have_params = true;
    return push_model(h);
}

// Tempered updates whose old vintage is a prefix of the new data (the reference's use case: new observations appended) and whose
// old likelihood is the same state-space model: the Kalman filter over the data passes through the old data's log-likelihood on
// its way (bit for bit: same x_0, P_0, same steps), so the mutation evaluates one filter instead of two (SMCMI_NO_LIK_PREFIX=1 keeps two).
static void update_lik_prefix(smcmi_handle *h) {
    static const int off = getenv("SMCMI_NO_LIK_PREFIX") ? atoi(getenv("SMCMI_NO_LIK_PREFIX")) : 0;
    const LikDev &a = h->h_model.lik[0], &b = h->h_model.lik[1];
    const std::vector<double> &da = h->lik_host_data[0], &db = h->lik_host_data[1];
    bool ok = !off && a.family == SMCMI_LIK_LGSS_KALMAN && b.family == SMCMI_LIK_LGSS_KALMAN && a.rows == b.rows && b.cols >= 1 && b.cols <= a.cols &&
              a.n_par == b.n_par && !da.empty() && !db.empty() && h->lik_host_aux[0] == h->lik_host_aux[1];
    for (int k = 0; ok && k < a.n_par; ++k) ok = a.par[k] == b.par[k];
    if (ok) ok = memcmp(da.data(), db.data(), sizeof(double) * db.size()) == 0;
    h->h_model.lik_prefix = ok ? (int)b.cols : 0;
}

extern "C" int smcmi_set_likelihood(smcmi_handle *h, int32_t which, int32_t family, const double *par, int64_t n_par,
                                    const double *data, int64_t rows, int64_t cols, const double *aux, int64_t aux_rows,
                                    int64_t aux_cols) {
    if (!h || which < 0 || which > 1) return set_err(SMCMI_ERR_ARG, "bad argument");
    if (n_par > LIK_PAR_MAX) return set_err(SMCMI_ERR_ARG, "too many likelihood parameters");
    HIP_TRY(hipSetDevice(h->cfg.device));
    LikDev &l = h->h_model.lik[which];
    if (h->d_data[which]) { hipFree(h->d_data[which]); h->d_data[which] = nullptr; }
    if (h->d_aux[which]) { hipFree(h->d_aux[which]); h->d_aux[which] = nullptr; }
    memset(&l, 0, sizeof(LikDev));
    l.family = family;
    h->cb[which] = nullptr; h->cb_ud[which] = nullptr;            // a device family (or none) replaces a registered host callback
    if (family == SMCMI_LIK_NONE || family == SMCMI_LIK_HOST_CALLBACK) {
        h->lik_host_data[which].clear(); h->lik_host_aux[which].clear();
        update_lik_prefix(h);
        if (which == 0) h->have_lik = true;
        return push_model(h);
    }
    if (family < SMCMI_LIK_GAUSS_ISO || family > SMCMI_LIK_LGSS_KALMAN) return set_err(SMCMI_ERR_ARG, "unknown likelihood family");
    if (family == SMCMI_LIK_LGSS_KALMAN && (n_par < 1 || rows != 3 || h->d != 13 || !aux || aux_rows * aux_cols < 112))
        return set_err(SMCMI_ERR_ARG, "lgss_kalman needs kappa, data 3 x T, aux = [C 8x8 | R 8x3 | Z 3x8] (112 doubles), d = 13");
    l.n_par = (int)n_par;
    for (int k = 0; k < n_par; ++k) l.par[k] = par[k];
    const int d = h->d;
    if (family == SMCMI_LIK_GAUSS_ISO && (n_par < 1 || rows * cols < d)) return set_err(SMCMI_ERR_ARG, "gauss_iso needs sigma and d means");
    if (family == SMCMI_LIK_LINREG && (n_par < 1 || cols != 2 || d != 2)) return set_err(SMCMI_ERR_ARG, "linreg needs sigma2, data n x 2, d = 2");
    if ((family == SMCMI_LIK_LINMODEL3 || family == SMCMI_LIK_CAPM_LITERAL) &&
        (rows != 3 || d != 9 || !aux || aux_cols < cols || (family == SMCMI_LIK_LINMODEL3 && aux_rows != 3)))
        return set_err(SMCMI_ERR_ARG, "3-equation families need data 3 x T, regressors with >= T columns, d = 9");
    l.c0 = lik_const_host(family, l.par, d);
    if (data && rows * cols > 0) {
        if (dmalloc(&h->d_data[which], (size_t)(rows * cols))) return SMCMI_ERR_HIP;
        HIP_TRY(hipMemcpy(h->d_data[which], data, sizeof(double) * rows * cols, hipMemcpyHostToDevice));
    }
    if (family == SMCMI_LIK_LGSS_KALMAN) {
        // structure block + the wave-uniform products the filter would otherwise rebuild every step (model.hpp kalman_lgss)
        std::vector<double> ext(KALMAN_AUX_TOTAL);
        for (int k = 0; k < KALMAN_AUX_USER; ++k) ext[k] = aux[k];
        const double kappa = par[0], *Rm = aux + 64;
        for (int k = 0; k < 64; ++k) ext[KALMAN_AUX_KC + k] = kappa * aux[k];
        for (int i = 0; i < 8; ++i)
            for (int j = i; j < 8; ++j)
                for (int m = 0; m < 3; ++m) ext[KALMAN_AUX_RR + ksym(i, j) * 3 + m] = Rm[i * 3 + m] * Rm[j * 3 + m];
        if (dmalloc(&h->d_aux[which], ext.size())) return SMCMI_ERR_HIP;
        HIP_TRY(hipMemcpy(h->d_aux[which], ext.data(), sizeof(double) * ext.size(), hipMemcpyHostToDevice));
        aux_rows = 1; aux_cols = KALMAN_AUX_TOTAL;
    } else if (aux && aux_rows * aux_cols > 0) {
        if (dmalloc(&h->d_aux[which], (size_t)(aux_rows * aux_cols))) return SMCMI_ERR_HIP;
        HIP_TRY(hipMemcpy(h->d_aux[which], aux, sizeof(double) * aux_rows * aux_cols, hipMemcpyHostToDevice));
    }
    l.data = h->d_data[which]; l.rows = rows; l.cols = cols;
    l.aux = h->d_aux[which]; l.aux_rows = aux_rows; l.aux_cols = aux_cols;
    h->lik_host_data[which].clear(); h->lik_host_aux[which].clear();
    if (family == SMCMI_LIK_LGSS_KALMAN && data && rows * cols > 0) {
        h->lik_host_data[which].assign(data, data + rows * cols);
        h->lik_host_aux[which].assign(aux, aux + KALMAN_AUX_USER);
    }
    update_lik_prefix(h);
    if (which == 0) h->have_lik = true;
    HIP_TRY(hipDeviceSynchronize());          // (the data / structure copies above ran on the null stream)
    return push_model(h);
}

// ------------------------------------------------------------------------------------------------ cloud transfer
extern "C" int smcmi_upload_cloud(smcmi_handle *h, const double *particles) {
    if (!h || !particles) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    HIP_TRY(hipMemcpy(h->cl.buf[h->h_st.cur], particles, sizeof(double) * h->n * h->R, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());          // (null-stream copy: not ordered with the handle's non-blocking stream)
    return 0;
}
extern "C" int smcmi_upload_cloud_device(smcmi_handle *h, const double *dev_particles) {
    if (!h || !dev_particles) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    // (the caller's pointer may be peer-device memory without peer access, or a misaligned view: the runtime's copy handles both)
    HIP_TRY(hipMemcpyAsync(h->cl.buf[h->h_st.cur], dev_particles, sizeof(double) * (size_t)h->n * h->R, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int smcmi_download_cloud(smcmi_handle *h, double *particles) {
    if (!h || !particles) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    HIP_TRY(hipMemcpy(particles, h->cl.buf[h->h_st.cur], sizeof(double) * h->n * h->R, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int smcmi_cloud_device_ptr(smcmi_handle *h, double **dev_ptr, int64_t *ld) {
    if (!h || !dev_ptr) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    *dev_ptr = h->cl.buf[h->h_st.cur];
    if (ld) *ld = h->n;
    return 0;
}
extern "C" int smcmi_sync(smcmi_handle *h) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

static inline int lik_level(int lik) { return lik; }
static int need_model(smcmi_handle *h, int lik) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    if (!h->have_params) return set_err(SMCMI_ERR_STATE, "smcmi_set_parameters has not been called");
    // lik: 1 = a device family is required (kernels evaluate it), 2 = a device family or a registered host callback
    const bool dev_ok = h->have_lik && h->h_model.lik[0].family >= 0 && h->h_model.lik[0].family != SMCMI_LIK_HOST_CALLBACK;
    if (lik && !(dev_ok || (lik_level(lik) == 2 && h->cb[0] != nullptr)))
        return set_err(SMCMI_ERR_STATE, "no likelihood set (smcmi_set_likelihood; smcmi_set_likelihood_callback for smcmi_run / smcmi_initialize_likelihoods)");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return set_err(SMCMI_ERR_HIP, "hipSetDevice failed");
    return 0;
}

static int callback_init_from_prior(smcmi_handle *h);
extern "C" int smcmi_init_from_prior(smcmi_handle *h) {
    if (int rc = need_model(h, 2)) return rc;
    if (h->cb[0]) return callback_init_from_prior(h);                 // user likelihood: device draws, host scores
    if (h->d > 64) return set_err(SMCMI_ERR_UNSUPPORTED, "device prior sampling: the RNG tags carry the parameter index in 6 bits");
    HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
    k_init_prior<<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, h->d_model, h->cfg.seed, h->cfg.gid0, h->d_flag);
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (flag) return set_err(SMCMI_ERR_STATE, "initial draw: no finite-likelihood draw found");
    return 0;
}

static int callback_fill_loglh(smcmi_handle *h, int which, int column);
static bool use_ls4_mutate(const smcmi_handle *h);
static bool use_wave_kalman(const smcmi_handle *h);
extern "C" int smcmi_initialize_likelihoods(smcmi_handle *h) {
    if (int rc = need_model(h, 2)) return rc;
    if (use_ls4_mutate(h) && !h->cb[0])
        k_initialize_likelihoods<4><<<(unsigned)((h->n + 63) / 64), 256, 64 * KALMAN4_SLOT_BYTES, h->stream>>>(h->cl, h->d_model);
    else if (use_wave_kalman(h) && !h->cb[0])
        k_initialize_likelihoods<1><<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_model);
    else
        k_initialize_likelihoods<0><<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_model);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->cb[0]) return callback_fill_loglh(h, 0, h->d);            // host likelihood: the kernel retired loglh and evaluated the prior
    return 0;
}

// ---- host likelihoods: the reference's `loglikelihood::Function` (src/smc_main.jl:118, src/mutation.jl:93-121)
extern "C" int smcmi_set_likelihood_callback(smcmi_handle *h, int32_t which, smcmi_lik_callback fn, void *user_data) {
    if (!h || which < 0 || which > 1) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->cb[which] = fn; h->cb_ud[which] = user_data;
    LikDev &l = h->h_model.lik[which];
    if (h->d_data[which]) { hipFree(h->d_data[which]); h->d_data[which] = nullptr; }
    if (h->d_aux[which]) { hipFree(h->d_aux[which]); h->d_aux[which] = nullptr; }
    memset(&l, 0, sizeof(LikDev));
    l.family = fn ? SMCMI_LIK_HOST_CALLBACK : SMCMI_LIK_NONE;
    if (which == 0) h->have_lik = fn != nullptr;
    return push_model(h);
}
extern "C" int smcmi_eval_cloud_callback(smcmi_handle *h, int32_t which, int32_t column) {
    if (int rc = need_model(h, false)) return rc;
    if (which < 0 || which > 1 || !h->cb[which]) return set_err(SMCMI_ERR_STATE, "no likelihood callback registered");
    if (column != h->d && column != h->d + 2) return set_err(SMCMI_ERR_ARG, "column must be the loglh (n_para) or old_loglh (n_para + 2) column");
    return callback_fill_loglh(h, which, column);
}
extern "C" int smcmi_callback_stats(smcmi_handle *h, int64_t *calls, int64_t *evaluations) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    if (calls) *calls = h->cb_calls;
    if (evaluations) *evaluations = h->cb_evals;
    return 0;
}

// ------------------------------------------------------------------------------------------------ stage primitives
static int upload_sched(smcmi_handle *h, const double *sched, int n_phi) {
    if (h->sched_len < n_phi) {
        if (h->d_sched) hipFree(h->d_sched);
        h->d_sched = nullptr;
        if (dmalloc(&h->d_sched, n_phi)) return SMCMI_ERR_HIP;
        h->sched_len = n_phi;
    }
    HIP_TRY(hipMemcpyAsync(h->d_sched, sched, sizeof(double) * n_phi, hipMemcpyHostToDevice, h->stream));
    return 0;
}

static const int DEFAULT_SOLVER_PASSES = 1;   // with the energy-sum predictor 1-2 passes certify the root; a stage that needs more stalls and is resumed (smcmi_run)
static const int FIRST_SOLVER_PASSES = 6;     // first two adaptive stages: no / poor prediction yet (1 schedule scan + bracketing passes)
static const int SHARDED_SOLVER_PASSES = 1;   // sharded driver: every pass costs a collective and the host syncs per stage anyway, so a stall is cheap

// P solver passes; pass p consumes the partials of pass p-1 in its prologue.  The correction pass that follows is pass P.
static void enqueue_solver(smcmi_handle *h, int passes, int p0 = 0) {
    for (int p = p0; p < passes; ++p)
        k_pass<KC, false><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, h->d_part_ess[(p + 1) & 1], h->d_part_ess[p & 1],
                                                           h->nb_e, p, nullptr, 0);
}

extern "C" int smcmi_ess_at(smcmi_handle *h, const double *phis, int32_t k, double phi_prev, double *ess_out) {
    if (!h || !phis || !ess_out || k < 1) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState saved = h->h_st;
    std::vector<double> tot(2 * KC);
    for (int base = 0; base < k; base += KC) {
        const int nv = std::min(KC, k - base);
        Solver &S = h->h_st.sol[0];
        h->h_st.done = 0; h->h_st.phi_prev = phi_prev;
        S.mode = MODE_SECTION; S.n_valid = nv;
        for (int q = 0; q < nv; ++q) S.cand[q] = phis[base + q];
        if (push_state(h)) return SMCMI_ERR_HIP;
        k_pass<KC, false><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, nullptr, h->d_part_ess[0], h->nb_e, 0, nullptr, 0);
        k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_ess[0], h->nb_e, 2 * KC, h->d_comm);
        HIP_TRY(hipMemcpyAsync(tot.data(), h->d_comm, sizeof(double) * 2 * KC, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int q = 0; q < nv; ++q) ess_out[base + q] = tot[q] * tot[q] / tot[KC + q];
    }
    h->h_st = saved;
    return push_state(h);
}

extern "C" int smcmi_solve_phi(smcmi_handle *h, const double *sched, int32_t n_phi, int32_t *j, double *phi_prop,
                               double phi_prev, double tempering_target, double ess_prev, int32_t *resampled_last,
                               double *phi_n) {
    if (!h || !sched || !j || !phi_prop || !resampled_last || !phi_n || n_phi < 2) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState saved = h->h_st;
    DevState &s = h->h_st;
    s.done = 0; s.err = 0; s.do_resample = 0; s.stage = 1; s.rp.use_fixed_schedule = 0; s.rp.n_phi = n_phi;
    s.rp.tempering_target = tempering_target; s.phi_n = phi_prev; s.phi_prop = *phi_prop; s.j = *j;
    s.resampled_last = *resampled_last; s.ess_prev = ess_prev;
    if (s.rp.phi_rtol <= 0.0) s.rp.phi_rtol = DEFAULT_PHI_RTOL;
    if (upload_sched(h, sched, n_phi) || push_state(h)) return SMCMI_ERR_HIP;
    k_stage_begin<<<1, BT, 0, h->stream>>>(h->d_st, h->d_sched, h->d_acc_part, 0, h->rec);
    // enough passes to walk the whole schedule in the worst case plus the bracketing passes
    const int passes = (n_phi + KC - 2) / (KC - 1) + 28;
    enqueue_solver(h, passes);
    k_solver_finish<<<1, TB, 0, h->stream>>>(h->d_st, h->d_sched, h->d_part_ess[(passes + 1) & 1], h->nb_e, passes);
    if (pull_state(h)) return SMCMI_ERR_HIP;
    const Solver S = s.sol[passes & 1];
    const int err = s.err, rl = s.resampled_last;
    h->h_st = saved;
    if (push_state(h)) return SMCMI_ERR_HIP;
    if (err) return err_from_state(err);
    if (S.mode != MODE_FINAL) return err_from_state(SMCMI_ERR_BRACKET);
    *phi_n = S.phi_n; *phi_prop = S.phi_prop; *j = S.j; *resampled_last = rl;
    return 0;
}

extern "C" int smcmi_correct(smcmi_handle *h, double phi_n, double phi_prev, double prior_weight, double log_prob_old_data,
                             double threshold_ratio, smcmi_stage_stats *out) {
    if (!h || !out) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    const DevState saved = s;
    s.done = 0; s.err = 0; s.phi_n = phi_n; s.phi_prev = phi_prev; s.rp.pw = prior_weight;
    s.rp.logp_old = log_prob_old_data; s.rp.threshold = threshold_ratio * (double)s.rp.n_parts; s.logz = 0.0;
    s.stage = h->cfg.max_stages;   // records of a stand-alone call land in the last (scratch) slot
    s.rp.store_history = 0;
    s.sol[0].mode = MODE_FINAL; s.sol[0].phi_n = phi_n; s.sol[0].j = s.j; s.sol[0].phi_prop = s.phi_prop;
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_pass<1, true><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, nullptr, h->d_part_fin, h->nb_e, 0, nullptr, 0);
    k_post_correct<<<1, TB, 0, h->stream>>>(h->d_st, h->d_part_fin, h->nb_e, nullptr, h->rec, 0);
    k_normalize_weights<<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, (double)h->cfg.n_parts);
    if (pull_state(h)) return SMCMI_ERR_HIP;
    out->ess = s.ess; out->sum_unnorm = s.sumw; out->logz_inc = s.logz; out->resample = s.do_resample;
    const int err = s.err;
    s = saved;
    if (push_state(h)) return SMCMI_ERR_HIP;
    return err_from_state(err);
}

extern "C" int smcmi_resample(smcmi_handle *h, int32_t method, uint32_t stage, const double *offsets, int64_t *ancestors_out) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    if (method != SMCMI_RESAMPLE_SYSTEMATIC && method != SMCMI_RESAMPLE_MULTINOMIAL)
        return set_err(SMCMI_ERR_ARG, "Invalid resampler in SMC. Options are systematic or multinomial");   // resample.jl:77
    HIP_TRY(hipSetDevice(h->cfg.device));
    const long long n = h->n;
    const double *d_off = nullptr;
    if (offsets) {
        const long long cnt = method == SMCMI_RESAMPLE_MULTINOMIAL ? n : 1;
        HIP_TRY(hipMemcpyAsync(h->d_offsets, offsets, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
        d_off = h->d_offsets;
    }
    k_weight_chunk_sums<<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_part_fin);
    k_chunk_offsets<<<1, 1, 0, h->stream>>>(h->d_st, h->d_part_fin, h->nb_e, h->d_chunk_off, 0.0, 1);
    k_scan_weights<<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_chunk_off, h->d_cum, 1, h->nb_e);
    k_resample_gather<<<(unsigned)((n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, h->d_cum, n, h->cfg.gid0, h->cfg.n_parts, method,
                                                                         h->cfg.seed, stage, d_off, h->d_anc, nullptr, 1);
    // the gathered cloud sits in buffer 1; the current cloud is always buffer 0
    HIP_TRY(hipMemcpyAsync(h->cl.buf[0], h->cl.buf[1], sizeof(double) * n * h->R, hipMemcpyDeviceToDevice, h->stream));
    if (ancestors_out) HIP_TRY(hipMemcpyAsync(ancestors_out, h->d_anc, sizeof(long long) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- tempered-update initialisation pieces (smc_main.jl:244-333): bridge resample, row copy, weight clean-up
extern "C" int smcmi_bridge_resample(smcmi_handle *dst, smcmi_handle *src, int32_t method, uint32_t stage, int64_t n_out,
                                     const double *offsets, int64_t *ancestors_out) {
    if (!dst || !src) return set_err(SMCMI_ERR_ARG, "null handle");
    if (method != SMCMI_RESAMPLE_SYSTEMATIC && method != SMCMI_RESAMPLE_MULTINOMIAL)
        return set_err(SMCMI_ERR_ARG, "Invalid resampler in SMC. Options are systematic or multinomial");
    if (dst->R != src->R || dst->cfg.device != src->cfg.device) return set_err(SMCMI_ERR_ARG, "bridge: clouds differ in n_para or device");
    if (n_out < 0 || n_out > dst->n) return set_err(SMCMI_ERR_ARG, "bridge: n_to_resample exceeds the new cloud");
    if (src->cfg.n_local != src->cfg.n_parts || dst->cfg.n_local != dst->cfg.n_parts)
        return set_err(SMCMI_ERR_ARG, "bridge: whole (unsharded) clouds only");
    if (n_out == 0) return 0;
    HIP_TRY(hipSetDevice(src->cfg.device));
    if (pull_state(src) || pull_state(dst)) return SMCMI_ERR_HIP;
    double *d_off = nullptr;
    if (offsets) {
        const long long cnt = method == SMCMI_RESAMPLE_MULTINOMIAL ? n_out : 1;
        HIP_TRY(hipMalloc(&d_off, sizeof(double) * cnt));
        HIP_TRY(hipMemcpyAsync(d_off, offsets, sizeof(double) * cnt, hipMemcpyHostToDevice, src->stream));
    }
    long long *d_anc = nullptr;
    if (ancestors_out) HIP_TRY(hipMalloc(&d_anc, sizeof(long long) * n_out));
    k_weight_chunk_sums<<<src->nb_e, TB, 0, src->stream>>>(src->cl, src->d_st, src->d_part_fin);
    k_chunk_offsets<<<1, 1, 0, src->stream>>>(src->d_st, src->d_part_fin, src->nb_e, src->d_chunk_off, 0.0, 1);
    k_scan_weights<<<src->nb_e, TB, 0, src->stream>>>(src->cl, src->d_st, src->d_chunk_off, src->d_cum, 1, src->nb_e);
    k_bridge_gather<<<(unsigned)((n_out + TB - 1) / TB), TB, 0, src->stream>>>(src->cl, src->h_st.cur, src->d_cum, src->n, dst->cl,
                                                                            dst->h_st.cur, n_out, method, src->cfg.seed, stage,
                                                                            d_off, d_anc);
    if (ancestors_out) HIP_TRY(hipMemcpyAsync(ancestors_out, d_anc, sizeof(long long) * n_out, hipMemcpyDeviceToHost, src->stream));
    HIP_TRY(hipStreamSynchronize(src->stream));
    if (d_off) hipFree(d_off);
    if (d_anc) hipFree(d_anc);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int smcmi_copy_rows(smcmi_handle *dst, int64_t dst_row0, smcmi_handle *src, int64_t src_row0, int64_t n_rows) {
    if (!dst || !src) return set_err(SMCMI_ERR_ARG, "null handle");
    if (dst->R != src->R || dst->cfg.device != src->cfg.device) return set_err(SMCMI_ERR_ARG, "copy_rows: clouds differ in n_para or device");
    if (n_rows < 0 || dst_row0 < 0 || src_row0 < 0 || dst_row0 + n_rows > dst->n || src_row0 + n_rows > src->n)
        return set_err(SMCMI_ERR_ARG, "copy_rows: row range outside the cloud");
    if (n_rows == 0) return 0;
    HIP_TRY(hipSetDevice(src->cfg.device));
    if (pull_state(src) || pull_state(dst)) return SMCMI_ERR_HIP;
    HIP_TRY(hipMemcpy2DAsync(dst->cl.buf[dst->h_st.cur] + dst_row0, sizeof(double) * dst->n, src->cl.buf[src->h_st.cur] + src_row0,
                             sizeof(double) * src->n, sizeof(double) * n_rows, (size_t)src->R, hipMemcpyDeviceToDevice, src->stream));
    HIP_TRY(hipStreamSynchronize(src->stream));
    return 0;
}

extern "C" int smcmi_normalize_weights(smcmi_handle *h, int32_t zero_bad_loglh) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const unsigned g = (unsigned)((h->n + TB - 1) / TB);
    if (zero_bad_loglh) k_zero_bad_weights<<<g, TB, 0, h->stream>>>(h->cl, h->d_st);
    k_weight_chunk_sums<<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_part_fin);
    k_chunk_offsets<<<1, 1, 0, h->stream>>>(h->d_st, h->d_part_fin, h->nb_e, h->d_chunk_off, 0.0, 1);
    k_normalize_weights<<<g, TB, 0, h->stream>>>(h->cl, h->d_st, (double)h->cfg.n_parts);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- moments launch: register-resident kernel for d <= 12, LDS-tiled kernel beyond
// fused_slot >= 0: the kernel also takes the post-correction decision itself (no k_post_correct / k_resample_gather in front)
template <int D>
static void launch_moments_reg(smcmi_handle *h, double *hist_W, int standalone, int fused_slot) {
    if (fused_slot >= 0)
        k_moments_reg<D><<<h->nb_mr, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_part_mom, hist_W, h->n, standalone, h->d_part_fin, h->nb_e,
                                                        fused_slot, h->rec);
    else
        k_moments_reg<D><<<h->nb_mr, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_part_mom, hist_W, h->n, standalone);
}
static bool can_fuse_post(const smcmi_handle *h) { return h->d <= 12; }
// the correction pass that also gathers the moments is used with the register mutation kernel (which normalises the weights)
static bool can_fuse_cm(const smcmi_handle *h) { return h->d <= 10; }
// prev / nb_prev: partials of the last solver pass (or, sharded, their all-reduced totals as one row)
template <int D>
static void launch_cm(smcmi_handle *h, int P, const double *prev, int nb_prev) {
    k_correct_moments<D><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, prev, h->d_part_fin, h->d_part_cm, nb_prev, P, h->d_hist_w, h->n,
                                                     h->spec_stage ? h->d_wt : nullptr);
}
static void launch_correct_moments(smcmi_handle *h, int P, const double *prev = nullptr, int nb_prev = 0) {
    if (!prev) { prev = h->d_part_ess[(P + 1) & 1]; nb_prev = h->nb_e; }
    switch (h->d) {
    case 1: launch_cm<1>(h, P, prev, nb_prev); break;
    case 2: launch_cm<2>(h, P, prev, nb_prev); break;
    case 3: launch_cm<3>(h, P, prev, nb_prev); break;
    case 4: launch_cm<4>(h, P, prev, nb_prev); break;
    case 5: launch_cm<5>(h, P, prev, nb_prev); break;
    case 6: launch_cm<6>(h, P, prev, nb_prev); break;
    case 7: launch_cm<7>(h, P, prev, nb_prev); break;
    case 8: launch_cm<8>(h, P, prev, nb_prev); break;
    case 9: launch_cm<9>(h, P, prev, nb_prev); break;
    default: launch_cm<10>(h, P, prev, nb_prev); break;
    }
}
// returns the number of blocks that wrote partials
static int launch_moments(smcmi_handle *h, double *hist_W, int standalone, int fused_slot = -1) {
    switch (h->d) {
    case 1: launch_moments_reg<1>(h, hist_W, standalone, fused_slot); break;
    case 2: launch_moments_reg<2>(h, hist_W, standalone, fused_slot); break;
    case 3: launch_moments_reg<3>(h, hist_W, standalone, fused_slot); break;
    case 4: launch_moments_reg<4>(h, hist_W, standalone, fused_slot); break;
    case 5: launch_moments_reg<5>(h, hist_W, standalone, fused_slot); break;
    case 6: launch_moments_reg<6>(h, hist_W, standalone, fused_slot); break;
    case 7: launch_moments_reg<7>(h, hist_W, standalone, fused_slot); break;
    case 8: launch_moments_reg<8>(h, hist_W, standalone, fused_slot); break;
    case 9: launch_moments_reg<9>(h, hist_W, standalone, fused_slot); break;
    case 10: launch_moments_reg<10>(h, hist_W, standalone, fused_slot); break;
    case 11: launch_moments_reg<11>(h, hist_W, standalone, fused_slot); break;
    case 12: launch_moments_reg<12>(h, hist_W, standalone, fused_slot); break;
    default:
        k_moments<<<h->nb_m, TB, h->mom_lds, h->stream>>>(h->cl, h->d_st, h->d_part_mom, hist_W, h->n, standalone);
        return h->nb_m;
    }
    return h->nb_mr;
}

extern "C" int smcmi_moments(smcmi_handle *h, double *mean, double *cov) {
    if (!h || !mean || !cov) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int d = h->d;
    const int nbm = launch_moments(h, nullptr, 1);
    k_moments_reduce<<<(h->npairs + 63) / 64, 1024, 0, h->stream>>>(h->d_st, h->d_part_mom, nbm, h->npairs, h->d_totals, 1);
    k_finalize_moments<<<1, 64, 0, h->stream>>>(h->d_st, h->d_totals, d);
    if (pull_state(h)) return SMCMI_ERR_HIP;
    for (int a = 0; a < d; ++a) mean[a] = h->h_st.mean[a];
    for (int e = 0; e < d * d; ++e) cov[e] = h->h_st.cov[e];
    return 0;
}

static int stage_blocks(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                        const int32_t *blocks_free, int32_t n_blocks, double c) {
    DevState &s = h->h_st;
    const ModelDev &m = h->h_model;
    const int nf = m.n_free, d = h->d;
    if (n_blocks < 1 || n_blocks > nf || block_ptr[0] != 0 || block_ptr[n_blocks] != nf) return set_err(SMCMI_ERR_ARG, "bad block structure");
    for (int a = 0; a < d; ++a) s.mean[a] = 0.0;
    for (int e = 0; e < d * d; ++e) s.cov[e] = 0.0;
    for (int a = 0; a < nf; ++a) {
        s.mean[m.free_inds[a]] = mu_free[a];
        for (int b = 0; b < nf; ++b) s.cov[m.free_inds[a] * d + m.free_inds[b]] = Sigma_free[a * nf + b];
    }
    s.n_blocks = n_blocks;
    int max_db = 0;
    for (int b = 0; b <= n_blocks; ++b) s.block_ptr[b] = block_ptr[b];
    for (int b = 0; b < n_blocks; ++b) max_db = std::max(max_db, block_ptr[b + 1] - block_ptr[b]);
    s.max_db = max_db;
    for (int i = 0; i < nf; ++i) {
        if (blocks_free[i] < 0 || blocks_free[i] >= nf) return set_err(SMCMI_ERR_ARG, "block index out of range");
        s.blocks_free[i] = blocks_free[i];
    }
    s.c = c;
    return 0;
}

// ---- mutation launch: register-resident kernel for models with n_para <= 10, generic LDS kernel beyond
static size_t reg_lds_bytes(int D) {
    return (size_t)(2 * D * D + 12 * D + 4 + 2 * LIK_PAR_MAX + LIK_LDS_CAP) * sizeof(double) + (size_t)(6 * D + 8) * sizeof(int) + 32;
}
template <int D>
static void launch_reg(smcmi_handle *h, const MutArgs &ma, int standalone) {
    if (h->launch_alpha1)
        k_mutate_reg<D, true><<<h->nb_reg, h->reg_T, reg_lds_bytes(D), h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, standalone,
                                                                                    h->launch_nb, h->h_model.n_free);
    else {
        // α < 1: the blocks' dense mixture matrices once per stage (k_mix_prepare), then the particles
        MutArgs mb = ma;
        mb.mix = h->d_mix; mb.mixpos = h->d_mixpos;
        k_mix_prepare<D><<<1, 256, 0, h->stream>>>(h->d_st, h->launch_nb, h->h_model.n_free, h->d_mix, h->d_mixpos);
        k_mutate_reg<D, false><<<h->nb_reg, h->reg_T, reg_lds_bytes(D), h->stream>>>(h->cl, h->d_st, h->d_model, mb, h->d_acc_part, standalone,
                                                                               h->launch_nb, h->h_model.n_free);
    }
}
static int set_mutate_attrs(smcmi_handle *) { return 0; }   // the register kernel needs < 64 KiB of dynamic LDS
static bool use_reg_mutate(const smcmi_handle *h) { return h->d <= 10; }
// lgss_kalman on both vintages (or no old vintage) and at most 32 768 particles on the handle: four lanes per particle (kernels.hpp
// k_mutate<0, 4>) - up to two of its wavefronts per SIMD; beyond that one thread per particle (fewer instructions per particle) is
// faster (measured: §6 of DESIGN.md).  SMCMI_KALMAN_LANES=1 / 4 forces one or the other (development / comparison).
static bool use_ls4_mutate(const smcmi_handle *h) {
    static const int lanes = getenv("SMCMI_KALMAN_LANES") ? atoi(getenv("SMCMI_KALMAN_LANES")) : 0;
    const int f0 = h->h_model.lik[0].family, f1 = h->h_model.lik[1].family;
    if (!(h->d == 13 && f0 == SMCMI_LIK_LGSS_KALMAN && (f1 == SMCMI_LIK_NONE || f1 == SMCMI_LIK_LGSS_KALMAN))) return false;
    return lanes == 4 || (lanes != 1 && h->n <= 32768);
}
// the same models on larger clouds: one thread per particle through kalman_lgss_wave
static bool use_wave_kalman(const smcmi_handle *h) {
    const int f0 = h->h_model.lik[0].family, f1 = h->h_model.lik[1].family;
    return !use_ls4_mutate(h) && h->d == 13 && f0 == SMCMI_LIK_LGSS_KALMAN && (f1 == SMCMI_LIK_NONE || f1 == SMCMI_LIK_LGSS_KALMAN);
}
// blocks (= rows of acceptance / energy partials) of the in-run mutation kernel
static int mut_blocks(const smcmi_handle *h) { return use_reg_mutate(h) ? h->nb_reg : (use_ls4_mutate(h) ? h->nb_mut_ls4 : h->nb_mut); }
// returns the number of blocks launched (= acceptance partials written)
static int launch_mutate(smcmi_handle *h, int n_blocks, int standalone, double alpha) {
    h->launch_nb = n_blocks;
    h->launch_alpha1 = (alpha == 1.0);
    MutArgs ma{};
    ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0;
    const int dbg = 0;           // (ablation bits: retired as a switch in round 6)
    ma.debug = dbg;
    ma.stage_consts = (h->d <= 13 && !(dbg & 512)) ? 1 : 0;
    ma.prof = h->d_prof;
    static const int no_pred = getenv("SMCMI_NO_PREDICTOR") ? atoi(getenv("SMCMI_NO_PREDICTOR")) : 0;   // development only
    ma.esum = (!standalone && h->run_adaptive && !no_pred) ? h->d_esum_part : nullptr;
    ma.emax = !standalone ? h->d_emax_part : nullptr;
    ma.zbuf = (!standalone && h->rng_ahead && use_reg_mutate(h)) ? h->d_zbuf : nullptr;
    ma.z_ahead = h->z_ahead;
    ma.normalize = (!standalone && h->fused_cm) ? 1 : 0;
    ma.hist_W = h->d_hist_W; ma.hist_ld = h->n;
    ma.wt = (!standalone && h->fused_cm && h->spec_stage) ? h->d_wt : nullptr;
    switch (h->d) {
    case 1: launch_reg<1>(h, ma, standalone); break;
    case 2: launch_reg<2>(h, ma, standalone); break;
    case 3: launch_reg<3>(h, ma, standalone); break;
    case 4: launch_reg<4>(h, ma, standalone); break;
    case 5: launch_reg<5>(h, ma, standalone); break;
    case 6: launch_reg<6>(h, ma, standalone); break;
    case 7: launch_reg<7>(h, ma, standalone); break;
    case 8: launch_reg<8>(h, ma, standalone); break;
    case 9: launch_reg<9>(h, ma, standalone); break;
    case 10: launch_reg<10>(h, ma, standalone); break;
    default:
        if (use_ls4_mutate(h)) {
            k_mutate<0, 4><<<h->nb_mut_ls4, 256, 4 * mutate_wave_bytes_ls4(13) + 64 + sizeof(MutStage), h->stream>>>(h->cl, h->d_st, h->d_model, ma,
                                                                                                             h->d_acc_part, standalone);
            return h->nb_mut_ls4;
        }
        k_mutate<0><<<h->nb_mut, h->mut_T, h->mut_lds, h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, standalone);
        return h->nb_mut;
    }
    return h->nb_reg;
}

extern "C" int smcmi_mutate(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                            const int32_t *blocks_free, int32_t n_blocks, double phi_n, double phi_prev, double c, double alpha,
                            int32_t n_mh_steps, uint32_t stage, double *accept_mean_out) {
    (void)phi_prev;
    if (int rc = need_model(h, true)) return rc;
    if (!mu_free || !Sigma_free || !block_ptr || !blocks_free) return set_err(SMCMI_ERR_ARG, "null argument");
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    const double c_save = s.c;
    if (int rc = stage_blocks(h, mu_free, Sigma_free, block_ptr, blocks_free, n_blocks, c)) return rc;
    s.done = 0; s.err = 0; s.do_resample = 0;
    s.mut_c = c; s.mut_alpha = alpha; s.mut_phi = phi_n; s.mut_steps = n_mh_steps; s.mut_stage = stage;
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_prepare_mutation<<<1, PT, h->prep_lds, h->stream>>>(h->d_st, h->d_model, h->d_totals, 0, h->cfg.seed, 0, 0, 1);
    const int nbl = launch_mutate(h, s.n_blocks, 1, alpha);
    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_acc_part, nbl, 1, h->d_comm);
    double asum = 0.0;
    HIP_TRY(hipMemcpyAsync(&asum, h->d_comm, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    const int err = s.err;
    s.c = c_save; s.err = 0; s.done = 0;
    if (push_state(h)) return SMCMI_ERR_HIP;
    if (err) return err_from_state(err);
    if (accept_mean_out) *accept_mean_out = asum / (double)h->n;
    return 0;
}

static int ensure_split_buffers(smcmi_handle *h) {
    if (h->d_prop) return 0;
    const long long n = h->n;
    if (dmalloc(&h->d_prop, (size_t)n * (h->d + 1)) || dmalloc(&h->d_prop_lp, n) || dmalloc(&h->d_prop_q, n) ||       // (+ 1: the chunk-major layout's log-prior column)
        dmalloc(&h->d_lik_new, n) || dmalloc(&h->d_lik_old, n) || dmalloc(&h->d_acc_count, n))
        return SMCMI_ERR_HIP;
    return 0;
}

extern "C" int smcmi_propose(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                             const int32_t *blocks_free, int32_t n_blocks, int32_t block, int32_t mh_step, double c, double alpha,
                             uint32_t stage, double *proposals_out, double *logprior_out, double *q_diff_out) {
    if (int rc = need_model(h, false)) return rc;
    if (!mu_free || !Sigma_free || !block_ptr || !blocks_free || !proposals_out) return set_err(SMCMI_ERR_ARG, "null argument");
    if (block < 0 || block >= n_blocks || mh_step < 0) return set_err(SMCMI_ERR_ARG, "bad block / step");
    if (ensure_split_buffers(h) || pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    if (int rc = stage_blocks(h, mu_free, Sigma_free, block_ptr, blocks_free, n_blocks, c)) return rc;
    s.done = 0; s.err = 0; s.do_resample = 0;
    s.mut_c = c; s.mut_alpha = alpha; s.mut_phi = 0.0; s.mut_steps = mh_step + 1; s.mut_stage = stage;
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_prepare_mutation<<<1, PT, h->prep_lds, h->stream>>>(h->d_st, h->d_model, h->d_totals, 0, h->cfg.seed, 0, 0, 1);
    MutArgs ma{};
    ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.proposals = h->d_prop; ma.prop_logprior = h->d_prop_lp;
    ma.prop_qdiff = h->d_prop_q; ma.acc_count = h->d_acc_count; ma.block = block; ma.step = mh_step;
    k_mutate<1><<<h->nb_mut, h->mut_T, h->mut_lds, h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, 1);
    HIP_TRY(hipMemcpyAsync(proposals_out, h->d_prop, sizeof(double) * h->n * h->d, hipMemcpyDeviceToHost, h->stream));
    if (logprior_out) HIP_TRY(hipMemcpyAsync(logprior_out, h->d_prop_lp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (q_diff_out) HIP_TRY(hipMemcpyAsync(q_diff_out, h->d_prop_q, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    return err_from_state(s.err);
}

extern "C" int smcmi_accept(smcmi_handle *h, const double *loglik_new, const double *loglik_old_new, double phi_n, int32_t block,
                            int32_t mh_step, int32_t n_blocks, uint32_t stage, int32_t last) {
    if (int rc = need_model(h, false)) return rc;
    if (!loglik_new || !h->d_prop) return set_err(SMCMI_ERR_STATE, "smcmi_accept needs a preceding smcmi_propose");
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    if (n_blocks != s.n_blocks) return set_err(SMCMI_ERR_ARG, "n_blocks differs from the preceding smcmi_propose");
    s.mut_phi = phi_n; s.mut_stage = stage; s.done = 0;
    if (push_state(h)) return SMCMI_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(h->d_lik_new, loglik_new, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
    if (loglik_old_new) HIP_TRY(hipMemcpyAsync(h->d_lik_old, loglik_old_new, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
    MutArgs ma{};
    ma.seed = h->cfg.seed; ma.gid0 = h->cfg.gid0; ma.proposals = h->d_prop; ma.prop_logprior = h->d_prop_lp;
    ma.prop_qdiff = h->d_prop_q; ma.lik_new = h->d_lik_new; ma.lik_old_new = loglik_old_new ? h->d_lik_old : nullptr;
    ma.acc_count = h->d_acc_count; ma.block = block; ma.step = mh_step; ma.last = last;
    k_mutate<2><<<h->nb_mut, h->mut_T, h->mut_lds, h->stream>>>(h->cl, h->d_st, h->d_model, ma, h->d_acc_part, 1);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// In-run proposal set-up: block 0 prepares the proposal, the other blocks draw the stage's random numbers ahead (RngAhead) when
// the register mutation kernel will run and the buffer fits.  from_totals as in k_prepare_mutation.
static int ensure_zbuf(smcmi_handle *h, int n_mh_steps, int n_blocks) {
    h->rng_ahead = false;
    if (!use_reg_mutate(h)) return 0;
    // Worth it only while the chip is under-occupied during the set-up launch: measured +4 % at n = 1e5, +1.5 % at 3e5, -5 % at 1e6
    // (config 2); beyond that the draws are cheaper inside the mutation kernel than a round trip through HBM.  With several
    // proposals per particle (MH steps x blocks) the first few are drawn ahead - as many as fit the window - and the rest in the
    // mutation kernel: up to 250 000 particle-proposals - about what the set-up launch's idle window (~10 µs on 255 CUs) absorbs.
    // Config 4 (3 proposals for each of 200 000 particles) per run: none ahead 30.5 ms, one 29.6-29.8, two 30.2-30.3, all three 30.7.
    const long long ahead_max = 500000;
    static const long long part_max = getenv("SMCMI_RNG_AHEAD_PART") ? atoll(getenv("SMCMI_RNG_AHEAD_PART")) : 250000;  // development only
    const int props = n_mh_steps * n_blocks;
    int k_ahead = props;
    if ((long long)h->n * props > ahead_max) k_ahead = (int)std::min<long long>(props, part_max / (long long)h->n);
    h->z_ahead = k_ahead;
    if (k_ahead < 1) return 0;
    const size_t need = (size_t)h->n * (size_t)(h->d + 2) * (size_t)k_ahead;
    if (need > h->zbuf_cap) {
        if (h->d_zbuf) { hipFree(h->d_zbuf); h->d_zbuf = nullptr; h->zbuf_cap = 0; }
        if (dmalloc(&h->d_zbuf, need)) return SMCMI_ERR_HIP;
        h->zbuf_cap = need;
    }
    h->rng_ahead = true;
    return 0;
}
static void launch_prepare_in_run(smcmi_handle *h, const double *partials, int nb_part, int from_totals, int sol_slot = 0) {
    RngAhead ra{};
    unsigned grid = 1;
    if (h->rng_ahead) {
        ra.zbuf = h->d_zbuf; ra.n = h->n; ra.gid0 = h->cfg.gid0; ra.D = h->d; ra.t_ahead = h->z_ahead;
        grid = RA_SKIP + (unsigned)((h->n + RA_T - 1) / RA_T);
    }
    // many rows: blocks 1..PREP_G of the launch total a chunk each (PrepRed, kernels.hpp) - one CU alone is bandwidth-bound on them
    PrepRed pr{};
    const int m_rows = from_totals == 3 ? h->npairs + 2 : h->npairs;
    if ((from_totals == 3 || from_totals == 1) && nb_part >= PREP_MIN_ROWS && m_rows <= PT && h->d_prep_rows) {
        pr.rows = h->d_prep_rows; pr.tick = h->d_prep_tick;
        if (grid < 1u + PREP_G) grid = 1u + PREP_G;
    }
    k_prepare_mutation<<<grid, PT, h->prep_lds, h->stream>>>(h->d_st, h->d_model, partials, nb_part, h->cfg.seed, from_totals, 1, 0, h->d_prof ? h->d_prof + 25 : nullptr, ra,
                                                             sol_slot, h->rec, pr, h->note_on ? h->d_note : nullptr);
}

#include "run1.hpp"

extern "C" int smcmi_stages_held(smcmi_handle *h, int32_t *n_stages_out) {
    if (!h || !n_stages_out) return set_err(SMCMI_ERR_ARG, "null argument");
    *n_stages_out = h->last_n_stages;
    return 0;
}

extern "C" int smcmi_get_stage_records(smcmi_handle *h, double *phi, double *ess, double *c, double *accept, int32_t *resampled) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int ns = h->last_n_stages;
    if (phi) HIP_TRY(hipMemcpy(phi, h->rec.phi, sizeof(double) * ns, hipMemcpyDeviceToHost));
    if (ess) HIP_TRY(hipMemcpy(ess, h->rec.ess, sizeof(double) * ns, hipMemcpyDeviceToHost));
    if (c) HIP_TRY(hipMemcpy(c, h->rec.c, sizeof(double) * ns, hipMemcpyDeviceToHost));
    if (accept) HIP_TRY(hipMemcpy(accept, h->rec.accept, sizeof(double) * ns, hipMemcpyDeviceToHost));
    if (resampled) HIP_TRY(hipMemcpy(resampled, h->rec.resampled, sizeof(int) * ns, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int smcmi_get_history(smcmi_handle *h, double *w, double *W) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    if (!h->cfg.store_history) return set_err(SMCMI_ERR_STATE, "history was not stored (store_history = 0)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t bytes = sizeof(double) * (size_t)h->n * h->last_n_stages;
    if (w) HIP_TRY(hipMemcpy(w, h->d_hist_w, bytes, hipMemcpyDeviceToHost));
    if (W) HIP_TRY(hipMemcpy(W, h->d_hist_W, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- intermediate save / continue (smc_main.jl:334-361, 499-507)
extern "C" int smcmi_get_loop_state(smcmi_handle *h, smcmi_loop_state *out) {
    if (!h || !out) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    const DevState &s = h->h_st;
    out->stage_index = s.stage; out->j = s.j; out->resampled_last_period = s.resampled_last; out->resamples = s.resamples;
    out->phi_n = s.phi_n; out->phi_prop = s.phi_prop; out->c = s.c; out->accept = s.accept; out->ess = s.ess_prev; out->logmdd = s.logz;
    return 0;
}

extern "C" int smcmi_set_loop_state(smcmi_handle *h, const smcmi_loop_state *in) {
    if (!h || !in) return set_err(SMCMI_ERR_ARG, "null argument");
    if (in->stage_index < 1 || in->stage_index > h->cfg.max_stages || in->j < 1 || !(in->phi_n >= 0.0 && in->phi_n <= 1.0))
        return set_err(SMCMI_ERR_ARG, "loop state out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    s.stage = in->stage_index; s.j = in->j; s.resampled_last = in->resampled_last_period ? 1 : 0; s.resamples = in->resamples;
    s.phi_n = in->phi_n; s.phi_prop = in->phi_prop; s.c = in->c; s.accept = in->accept;
    s.ess = in->ess; s.ess_prev = in->ess; s.logz = in->logmdd;
    s.done = 0; s.err = 0; s.do_resample = 0; s.cur = 0;
    s.e_seen = __builtin_nan("");               // (saved scalars carry no energy maximum: the continuation's first begin takes the cloud's)
    h->last_n_stages = in->stage_index;
    return push_state(h);
}

extern "C" int smcmi_set_stage_records(smcmi_handle *h, int32_t n_stages, const double *phi, const double *ess, const double *c,
                                       const double *accept, const int32_t *resampled) {
    if (!h || n_stages < 1 || n_stages > h->cfg.max_stages) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (phi) HIP_TRY(hipMemcpy(h->rec.phi, phi, sizeof(double) * n_stages, hipMemcpyHostToDevice));
    if (ess) HIP_TRY(hipMemcpy(h->rec.ess, ess, sizeof(double) * n_stages, hipMemcpyHostToDevice));
    if (c) HIP_TRY(hipMemcpy(h->rec.c, c, sizeof(double) * n_stages, hipMemcpyHostToDevice));
    if (accept) HIP_TRY(hipMemcpy(h->rec.accept, accept, sizeof(double) * n_stages, hipMemcpyHostToDevice));
    if (resampled) HIP_TRY(hipMemcpy(h->rec.resampled, resampled, sizeof(int) * n_stages, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int smcmi_set_history(smcmi_handle *h, int32_t n_stages, const double *w, const double *W) {
    if (!h || n_stages < 1 || n_stages > h->cfg.max_stages) return set_err(SMCMI_ERR_ARG, "bad argument");
    if (!h->cfg.store_history) return set_err(SMCMI_ERR_STATE, "history is not stored (store_history = 0)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t bytes = sizeof(double) * (size_t)h->n * n_stages;
    if (w) HIP_TRY(hipMemcpy(h->d_hist_w, w, bytes, hipMemcpyHostToDevice));
    if (W) HIP_TRY(hipMemcpy(h->d_hist_W, W, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

// ------------------------------------------------------------------------------------------------ shard-level pieces
extern "C" int smcmi_comm_buffer(smcmi_handle *h, double **dev_ptr, int64_t *capacity) {
    if (!h || !dev_ptr) return set_err(SMCMI_ERR_ARG, "null argument");
    *dev_ptr = h->d_comm;
    if (capacity) *capacity = h->comm_cap;
    return 0;
}

extern "C" int smcmi_comm_read(smcmi_handle *h, double *out, int64_t count) {
    if (!h || !out || count < 0 || count > h->comm_cap) return set_err(SMCMI_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipMemcpyAsync(out, h->d_comm, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int smcmi_shard_ess_partial(smcmi_handle *h, const double *phis, int32_t k, double phi_prev) {
    if (!h || !phis || k < 1 || k > KC) return set_err(SMCMI_ERR_ARG, "bad argument (1 <= k <= SMCMI_MAX_CAND)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    s.done = 0; s.phi_prev = phi_prev;
    s.sol[0].mode = MODE_SECTION; s.sol[0].n_valid = k;
    for (int q = 0; q < k; ++q) s.sol[0].cand[q] = phis[q];
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_pass<KC, false><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, nullptr, h->d_part_ess[0], h->nb_e, 0, nullptr, 0);
    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_ess[0], h->nb_e, 2 * KC, h->d_comm);   // comm[q] = Σv, comm[KC+q] = Σv²
    return 0;
}

extern "C" int smcmi_shard_correct_partial(smcmi_handle *h, double phi_n, double phi_prev, double prior_weight,
                                           double log_prob_old_data, int32_t stage_col) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    if (stage_col + 1 > h->cfg.max_stages) return set_err(SMCMI_ERR_CAPACITY, "max_stages exceeded");
    s.done = 0; s.phi_n = phi_n; s.phi_prev = phi_prev; s.rp.pw = prior_weight; s.rp.logp_old = log_prob_old_data;
    s.stage = stage_col + 1; s.rp.store_history = h->cfg.store_history;
    s.sol[0].mode = MODE_FINAL; s.sol[0].phi_n = phi_n;
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_pass<1, true><<<h->nb_e, TB, 0, h->stream>>>(h->cl, h->d_st, h->d_sched, nullptr, h->d_part_fin, h->nb_e, 0, h->d_hist_w, h->n);
    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_part_fin, h->nb_e, 2, h->d_comm);
    h->last_n_stages = std::max(h->last_n_stages, stage_col + 1);
    return 0;
}

extern "C" int smcmi_shard_normalize_moments_partial(smcmi_handle *h, double sum_unnorm, int32_t resampled, const double *shift,
                                                     int32_t stage_col) {
    if (!h) return set_err(SMCMI_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    s.done = 0; s.sumw = sum_unnorm; s.do_resample = 0; s.stage = stage_col + 1; s.rp.store_history = h->cfg.store_history;
    for (int a = 0; a < h->d; ++a) s.shift[a] = shift ? shift[a] : 0.0;
    if (resampled) s.sumw = (double)h->cfg.n_parts;   // after a gather the weights are already 1: (w*N)/N == w
    if (push_state(h)) return SMCMI_ERR_HIP;
    const int nbm = launch_moments(h, h->d_hist_W, 0);
    k_moments_reduce<<<(h->npairs + 63) / 64, 1024, 0, h->stream>>>(h->d_st, h->d_part_mom, nbm, h->npairs, h->d_comm, 0);
    return 0;
}

extern "C" int smcmi_shard_weights_device_ptr(smcmi_handle *h, double **dev_ptr) {
    if (!h || !dev_ptr) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (pull_state(h)) return SMCMI_ERR_HIP;
    *dev_ptr = h->cl.buf[h->h_st.cur] + (long long)(h->R - 1) * h->n;
    return 0;
}

// Selection on one shard of a sharded population: `dev_full_weights` (n_parts unnormalised weights, all-gathered) gives the
// global cumulative sum; this shard's output slots pick their ancestors (global ids) and copy the rows from the
// all-gathered cloud `dev_full_cloud` (n_parts x R column-major).
extern "C" int smcmi_shard_resample(smcmi_handle *h, const double *dev_full_weights, const double *dev_full_cloud, int32_t method,
                                    uint32_t stage, int64_t *ancestors_out) {
    if (!h || !dev_full_weights || !dev_full_cloud) return set_err(SMCMI_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const long long N = h->cfg.n_parts;
    if (!h->d_cum_full) {
        if (dmalloc(&h->d_cum_full, N)) return SMCMI_ERR_HIP;
        h->nb_full = (int)std::min<long long>(1024, std::max<long long>(1, (N + 511) / 512));
        if (dmalloc(&h->d_part_full, (size_t)h->nb_full * 2) || dmalloc(&h->d_off_full, h->nb_full)) return SMCMI_ERR_HIP;
    }
    CloudPtrs wcl{};                      // view the full weight vector as a 1-column "cloud"
    wcl.buf[0] = wcl.buf[1] = const_cast<double *>(dev_full_weights);
    wcl.n = N; wcl.R = 1;
    k_weight_chunk_sums<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_part_full);
    k_chunk_offsets<<<1, 1, 0, h->stream>>>(h->d_st, h->d_part_full, h->nb_full, h->d_off_full, 0.0, 1);
    k_scan_weights<<<h->nb_full, TB, 0, h->stream>>>(wcl, h->d_st, h->d_off_full, h->d_cum_full, 1, h->nb_full);
    k_resample_gather<<<(unsigned)((h->n + TB - 1) / TB), TB, 0, h->stream>>>(h->cl, h->d_st, h->d_cum_full, N, h->cfg.gid0, N, method,
                                                                            h->cfg.seed, stage, nullptr, h->d_anc, dev_full_cloud, 1);
    if (ancestors_out) HIP_TRY(hipMemcpyAsync(ancestors_out, h->d_anc, sizeof(long long) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int smcmi_shard_mutate_partial(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                                          const int32_t *blocks_free, int32_t n_blocks, double phi_n, double phi_prev, double c,
                                          double alpha, int32_t n_mh_steps, uint32_t stage) {
    (void)phi_prev;
    if (int rc = need_model(h, true)) return rc;
    if (!mu_free || !Sigma_free || !block_ptr || !blocks_free) return set_err(SMCMI_ERR_ARG, "null argument");
    if (pull_state(h)) return SMCMI_ERR_HIP;
    DevState &s = h->h_st;
    if (int rc = stage_blocks(h, mu_free, Sigma_free, block_ptr, blocks_free, n_blocks, c)) return rc;
    s.done = 0; s.err = 0; s.do_resample = 0;
    s.mut_c = c; s.mut_alpha = alpha; s.mut_phi = phi_n; s.mut_steps = n_mh_steps; s.mut_stage = stage;
    if (push_state(h)) return SMCMI_ERR_HIP;
    k_prepare_mutation<<<1, PT, h->prep_lds, h->stream>>>(h->d_st, h->d_model, h->d_totals, 0, h->cfg.seed, 0, 0, 1);
    const int nbl = launch_mutate(h, s.n_blocks, 1, alpha);
    k_reduce_partials<<<1, TB, 0, h->stream>>>(h->d_acc_part, nbl, 1, h->d_comm);
    if (pull_state(h)) return SMCMI_ERR_HIP;
    return err_from_state(s.err);
}

#include "callback.hpp"
extern "C" int smcmi_callback_phases(smcmi_handle *h, double *ms_out, int32_t n) {
    if (!h || !ms_out || n < 1) return set_err(SMCMI_ERR_ARG, "bad argument");
    for (int k = 0; k < n; ++k) ms_out[k] = (h->cbuf && k < CBP_N) ? h->cbuf->phase_ms[k] : 0.0;
    return 0;
}
#include "sharded.hpp"
#include "launch2.hpp"
#include "run2.hpp"
// Engine 3's persistent segments rely on every block of their grid being resident.  Residency is verified once per handle (k3_census); if
// the GPU is shared later (another process, CU masking, a second stream) a hand-over inside a segment can time out: the run is void and
// the cloud already overwritten.  A single-handle run therefore keeps what a repeat needs - the cloud it started from (one
// device-to-device copy of n x R doubles, ~10 µs at config 2) and the loop state - and repeats itself on engine 2's launches, which the
// time-out has made the handle's engine from then on.
// A group of handles (sharded segments, run2.hpp) does the same: a hand-over that ran out anywhere stops that rank's posts, so every rank's
// run ends in the time-out and every rank repeats - the same decision everywhere without an exchange.
static int run2_guarded(ShardGroup &g, const smcmi_run_config *rc, smcmi_result *res) {
    static const int e3_off = getenv("SMCMI_ENGINE3") ? (atoi(getenv("SMCMI_ENGINE3")) == 0) : 0;
    static const int e3_sharded = getenv("SMCMI_ENGINE3") ? (atoi(getenv("SMCMI_ENGINE3")) != 2) : 1;
    smcmi_handle *h0 = g.hs[0];
    const bool single = g.world == 1 && !g.rccl && g.hs.size() == 1;
    bool may_seg = !e3_off && h0->d <= 10 && (single || e3_sharded);
    for (auto *h : g.hs) may_seg = may_seg && h->n <= (single ? 253952 : 131072) && (!h->e2 || h->e2->e3_state >= 0);
    const long long state_n = (long long)((sizeof(DevState) + 7) / 8);     // (doubles)
    if (may_seg) {
        for (auto *h : g.hs) {
            const long long cloud_n = (long long)h->n * h->R;
            HIP_TRY(hipSetDevice(h->cfg.device));
            if (!h->d_snap && dmalloc(&h->d_snap, (size_t)(cloud_n + state_n))) return SMCMI_ERR_HIP;
            // (both copies stay on the device, in stream order: no host round trip at the start of a run)
            launch_copy_f64(h->d_snap, h->cl.buf[0], cloud_n, h->stream);
            HIP_TRY(hipMemcpyAsync(h->d_snap + cloud_n, h->d_st, sizeof(DevState), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    int e = run2_impl(g, rc, res);
    if (e == SMCMI_ERR_TIMEOUT && may_seg && h0->e2 && h0->e2->e3_state < 0) {
        if (getenv("SMCMI_TRACE")) fprintf(stderr, "[smcmi3] segment time-out: the run is repeated as launches from the cloud it started with\n");
        for (auto *h : g.hs) {
            const long long cloud_n = (long long)h->n * h->R;
            HIP_TRY(hipSetDevice(h->cfg.device));
            launch_copy_f64(h->cl.buf[0], h->d_snap, cloud_n, h->stream);
            HIP_TRY(hipMemcpyAsync(h->d_st, h->d_snap + cloud_n, sizeof(DevState), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            h->seg_timeouts += 1;
        }
        e = run2_impl(g, rc, res);
    }
    return e;
}
static int run2_single(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res) {
    ShardGroup g;
    g.hs = {h}; g.world = 1; g.rccl = false;
    return run2_guarded(g, rc, res);
}


// compute_proposal_densities (src/helpers.jl:128-164) of one move through the device's dense mixture form (kernels.hpp mix_densities):
// the reference's own fixture (test/helpers.jl:101-127) reaches the HIP code the alpha < 1 mutation kernels run
extern "C" int smcmi_debug_proposal_densities(const double *para_draw, const double *para_subset, const double *mu, const double *Sigma, int32_t d,
                                              double c, double alpha, double *q0, double *q1) {
    if (!para_draw || !para_subset || !mu || !Sigma || !q0 || !q1) return set_err(SMCMI_ERR_ARG, "null argument");
    if (d < 1 || d > 16) return set_err(SMCMI_ERR_ARG, "block length out of range (1..16)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_err(SMCMI_ERR_HIP, "no HIP device available: libsmcmi has no CPU fallback");
    double *dbuf = nullptr;
    const size_t nd = (size_t)3 * d + (size_t)d * d + 3;
    HIP_TRY(hipMalloc((void **)&dbuf, nd * sizeof(double)));
    std::vector<double> hb(nd);
    memcpy(&hb[0], para_draw, sizeof(double) * d); memcpy(&hb[d], para_subset, sizeof(double) * d); memcpy(&hb[2 * d], mu, sizeof(double) * d);
    memcpy(&hb[3 * d], Sigma, sizeof(double) * d * d);
    hipError_t e = hipMemcpy(dbuf, hb.data(), nd * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_debug_mix_densities<16><<<1, 256>>>(dbuf, dbuf + d, dbuf + 2 * d, dbuf + 3 * d, d, c, alpha, dbuf + 3 * d + (size_t)d * d);
        e = hipGetLastError();
    }
    double o[3] = {0.0, 0.0, 0.0};
    if (e == hipSuccess) e = hipMemcpy(o, dbuf + 3 * d + (size_t)d * d, sizeof(o), hipMemcpyDeviceToHost);
    hipFree(dbuf);
    if (e != hipSuccess) return set_err(SMCMI_ERR_HIP, std::string("smcmi_debug_proposal_densities: ") + hipGetErrorString(e));
    if (o[2] != 0.0) return err_from_state(SMCMI_ERR_POSDEF);
    *q0 = o[0]; *q1 = o[1];
    return 0;
}

// ---- peer mailbox across processes (include/smcmi.h): the caller may exchange the 64-byte table handles itself
extern "C" int smcmi_mailbox_export(smcmi_handle *h, uint8_t *handle_out) {
    if (!h || !handle_out) return set_err(SMCMI_ERR_ARG, "null argument");
    return mbox_export(h, handle_out);
}
extern "C" int smcmi_mailbox_import(smcmi_handle *h, int32_t rank, int32_t world, const uint8_t *all_handles) {
    if (!h || !all_handles) return set_err(SMCMI_ERR_ARG, "null argument");
    return mbox_import(h, world, rank, all_handles);
}
extern "C" int smcmi_mailbox_selftest(smcmi_handle *h, int32_t rank, int32_t world, int32_t rounds, int32_t *errors_out) {
    if (!h || !errors_out || rounds < 1) return set_err(SMCMI_ERR_ARG, "bad argument");
    int e = 0;
    if (int rc = mbox_selftest(h, world, rank, rounds, &e)) return rc;
    *errors_out = e;
    return 0;
}
extern "C" int smcmi_mailbox_active(smcmi_handle *h, int32_t *active_out) {
    if (!h || !active_out) return set_err(SMCMI_ERR_ARG, "null argument");
    *active_out = h->mbox_used ? 1 : 0;
    return 0;
}
