/* The reference's real entry point from plain C: smc(loglikelihood::Function, parameters, data; ...) (src/smc_main.jl:118) with the
 * user's likelihood as a C function pointer (smcmi_set_likelihood_callback, include/smcmi.h) - what a Julia `@cfunction` trampoline
 * hands over.  Config 2's workload (10-dim isotropic Gaussian, adaptive tempering) twice on the same Philox seed:
 * once with the built-in device family, once with the callback; the runs must agree (stage / resample counts, log-MDD to 1e-9),
 * and the callback path's throughput is printed.
 *
 *   gcc -std=c99 -O2 -ffp-contract=off -Iinclude examples/c_abi_callback.c -Lsmc.jl_amd/csrc -lsmcmi -lm -o c_abi_callback
 */
#define _USE_MATH_DEFINES
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smcmi.h"

#define D 10
#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, smcmi_last_error()); return 1; } \
    } while (0)

typedef struct { double mean[D], sigma, c0; long long calls; } gauss_data;

/* loglikelihood(parameters, data) for a batch: theta is m x d column-major (proposal k = theta[k + m * j]) */
static int gauss_loglik(const double *theta, int64_t m, int64_t d, double *out, void *ud) {
    gauss_data *g = (gauss_data *)ud;
    for (int64_t k = 0; k < m; ++k) {
        double acc = 0.0;
        for (int64_t j = 0; j < d; ++j) { const double e = theta[k + m * j] - g->mean[j]; acc += e * e; }
        out[k] = g->c0 - acc / (2.0 * g->sigma * g->sigma);
    }
    g->calls += 1;
    return 0;
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 100000;
    smcmi_result res[2];
    double secs[2];
    gauss_data g;
    memset(&g, 0, sizeof g);
    g.sigma = 0.25;
    for (int k = 0; k < D; ++k) g.mean[k] = -1.0 + 2.0 * (double)k / (double)(D - 1);
    g.c0 = -0.5 * (double)D * log(2.0 * M_PI * g.sigma * g.sigma);
    for (int mode = 0; mode < 2; ++mode) {
        smcmi_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.n_parts = n; cfg.n_local = n; cfg.n_para = D; cfg.seed = 1; cfg.max_stages = 1500; cfg.store_history = 0;
        smcmi_handle *h = NULL;
        CHECK(smcmi_create(&cfg, &h));
        int32_t fixed[D], fam[D];
        double lo[D], hi[D], pa[D], pb[D];
        for (int k = 0; k < D; ++k) { fixed[k] = 0; fam[k] = SMCMI_PRIOR_NORMAL; lo[k] = -1e5; hi[k] = 1e5; pa[k] = 0.0; pb[k] = 5.0; }
        CHECK(smcmi_set_parameters(h, fixed, lo, hi, fam, pa, pb));
        /* the initial draw (initial_draw!, initialization.jl:88-119) by the device family in both modes: the same starting cloud */
        CHECK(smcmi_set_likelihood(h, SMCMI_WHICH_NEW, SMCMI_LIK_GAUSS_ISO, &g.sigma, 1, g.mean, D, 1, NULL, 0, 0));
        CHECK(smcmi_set_likelihood(h, SMCMI_WHICH_OLD, SMCMI_LIK_NONE, NULL, 0, NULL, 0, 0, NULL, 0, 0));
        CHECK(smcmi_init_from_prior(h));
        if (mode == 1) CHECK(smcmi_set_likelihood_callback(h, SMCMI_WHICH_NEW, gauss_loglik, &g));
        smcmi_run_config rc;
        memset(&rc, 0, sizeof rc);
        rc.n_blocks = 1; rc.n_mh_steps = 1; rc.lambda = 2.1; rc.n_phi = 300; rc.resampling_method = SMCMI_RESAMPLE_SYSTEMATIC;
        rc.threshold_ratio = 0.5; rc.c = 0.5; rc.alpha = 1.0; rc.target = 0.25; rc.use_fixed_schedule = 0; rc.tempering_target = 0.97;
        CHECK(smcmi_run(h, &rc, &res[mode]));
        secs[mode] = res[mode].seconds;
        CHECK(smcmi_destroy(h));
    }
    const double ps0 = (double)n * (res[0].n_stages - 1) / secs[0], ps1 = (double)n * (res[1].n_stages - 1) / secs[1];
    printf("{\"n_parts\": %lld, \"device\": {\"n_stages\": %d, \"resamples\": %d, \"logmdd\": %.17g, \"particle_stages_per_s\": %.4g}, "
           "\"callback\": {\"n_stages\": %d, \"resamples\": %d, \"logmdd\": %.17g, \"particle_stages_per_s\": %.4g, \"calls\": %lld}}\n",
           n, res[0].n_stages, res[0].resamples, res[0].logmdd, ps0, res[1].n_stages, res[1].resamples, res[1].logmdd, ps1, g.calls);
    if (res[0].n_stages != res[1].n_stages || res[0].resamples != res[1].resamples || fabs(res[0].logmdd - res[1].logmdd) > 1e-9 ||
        g.calls != res[1].n_stages - 1) {
        printf("MISMATCH\n");
        return 2;
    }
    printf("OK\n");
    return 0;
}
