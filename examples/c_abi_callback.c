/* The reference's real entry point from plain C: smc(loglikelihood::Function, parameters, data; ...) (src/smc_main.jl:118) with the
 * user's likelihood as a C function pointer (smcmi_set_likelihood_callback, include/smcmi.h) - what a Julia `@cfunction` trampoline
 * hands over.  Config 2's workload (10-dim isotropic Gaussian, adaptive tempering) twice on the same Philox seed:
 * once with the built-in device family, once with the callback; the runs must agree (stage / resample counts, log-MDD to 1e-9),
 * and the callback path's throughput is printed.
 *
 *   gcc -std=c99 -O2 -ffp-contract=off -fopenmp -Iinclude examples/c_abi_callback.c -Lsmc.jl_amd/csrc -lsmcmi -lm -o c_abi_callback
 *   (-DCB_THREADS=<k>: threads the callback shares a chunk's rows among, default 4; without -fopenmp: one)
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
/* (column by column: every column of the block is one unit-stride stream - the block has just crossed PCIe, so the rate of this
 * function is the rate the host reads it from memory; the sum over j keeps its order, so the values are those of the row-wise loop.
 * Built with -fopenmp the rows are shared among CB_THREADS threads - the reference evaluates its likelihood on all workers too,
 * src/smc_main.jl:472-476 - while the library still invokes the function on the calling thread only.) */
#ifndef CB_THREADS
#define CB_THREADS 4
#endif
#ifdef _OPENMP
#define CB_THREADS_USED CB_THREADS
#else
#define CB_THREADS_USED 1
#endif
static int gauss_loglik(const double *theta, int64_t m, int64_t d, double *out, void *ud) {
    gauss_data *g = (gauss_data *)ud;
    const double c0 = g->c0;
    const int64_t tile = 256;                       /* (a chunk of 12 500 rows = 49 tiles: enough for every thread) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(CB_THREADS)
#endif
    for (int64_t k0 = 0; k0 < m; k0 += tile) {
        const int64_t k1 = k0 + tile < m ? k0 + tile : m;
        double acc[256];
        for (int64_t k = k0; k < k1; ++k) acc[k - k0] = 0.0;
        for (int64_t j = 0; j < d; ++j) {
            const double *col = theta + m * j, mu = g->mean[j];
            for (int64_t k = k0; k < k1; ++k) { const double e = col[k] - mu; acc[k - k0] += e * e; }
        }
        for (int64_t k = k0; k < k1; ++k) out[k] = c0 - acc[k - k0] / (2.0 * g->sigma * g->sigma);
    }
    g->calls += 1;
    return 0;
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 100000;
    smcmi_result res[2];
    double secs[2], phases[8] = {0};
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
        if (mode == 1) CHECK(smcmi_callback_phases(h, phases, 8));
        CHECK(smcmi_destroy(h));
    }
    const double ps0 = (double)n * (res[0].n_stages - 1) / secs[0], ps1 = (double)n * (res[1].n_stages - 1) / secs[1];
    const double st = (double)(res[1].n_stages - 1);
    printf("{\"n_parts\": %lld, \"device\": {\"n_stages\": %d, \"resamples\": %d, \"logmdd\": %.17g, \"particle_stages_per_s\": %.4g}, "
           "\"callback\": {\"n_stages\": %d, \"resamples\": %d, \"logmdd\": %.17g, \"particle_stages_per_s\": %.4g, \"calls\": %lld, \"callback_threads\": %d, "
           "\"ms_per_stage\": %.4f, \"phases_ms_per_stage\": {\"first_chunk_wait\": %.4f, \"later_chunk_wait\": %.4f, \"pack\": %.4f, \"callback\": %.4f, "
           "\"scatter\": %.4f, \"enqueue\": %.4f, \"stage_device_part\": %.4f}}}\n",
           n, res[0].n_stages, res[0].resamples, res[0].logmdd, ps0, res[1].n_stages, res[1].resamples, res[1].logmdd, ps1, g.calls, CB_THREADS_USED,
           1e3 * secs[1] / st, phases[0] / st, phases[1] / st, phases[2] / st, phases[3] / st, phases[4] / st, phases[5] / st, phases[6] / st);
    /* (one invocation per chunk of a stage's batch: include/smcmi.h) */
    if (res[0].n_stages != res[1].n_stages || res[0].resamples != res[1].resamples || fabs(res[0].logmdd - res[1].logmdd) > 1e-9 ||
        g.calls < res[1].n_stages - 1 || g.calls % (res[1].n_stages - 1) != 0) {
        printf("MISMATCH\n");
        return 2;
    }
    printf("OK\n");
    return 0;
}
