/* The drop-in boundary used from plain C (no Python, no torch): config 2 of BASELINE.json - 10-dim isotropic Gaussian likelihood,
 * N(0, 5) priors, adaptive tempering - through include/smcmi.h only.  This is the call sequence a Julia `ccall` shim issues
 * (INTEGRATION.md): create -> set_parameters / set_likelihood -> init_from_prior -> run -> records / cloud.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_config2.c -Lsmc.jl_amd/csrc -lsmcmi -lm -o c_abi_config2
 *   LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./c_abi_config2 [n_parts] [seed]
 */
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

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 100000;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
    smcmi_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_parts = n; cfg.n_local = n; cfg.gid0 = 0; cfg.n_para = D; cfg.device = 0; cfg.seed = seed;
    cfg.max_stages = 1500; cfg.store_history = 0;
    smcmi_handle *h = NULL;
    CHECK(smcmi_create(&cfg, &h));

    /* parameters: all free, wide closed bounds, Normal(0, 5) priors (tests/models.py gauss_spec) */
    int32_t fixed[D], fam[D];
    double lo[D], hi[D], pa[D], pb[D], mean[D];
    for (int k = 0; k < D; ++k) {
        fixed[k] = 0; fam[k] = SMCMI_PRIOR_NORMAL; lo[k] = -1e5; hi[k] = 1e5; pa[k] = 0.0; pb[k] = 5.0;
        mean[k] = -1.0 + 2.0 * (double)k / (double)(D - 1);      /* the "data": the likelihood's mean vector m_j */
    }
    CHECK(smcmi_set_parameters(h, fixed, lo, hi, fam, pa, pb));
    const double sigma = 0.25;
    CHECK(smcmi_set_likelihood(h, SMCMI_WHICH_NEW, SMCMI_LIK_GAUSS_ISO, &sigma, 1, mean, D, 1, NULL, 0, 0));
    CHECK(smcmi_set_likelihood(h, SMCMI_WHICH_OLD, SMCMI_LIK_NONE, NULL, 0, NULL, 0, 0, NULL, 0, 0));
    CHECK(smcmi_init_from_prior(h));

    smcmi_run_config rc;
    memset(&rc, 0, sizeof rc);
    rc.n_blocks = 1; rc.n_mh_steps = 1; rc.lambda = 2.1; rc.n_phi = 300; rc.resampling_method = SMCMI_RESAMPLE_SYSTEMATIC;
    rc.threshold_ratio = 0.5; rc.c = 0.5; rc.alpha = 1.0; rc.target = 0.25; rc.use_fixed_schedule = 0; rc.tempering_target = 0.97;
    smcmi_result res;
    CHECK(smcmi_run(h, &rc, &res));

    double *phi = malloc(sizeof(double) * (size_t)res.n_stages), *ess = malloc(sizeof(double) * (size_t)res.n_stages);
    CHECK(smcmi_get_stage_records(h, phi, ess, NULL, NULL, NULL));
    double *P = malloc(sizeof(double) * (size_t)n * (D + 5));
    CHECK(smcmi_download_cloud(h, P));
    double mu0 = 0.0, sw = 0.0;                      /* weighted posterior mean of the first parameter */
    for (long long i = 0; i < n; ++i) { mu0 += P[(D + 4) * n + i] * P[i]; sw += P[(D + 4) * n + i]; }
    printf("{\"n_parts\": %lld, \"n_stages\": %d, \"resamples\": %d, \"logmdd\": %.17g, \"phi_last\": %.17g, \"ess_last\": %.17g, "
           "\"mean0\": %.17g, \"seconds\": %.6f}\n",
           n, res.n_stages, res.resamples, res.logmdd, phi[res.n_stages - 1], ess[res.n_stages - 1], mu0 / sw, res.seconds);
    free(phi); free(ess); free(P);
    CHECK(smcmi_destroy(h));
    return 0;
}
