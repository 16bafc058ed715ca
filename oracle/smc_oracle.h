/*
 * smc_oracle.h - CPU ORACLE for the SMC correction/selection/mutation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library
 * (smc.jl_amd/csrc -> libsmcmi.so); it never links, imports or calls anything in oracle/.
 *
 * It restates, in plain FP64 C, the algorithm of FRBNY-DSGE/SMC.jl v0.1.15 (pure Julia; Julia is
 * not installed here and its dependencies are not vendored, so the reference cannot be built or
 * run: there is no oracle/_ref).  Each function cites the reference file:line it follows.
 *
 * Pinning: RNG-free functions are checked against the reference's own golden fixtures
 * (tests/golden/*.npz, extracted by tests/golden/make_fixtures.py).  RNG-dependent functions
 * (mixture draw, block shuffle, resampling offsets, prior draws) are "parity unpinned" at the
 * stream level - the reference's goldens are tied to Julia's MersenneTwister - and use the
 * Philox4x32-10 contract documented in DESIGN.md, which the HIP path implements independently.
 *
 * Layout: a cloud is the reference's `cloud.particles` matrix, Julia column-major N x R with
 * R = n_para + 5: element (i, col) lives at p[col * N + i]  (src/particle.jl:31-63).
 * Columns: 0..d-1 parameters | d loglh | d+1 logprior | d+2 old_loglh | d+3 accept | d+4 weight.
 */
#ifndef SMC_ORACLE_H
#define SMC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* prior families (ModelConstructors / Distributions; dependency absent from the tree) */
enum { ORC_PRIOR_NORMAL = 0, ORC_PRIOR_UNIFORM = 1, ORC_PRIOR_GAMMA = 2, ORC_PRIOR_BETA = 3,
       ORC_PRIOR_INVGAMMA = 4, ORC_PRIOR_ROOTINVGAMMA = 5 };
/* built-in likelihood families */
enum { ORC_LIK_GAUSS_ISO = 0, ORC_LIK_LINREG = 1, ORC_LIK_LINMODEL3 = 2, ORC_LIK_CAPM_LITERAL = 3, ORC_LIK_LGSS_KALMAN = 4,
       ORC_LIK_NONE = -1 };
enum { ORC_RESAMPLE_SYSTEMATIC = 0, ORC_RESAMPLE_MULTINOMIAL = 1 };

typedef struct {
    int32_t family;          /* ORC_LIK_* */
    const double *par;       /* family parameters */
    int64_t n_par;
    const double *data;      /* column-major rows x cols (Julia layout) */
    int64_t rows, cols;
    const double *aux;       /* regressors etc., column-major */
    int64_t aux_rows, aux_cols;
} orc_lik;

typedef struct {
    int32_t n_para;
    const int32_t *fixed;        /* [d] 1 = fixed */
    const double *lo, *hi;       /* [d] valuebounds, closed interval */
    const int32_t *prior_family; /* [d] */
    const double *prior_a, *prior_b; /* [d] */
    orc_lik lik;                 /* loglikelihood(parameters, data) */
    orc_lik old_lik;             /* old_loglikelihood(parameters, old_data); family NONE => old_data empty */
} orc_model;

typedef struct {
    int64_t n_parts;
    int32_t n_blocks, n_mh_steps;
    double lambda;               /* λ */
    int32_t n_phi;               /* n_Φ */
    int32_t resampling_method;
    double threshold_ratio;
    double c, alpha, target;
    int32_t use_fixed_schedule;
    double tempering_target;
    double prior_weight;         /* tempered_update_prior_weight */
    double log_prob_old_data;
    uint64_t seed;
    int32_t max_stages;          /* capacity of per-stage outputs (incl. stage 1) */
    int32_t n_threads;           /* OpenMP threads for the mutation loop (results do not depend on it) */
    double initial_ess;          /* cloud.ESS[1] when continuing from an old cloud (tempered update); 0 => n_parts */
    int32_t variant;             /* CPU-baseline variants (bench.py; results of 0 and 1 are bit-identical):
                                    0 = default: block factors hoisted out of the particle loop, serial ESS evaluations;
                                    1 = "reference-faithful" cost model: MvNormal(...) re-factorised for every particle
                                        (mutation.jl:81), serial bisection (helpers.jl:49), mutation over n_threads workers;
                                    2 = "optimised OpenMP": hoisted factors + every ESS evaluation reduced over n_threads
                                        (summation order differs => phi_n agrees to rounding only) */
    int32_t pad_;
} orc_run_config;

typedef struct {
    int32_t n_stages;            /* final cloud.stage_index (== number of schedule entries) */
    int32_t resamples;
    double logmdd;
    double c, accept;
    double seconds;              /* sum of per-stage wall time, reference bracket smc_main.jl:378,489 */
} orc_run_result;

/* --- RNG contract (DESIGN.md "RNG contract") --- */
void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);
void orc_uniform_pair(uint64_t seed, uint64_t pid, uint32_t stage, uint32_t tag, double *ua, double *ub);

/* --- helpers.jl --- */
double orc_compute_ess(const double *loglh, const double *w, const double *old_loglh, int64_t n,
                       double phi_n, double phi_n1);
int orc_solve_adaptive_phi(const double *particles, int64_t n, int32_t R, double ess_prev,
                           const double *sched, int32_t n_phi, int32_t *j, double *phi_prop, double phi_n1,
                           double target, int32_t *resampled_last, double *phi_n, int32_t *n_evals);
void orc_proposal_densities(const double *para_draw, const double *para_subset, const double *mu,
                            const double *Sigma, int32_t db, double c, double alpha, double *q0, double *q1);
int orc_mixture_draw(const double *theta_old, const double *mu, const double *Sigma, int32_t db, double c,
                     double alpha, uint64_t seed, uint64_t pid, uint32_t stage, uint32_t t, double *theta_new);
void orc_generate_blocks(int32_t n_free, int32_t n_blocks, const int32_t *free_inds, uint64_t seed,
                         uint32_t stage, int32_t *blocks_free, int32_t *blocks_all, int32_t *block_ptr);

/* --- particle.jl / smc_main.jl stage pieces --- */
void orc_correct(double *particles, int64_t n, int32_t R, double phi_n, double phi_n1, double pw,
                 double logp_old, double *inc_w, double *norm_w, double *ess, double *sum_unnorm);
void orc_resample(const double *weights, int64_t nw, int64_t n_parts, int32_t method, uint64_t seed,
                  uint32_t stage, int64_t *idx);
void orc_resample_with_offsets(const double *weights, int64_t nw, int64_t n_parts, int32_t method,
                               const double *offsets, int64_t *idx);
void orc_weighted_mean(const double *particles, int64_t n, int32_t R, double *mean);
void orc_weighted_cov(const double *particles, int64_t n, int32_t R, double *cov);
double orc_update_c(double c, double accept, double target);

/* --- model pieces (ModelConstructors prior/update!, user likelihoods) --- */
double orc_logprior(const orc_model *m, const double *theta);
int orc_in_bounds(const orc_model *m, const double *theta);
double orc_loglik(const orc_lik *l, const double *theta, int32_t d);

/* --- mutation.jl --- */
int orc_mutation(const orc_model *m, double *p /* length R, strided */, int64_t stride, const double *mu_free,
                 const double *Sigma_free, int32_t n_free, const int32_t *blocks_free,
                 const int32_t *blocks_all, const int32_t *block_ptr, int32_t n_blocks, double phi_n,
                 double phi_n1, double c, double alpha, int32_t n_mh_steps, uint64_t seed, uint64_t pid,
                 uint32_t stage);
int orc_mutate_cloud(const orc_model *m, double *particles, int64_t n, int64_t pid0, const double *mu_free,
                     const double *Sigma_free, int32_t n_free, const int32_t *blocks_free,
                     const int32_t *blocks_all, const int32_t *block_ptr, int32_t n_blocks, double phi_n,
                     double phi_n1, double c, double alpha, int32_t n_mh_steps, uint64_t seed, uint32_t stage,
                     int32_t n_threads);

/* --- initialization.jl --- */
int orc_initial_draw(const orc_model *m, double *particles, int64_t n, int64_t pid0, uint64_t seed);

/* --- smc_main.jl:377-508 whole loop --- */
void orc_initialize_likelihoods(const orc_model *m, double *particles, int64_t n);
int orc_smc_run(const orc_model *m, const orc_run_config *cfg, double *particles, double *sched_out,
                double *ess_out, double *c_out, double *accept_out, int32_t *resampled_out,
                double *w_hist, double *W_hist, orc_run_result *res);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
