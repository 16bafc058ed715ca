// devstate.hpp - device-resident state of one engine handle.
//
// Everything a stage needs lives in HBM so that the stage is a fixed kernel sequence with no host
// decisions: tempering scalars, the ϕ-solver bracket, resample decision, proposal factors, per-stage
// records.  Field comments cite the reference variable they replace (src/smc_main.jl unless noted).
#pragma once
#include <stdint.h>

#include "../../include/smcmi.h"

namespace smcmi {

constexpr int MAXD = SMCMI_MAX_PARA;
constexpr int KC = SMCMI_MAX_CAND;          // candidates per ESS pass
constexpr int NPAIR_MAX = (MAXD + 1) * (MAXD + 2) / 2;
constexpr int LIK_PAR_MAX = 16;
constexpr int ES = 32;                       // doubles per partial row of the mutation epilogue (256 B: two rows per wave-load)
constexpr int EKA = 16;                      // energy power sums a_0..a_15 = Σ W p^k          (slots 0..15)
constexpr int EKB = 15;                      //                   b_0..b_14 = Σ W² p^k         (slots 16..30)
constexpr int EKU = 31;                      // after a resample (all W = 1, b = a): a_0..a_30 (slots 0..30)
constexpr int EACC = 31;                     // slot 31: the block's acceptance sum

struct LikDev {
    int family;
    int n_par;
    double par[LIK_PAR_MAX];
    double c0;                 // family constant precomputed on the host
    const double *data;        // column-major rows x cols
    long long rows, cols;
    const double *aux;
    long long aux_rows, aux_cols;
};

struct ModelDev {
    int d, n_free;
    int has_other_priors;      // some free parameter has a prior family other than Normal / Uniform
    int lik_prefix;            // > 0: lik[1] is the lgss_kalman family on the first lik_prefix columns of lik[0]'s data (same structure
                               // block): one filter pass yields both log-likelihoods (model.hpp kalman_lgss2)
    int fixed[MAXD];
    int free_inds[MAXD];
    double lo[MAXD], hi[MAXD];
    int prior_family[MAXD];
    double prior_a[MAXD], prior_b[MAXD];
    double prior_k[MAXD];      // family constant precomputed on the host (e.g. log σ, -log(b-a))
    LikDev lik[2];             // [0] loglikelihood/data, [1] old_loglikelihood/old_data
};

enum SolveMode { MODE_IDLE = 0, MODE_SCAN = 1, MODE_SECTION = 2, MODE_FINAL = 3 };

// Default relative resolution of the adaptive-ϕ root (Roots.fzero's xtol = 0 would be adjacent floats; per-stage errors of the
// step carry forward through ϕ_n = ϕ_{n-1} + δ, so the default sits two decades above the rounding noise of the ESS sums).
constexpr double DEFAULT_PHI_RTOL = 1e-12;
// A predicted root (predict-correct-verify stages) is accepted when the ESS it produced puts it within this relative distance of
// the true root.  Well-conditioned steps verify at ~1e-14; long steps (tempering_target 0.9) land around 1e-11, and rejecting
// those costs a host round trip per stage for nothing the 1e-3 log-MDD contract could see.
constexpr double SPEC_VERIFY_RTOL = 1e-10;

struct RunParams {             // smc() kwargs, uploaded once per run
    long long n_parts;         // global N
    int n_blocks, n_mh_steps;
    int n_phi;
    int resampling_method;
    int use_fixed_schedule;
    int max_stages;
    int store_history;
    int stall_on_exhaust;      // out of solver passes: 1 = stall the run (done = 2) for the host to resume, 0 = accept the bracket
    int stop_stage;            // > 0: pause (done = 5) once stage_index has reached it (intermediate saves, smc_main.jl:499-507)
    int shift_lag;             // engines 2 / 3, fixed schedules: the energy shift of a stage's incremental weights is the cloud's largest energy
                               // one mutation earlier than the latest (stage2.hpp Begin2::e_seen) - a stage then needs ONE chip-wide hand-over
    double threshold;          // threshold_ratio * n_parts (:203)
    double alpha, target;
    double tempering_target;
    double pw, logp_old;       // tempered_update_prior_weight, log_prob_old_data
    double phi_rtol;           // bracket width (relative) at which the adaptive-ϕ root is accepted
};

// Adaptive-ϕ solver state (src/helpers.jl:9-56).  Two copies are kept (ping-pong by pass parity): every block of
// pass p reads copy (p-1)&1, all blocks recompute the same decision, block 0 alone writes copy p&1.
struct Solver {
    int mode;
    int n_valid;               // valid entries of cand[]
    int j;                     // 1-based index into the proposed fixed schedule (:129)
    int unconverged;           // passes ran out before the bracket reached phi_rtol (diagnostic)
    double phi_prop, ess_bar;
    double lo, hi, glo, ghi;   // bracket with g(lo) >= 0 > g(hi), g = ESS(ϕ) - ESS_bar
    double phi_n;
    double phi0;               // ϕ_{n-1}, the start of the tempering step being solved
    double gprime;             // dESS/dϕ at the predicted root (Taylor model): turns an ESS mismatch into a ϕ error (spec stages)
    int spec;                  // 1: ϕ_n is the PREDICTED root, taken without a certificate pass; k_prepare_mutation verifies it
    int pad2_;
    double cand[KC];
    int cj[KC];                // SCAN: walk step of cand[k] (0 = ϕ_prop, q = schedule[j+q-1]); -1 = predictor ring point
};

struct DevState {
    RunParams rp;
    // ---- loop scalars
    int stage;                 // i == cloud.stage_index
    int j;                     // 1-based index into the proposed fixed schedule
    int resampled_last;        // resampled_last_period
    int do_resample;           // this stage's selection decision (:435)
    int done;                  // ϕ_n reached 1 (or error)
    int err;
    int cur;                   // ping-pong cloud buffer holding the current particles
    int resamples;             // cloud.resamples
    double phi_prev, phi_n, phi_prop;
    double ess_prev;           // cloud.ESS[i-1]
    double ess;
    double sumw, sumw2;        // Σ W̃, Σ W̃² at ϕ_n (unnormalised)
    double logz;               // running log-MDD
    double c, accept;          // cloud.c, cloud.accept
    long long solver_passes;   // diagnostic: number of particle passes spent in the adaptive-ϕ solver
    double e_center;           // centre of the energy power sums the mutation epilogue accumulates (predictor, kernels.hpp)
    double pred_delta;         // diagnostic: predicted ϕ_n - ϕ_{n-1} of the current stage (NaN: no prediction)
    double e_shift;            // energy shift of this stage's incremental weights (largest loglh - old_loglh of the cloud; 0 = none)
    int skip_fold;             // continued run: the acceptance / energy sums of the last mutation are already folded (or were never here)
    int pad2_;
    double e_seen;             // engines 2 / 3: Post2::e_seen across a pause (NaN: none)
    Solver sol[2];
    // ---- moments / proposal (smc_main.jl:457-469, mutation.jl:81)
    double shift[MAXD];        // centering used by the one-pass moment kernel (previous mean)
    double mean[MAXD];         // θ_bar
    double cov[MAXD * MAXD];   // R (row-major d x d)
    int n_blocks;
    int max_db;                // largest block length
    int block_ptr[MAXD + 1];
    int blocks_free[MAXD];     // positions in the free-parameter list, block order
    int blocks_all[MAXD];      // parameter indices, block order
    int l_off[MAXD];           // offset of block b's factor in L
    double mu_b[MAXD];         // θ_bar_fr in block order
    double L[MAXD * MAXD];     // chol(c² Σ_b), row-major per block
    double sd_draw[MAXD];      // sqrt(c² Σ_ii)   (helpers.jl:94)
    double sd_dens[MAXD];      // sqrt(Σ_ii)      (helpers.jl:146, quirk Q1)
    double logdet[MAXD];       // log det(c² Σ_b)
    double mut_c, mut_alpha, mut_phi;
    int mut_steps;
    unsigned mut_stage;
};

// per-stage records (cloud.tempering_schedule, cloud.ESS, c, accept, resample flag), length max_stages
struct Records {
    double *phi, *ess, *c, *accept;
    int *resampled;
};

}  // namespace smcmi
