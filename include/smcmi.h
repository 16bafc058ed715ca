/*
 * smcmi.h - C ABI of libsmcmi.so, the MI355X-native SMC particle engine.
 *
 * Drop-in boundary for the correction / selection / mutation loop of FRBNY-DSGE/SMC.jl
 * (src/smc_main.jl:377-508 and everything it calls).  The reference is pure Julia and has no FFI
 * of its own; the entry points below are what a Julia `ccall` shim binds (see INTEGRATION.md and
 * smc.jl_amd/julia/SMCMI.jl) to keep `smc(loglikelihood, parameters, data; ...)` and the `Cloud`
 * accessors unchanged.  Each function cites the reference interface it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative SMCMI_ERR_* otherwise; text via smcmi_last_error().
 *  - no exceptions cross the ABI; all pointers are plain host pointers unless named `dev_*`.
 *  - a cloud is the reference's `cloud.particles`: Float64, Julia column-major N x R, R = n_para + 5,
 *    element (i, col) at p[col * N + i]; columns 0..d-1 parameters | d loglh | d+1 logprior |
 *    d+2 old_loglh | d+3 accept | d+4 weight           (src/particle.jl:31-63).  That layout is
 *    already struct-of-arrays, so upload/download are single contiguous copies.
 *  - indices returned to the caller are 0-based.
 *  - a handle owns its device memory and one HIP stream; calls on one handle are not re-entrant.
 *  - one handle = one shard: `n_local` particles starting at global id `gid0` out of `n_parts`.
 *    Single-GPU use has n_local == n_parts, gid0 == 0.
 */
#ifndef SMCMI_H
#define SMCMI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMCMI_MAX_PARA 64      /* n_para (incl. fixed) supported by the device kernels */
#define SMCMI_MAX_CAND 16      /* tempering candidates evaluated per ESS pass */

enum { SMCMI_OK = 0, SMCMI_ERR_ARG = -1, SMCMI_ERR_HIP = -2, SMCMI_ERR_NAN_ESS = -3, SMCMI_ERR_POSDEF = -4,
       SMCMI_ERR_CAPACITY = -5, SMCMI_ERR_BRACKET = -6, SMCMI_ERR_UNSUPPORTED = -7, SMCMI_ERR_STATE = -8,
       SMCMI_ERR_CALLBACK = -9, SMCMI_ERR_TIMEOUT = -10 };

/* prior families: Distributions.jl / ModelConstructors priors reachable from `prior(parameters)` (src/mutation.jl:95) */
enum { SMCMI_PRIOR_NORMAL = 0, SMCMI_PRIOR_UNIFORM = 1, SMCMI_PRIOR_GAMMA = 2, SMCMI_PRIOR_BETA = 3,
       SMCMI_PRIOR_INVGAMMA = 4, SMCMI_PRIOR_ROOTINVGAMMA = 5 };
/* device likelihood families standing in for the user callback `loglikelihood(parameters, data)` (src/mutation.jl:96) */
enum { SMCMI_LIK_NONE = -1, SMCMI_LIK_GAUSS_ISO = 0, SMCMI_LIK_LINREG = 1, SMCMI_LIK_LINMODEL3 = 2,
       SMCMI_LIK_CAPM_LITERAL = 3, SMCMI_LIK_LGSS_KALMAN = 4, SMCMI_LIK_HOST_CALLBACK = 100 };
/* src/resample.jl:23 `method`; :polyalgo is served by the multinomial kernel (same distribution) */
enum { SMCMI_RESAMPLE_SYSTEMATIC = 0, SMCMI_RESAMPLE_MULTINOMIAL = 1 };
enum { SMCMI_WHICH_NEW = 0, SMCMI_WHICH_OLD = 1 };

typedef struct smcmi_handle smcmi_handle;

typedef struct {
    int64_t n_parts;        /* global N                                    smc kwarg n_parts (smc_main.jl:123) */
    int64_t n_local;        /* particles held by this handle (0 => n_parts) */
    int64_t gid0;           /* global id of the first local particle */
    int32_t n_para;         /* d = length(parameters) (+ regime columns)   smc_main.jl:207-216 */
    int32_t device;         /* HIP device ordinal */
    uint64_t seed;          /* Philox key (replaces Random.seed!) */
    int32_t max_stages;     /* capacity of per-stage records / history columns (incl. stage 1) */
    int32_t store_history;  /* keep the N x n_stages w / W matrices (smc_main.jl:363-366,419-420) */
} smcmi_config;

/* kwargs of smc() that shape the loop (smc_main.jl:119-161) */
typedef struct {
    int32_t n_blocks, n_mh_steps;          /* :125-126 */
    double lambda;                         /* λ :128 */
    int32_t n_phi;                         /* n_Φ :129 */
    int32_t resampling_method;             /* :131 */
    double threshold_ratio;                /* :132 */
    double c, alpha, target;               /* :135-137 */
    int32_t use_fixed_schedule;            /* :139 */
    double tempering_target;               /* :140 */
    double tempered_update_prior_weight;   /* :156 */
    double log_prob_old_data;              /* :161 */
    int32_t solver_passes;                 /* kernel passes enqueued for the adaptive-ϕ solver per stage (0 => default 1); a stage that needs more is resumed */
    int32_t sync_every;                    /* adaptive schedule: host checks the done flag every k stages (0 => default) */
    int32_t use_graph;                     /* 2: HIP events around the mutation kernel / the segment launches (smcmi_result::kernel_ms_*); 0 otherwise.
                                              (1 replayed the stage as a hipGraph until round 5: no faster than direct launches, retired - taken as 0) */
    double initial_ess;                    /* cloud.ESS[1] for a tempered update started from an old cloud (0 => n_parts; initialization.jl:199-200) */
    double phi_rtol;                       /* relative bracket width accepted as the adaptive-ϕ root on stages that run certificate
                                              passes (0 => 1e-12; <0 => adjacent floats, and no stage is predicted).  Stages on the
                                              predict -> correct -> verify path (most adaptive stages when n_para <= 10) accept the
                                              predicted ϕ_n when the ESS measured by the correction is within max(phi_rtol, 1e-10)
                                              of the target in ϕ units; values below 1e-10 therefore need phi_rtol < 0 to hold on
                                              every stage */
    int32_t stop_after_stage;              /* > 0: return (result.paused = 1) once cloud.stage_index has reached it - the save point of
                                              `save_intermediate` / `intermediate_stage_increment` (smc_main.jl:499-507); shards pause in lock step */
    int32_t continue_run;                  /* 1: go on from the handle's loop state (after a pause, or after smcmi_set_loop_state:
                                              `continue_intermediate`, smc_main.jl:334-335,355-361) instead of starting at stage 1 */
} smcmi_run_config;

typedef struct {
    int32_t n_stages;        /* cloud.stage_index at exit == number of tempering_schedule entries */
    int32_t resamples;       /* cloud.resamples */
    double logmdd;           /* Σ_n log((1/N) Σ_i w[i,n] W[i,n-1]) */
    double c, accept;        /* cloud.c, cloud.accept */
    double seconds;          /* wall time of the loop = cloud.total_sampling_time (smc_main.jl:489-490) */
    double kernel_ms_mutate; /* HIP-event time spent in the mutation kernel over the run (0 if not measured) */
    int32_t n_mutate_launches;
    int64_t solver_passes;   /* particle passes spent in the adaptive-ϕ solver over the run */
    int32_t solver_stalls;   /* stages that ran out of enqueued solver passes and were resumed by the host */
    int32_t select_stalls;   /* stages enqueued without selection kernels that had to resample after all (host resumed them); on a fixed
                                schedule with a large cloud every stage is enqueued that way, so this equals the number of resample stages */
    int32_t spec_stalls;     /* stages enqueued without a certificate pass whose predicted ϕ_n was unusable / not verified (resumed) */
    int32_t paused;          /* 1: stopped at stop_after_stage with ϕ_n < 1; continue with continue_run = 1 */
    int32_t n_segments;      /* persistent stage-segment launches of the run (small clouds on one handle: runs of stages that neither
                                resample nor need a certificate pass execute as one launch each) */
    int32_t segment_stages;  /* stages those launches completed (use_graph = 2: the stages behind kernel_ms_segments) */
    double kernel_ms_segments; /* HIP-event time of the segment launches (use_graph = 2, else 0) */
    int32_t segment_blocks;  /* blocks (one per CU, all resident) of a segment launch of this run: workers + gatherers; 0: the run had no segment */
    int32_t segment_state;   /* the handle's persistent-segment state after the run: 1 the residency self-test passed; 0 segments do not apply to this
                                cloud / were not tried; -1 the self-test failed or a hand-over timed out (e.g. a partitioned or shared GPU where
                                not every block is resident): every later run of the handle uses launches only */
    int32_t segment_timeouts;/* runs of this handle so far that met a hand-over time-out inside a segment and were repeated as launches from the
                                cloud they started with (each cost its time-out, SMCMI_SEG_TIMEOUT_MS, before the repeat) */
    int32_t shift_fallback_stage; /* fixed schedules: the stage from which the run shifted the incremental weights by the cloud's CURRENT largest
                                energy because a stage's sums overflowed under the lagged shift of the one-hand-over stage; 0: never (the rule) */
} smcmi_result;

typedef struct {             /* the loop scalars an intermediate save holds (smc_main.jl:499-507: cloud fields + j) */
    int32_t stage_index;     /* cloud.stage_index (i) */
    int32_t j;               /* 1-based position in the proposed fixed schedule */
    int32_t resampled_last_period;   /* the reference does not save this one: it continues with false */
    int32_t resamples;       /* cloud.resamples */
    double phi_n;            /* cloud.tempering_schedule[i] */
    double phi_prop;         /* proposed_fixed_schedule[j] when continuing (smc_main.jl:361) */
    double c, accept;        /* cloud.c, cloud.accept */
    double ess;              /* cloud.ESS[i] */
    double logmdd;           /* running Σ log((1/N) Σ w W) over the stages done so far */
} smcmi_loop_state;

typedef struct {             /* what one correction step reports (smc_main.jl:401-432) */
    double ess, sum_unnorm, logz_inc;
    int32_t resample;        /* ESS < threshold_ratio * N */
} smcmi_stage_stats;

/* ---- lifetime ------------------------------------------------------------------------------- */
int smcmi_create(const smcmi_config *cfg, smcmi_handle **out);          /* Cloud(n_params, n_parts), particle.jl:50-53 */
int smcmi_destroy(smcmi_handle *h);
const char *smcmi_last_error(void);
int smcmi_version(void);

/* ---- model: ParameterVector + likelihoods (sendto(workers(), parameters/data), smc_main.jl:169-170) */
int smcmi_set_parameters(smcmi_handle *h, const int32_t *fixed, const double *lo, const double *hi,
                         const int32_t *prior_family, const double *prior_a, const double *prior_b);
int smcmi_set_likelihood(smcmi_handle *h, int32_t which, int32_t family, const double *par, int64_t n_par,
                         const double *data, int64_t rows, int64_t cols, const double *aux, int64_t aux_rows,
                         int64_t aux_cols);

/* ---- user likelihood on the host: the reference's `loglikelihood::Function` argument of smc() (src/smc_main.jl:118), called
   inside mutation() after update!/prior (src/mutation.jl:93-121).  Batch form: theta is m x d column-major (proposal k =
   theta[k + m*j], j < d) holding only proposals that passed the bounds check; write out[k] = log-likelihood (-Inf allowed, NaN is
   taken as -Inf like the reference's try/catch); return 0, anything else aborts smcmi_run with SMCMI_ERR_CALLBACK.  Invoked
   synchronously on the thread that called smcmi_run / smcmi_initialize_likelihoods / smcmi_eval_cloud_callback, never from another
   thread.  Inside smcmi_run / smcmi_run_sharded the batch of one MH step x block arrives in K = min(8, max(1, n / 12288)) CHUNKS of
   ceil(n / K) consecutive particles, the last one possibly shorter (n = 98 305: 7 x 12 289 + 12 282; a shard below 24 576 particles: one
   chunk = the batch): one invocation per chunk (two with an old-data callback), each on its own m x d block, while the next chunk is
   still crossing PCIe; the results do not depend on the chunking, smcmi_callback_stats counts invocations = chunks.  A callback that
   keeps per-batch state or counts its invocations sets SMCMI_CB_CHUNKS=1 in the environment (whole batches, one invocation per MH step x
   block as in rounds 1-4).  which = SMCMI_WHICH_NEW: loglikelihood(parameters, data);
   SMCMI_WHICH_OLD: old_loglikelihood(parameters, old_data) (tempered updates; leave unset when old_data is empty).  fn = NULL
   unregisters.  With a callback registered smcmi_run keeps ϕ solver, correction, selection, moments, proposal and the MH decision
   on the device and ships only the n x d proposals and the n log-likelihoods across PCIe per step. */
typedef int (*smcmi_lik_callback)(const double *theta, int64_t m, int64_t d, double *out, void *user_data);
int smcmi_set_likelihood_callback(smcmi_handle *h, int32_t which, smcmi_lik_callback fn, void *user_data);
/* loglh (column = n_para, which = NEW) or old_loglh (column = n_para + 2, which = OLD) of the uploaded cloud from the callback:
   rows whose logprior column is -Inf are skipped (-Inf).  For initial clouds drawn by the caller (initial_draw!). */
int smcmi_eval_cloud_callback(smcmi_handle *h, int32_t which, int32_t column);
int smcmi_callback_stats(smcmi_handle *h, int64_t *calls, int64_t *evaluations);   /* of the last smcmi_run */
/* where the last run with a host callback spent its wall time on the calling thread, in ms (the first n <= 8 of: waiting for the propose
   kernel and the first chunk; waiting for later chunks; packing chunks that hold out-of-bounds proposals; inside the callback; NaN -> -Inf /
   scatter; enqueueing copies and kernels; the stage's device part up to the proposal incl. its sync; reserved) */
int smcmi_callback_phases(smcmi_handle *h, double *ms_out, int32_t n);

/* ---- cloud transfer (cloud.particles; get_vals/get_loglh/... read columns of the download) ---- */
int smcmi_upload_cloud(smcmi_handle *h, const double *particles);       /* n_local x R, column-major */
int smcmi_download_cloud(smcmi_handle *h, double *particles);
int smcmi_upload_cloud_device(smcmi_handle *h, const double *dev_particles);   /* device-to-device, same layout */
int smcmi_init_from_prior(smcmi_handle *h);                              /* initial_draw!, initialization.jl:88-119 */
int smcmi_initialize_likelihoods(smcmi_handle *h);                       /* initialize_likelihoods!, initialization.jl:153-186 */
/* ---- tempered update from an old cloud (smc_main.jl:244-333); both handles whole clouds on one device ----
   bridge_resample: resample(get_weights(old); n_parts = n_out, method) and copy those rows (weights included, as
   update_cloud! does) to rows [0, n_out) of dst (:266-279).  offsets as in smcmi_resample.
   copy_rows: vcat(bridge_cloud.particles, prior_cloud.particles) (:296).
   normalize_weights: zero_bad_loglh_weights! (optional) then normalize_weights! (:313-314, particle.jl:362-366,392-396). */
int smcmi_bridge_resample(smcmi_handle *dst, smcmi_handle *src, int32_t method, uint32_t stage, int64_t n_out,
                          const double *offsets, int64_t *ancestors_out);
int smcmi_copy_rows(smcmi_handle *dst, int64_t dst_row0, smcmi_handle *src, int64_t src_row0, int64_t n_rows);
int smcmi_normalize_weights(smcmi_handle *h, int32_t zero_bad_loglh);
int smcmi_cloud_device_ptr(smcmi_handle *h, double **dev_ptr, int64_t *ld); /* current buffer, for zero-copy hosts */

/* ---- stage primitives (same kernels smcmi_run launches) --------------------------------------- */
/* compute_ESS(loglh, weights, ϕ, ϕ_n1; old_loglh) for k candidate ϕ (helpers.jl:173-181) */
int smcmi_ess_at(smcmi_handle *h, const double *phis, int32_t k, double phi_prev, double *ess_out);
/* solve_adaptive_ϕ (helpers.jl:9-56); j is the reference's 1-based index */
int smcmi_solve_phi(smcmi_handle *h, const double *sched, int32_t n_phi, int32_t *j, double *phi_prop,
                    double phi_prev, double tempering_target, double ess_prev, int32_t *resampled_last,
                    double *phi_n);
/* correction: incremental weights, update_weights!, normalize_weights!, ESS (smc_main.jl:401-432) */
int smcmi_correct(smcmi_handle *h, double phi_n, double phi_prev, double prior_weight, double log_prob_old_data,
                  double threshold_ratio, smcmi_stage_stats *out);
/* resample(normalized_weights/n_parts; method) + gather + reset_weights! (smc_main.jl:438-442, resample.jl:23-72).
   offsets: NULL => Philox (stage); else 1 offset (systematic) or n_parts offsets (multinomial). */
int smcmi_resample(smcmi_handle *h, int32_t method, uint32_t stage, const double *offsets, int64_t *ancestors_out);
/* weighted_mean / weighted_cov (particle.jl:481-483, 526-529); cov row-major d x d */
int smcmi_moments(smcmi_handle *h, double *mean, double *cov);
/* all particles' mutation() (mutation.jl:56-138, smc_main.jl:472-484); mu_free/Sigma_free are θ̄_fr, R_fr;
   blocks 0-based: block b = block_idx[block_ptr[b] .. block_ptr[b+1]) over free-parameter positions */
int smcmi_mutate(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                 const int32_t *blocks_free, int32_t n_blocks, double phi_n, double phi_prev, double c, double alpha,
                 int32_t n_mh_steps, uint32_t stage, double *accept_mean_out);
/* host-callback split of mutation for arbitrary user likelihoods: propose -> (host evaluates) -> accept */
int smcmi_propose(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                  const int32_t *blocks_free, int32_t n_blocks, int32_t block, int32_t mh_step, double c, double alpha,
                  uint32_t stage, double *proposals_out /* n_local x d col-major */, double *logprior_out,
                  double *q_diff_out);
int smcmi_accept(smcmi_handle *h, const double *loglik_new, const double *loglik_old_new, double phi_n,
                 int32_t block, int32_t mh_step, int32_t n_blocks, uint32_t stage, int32_t last);

/* ---- whole loop on device (smc_main.jl:377-508) ------------------------------------------------ */
int smcmi_run(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res);
/* Number of stages the handle holds records / history columns for: what the last run (or smcmi_set_stage_records / _set_history /
   _set_loop_state) left, at most max_stages.  The two getters below copy exactly that many entries / columns - size the buffers by it,
   not by the n_stages a caller expects. */
int smcmi_stages_held(smcmi_handle *h, int32_t *n_stages_out);
/* per-stage records: cloud.tempering_schedule, cloud.ESS, c, accept, resample flags; arrays of smcmi_stages_held() entries */
int smcmi_get_stage_records(smcmi_handle *h, double *phi, double *ess, double *c, double *accept, int32_t *resampled);
int smcmi_get_history(smcmi_handle *h, double *w, double *W);           /* n_local x smcmi_stages_held() each, column-major */
/* intermediate save / continue (smc_main.jl:334-361, 499-507): the loop scalars, and - for a continuation in a fresh
 * handle - the records and history columns of the stages already done (the cloud itself goes through smcmi_upload_cloud) */
int smcmi_get_loop_state(smcmi_handle *h, smcmi_loop_state *out);
int smcmi_set_loop_state(smcmi_handle *h, const smcmi_loop_state *in);
int smcmi_set_stage_records(smcmi_handle *h, int32_t n_stages, const double *phi, const double *ess, const double *c,
                            const double *accept, const int32_t *resampled);
int smcmi_set_history(smcmi_handle *h, int32_t n_stages, const double *w, const double *W);

/* ---- shard-level pieces for multi-GPU hosts (one handle per GPU; host does the collective) ----- */
/* Every stage step is split as: partial (kernel writes this shard's partial sums into the comm buffer) ->
   host all-reduces comm buffer over ranks -> apply (kernels consume the global totals). */
int smcmi_comm_buffer(smcmi_handle *h, double **dev_ptr, int64_t *capacity);
int smcmi_comm_read(smcmi_handle *h, double *out, int64_t count);             /* synchronous device-to-host copy of comm[0..count) */
int smcmi_shard_ess_partial(smcmi_handle *h, const double *phis, int32_t k, double phi_prev);       /* comm[0..2k) = Σv, Σv² */
int smcmi_shard_correct_partial(smcmi_handle *h, double phi_n, double phi_prev, double prior_weight,
                                double log_prob_old_data, int32_t stage_col);                         /* comm[0..2) */
int smcmi_shard_normalize_moments_partial(smcmi_handle *h, double sum_unnorm, int32_t resampled,
                                          const double *shift, int32_t stage_col);                    /* comm[0..1+d+d(d+1)/2) */
int smcmi_shard_weights_device_ptr(smcmi_handle *h, double **dev_ptr);
int smcmi_shard_resample(smcmi_handle *h, const double *dev_full_weights, const double *dev_full_cloud, int32_t method,
                         uint32_t stage, int64_t *ancestors_out);
int smcmi_shard_mutate_partial(smcmi_handle *h, const double *mu_free, const double *Sigma_free, const int32_t *block_ptr,
                               const int32_t *blocks_free, int32_t n_blocks, double phi_n, double phi_prev, double c,
                               double alpha, int32_t n_mh_steps, uint32_t stage);                      /* comm[0] = Σ accept */
int smcmi_sync(smcmi_handle *h);

/* ---- sharded whole-loop drivers (csrc/sharded.hpp) -------------------------------------------------
   One handle per GPU/process, equal contiguous shards.  smcmi_comm_unique_id on rank 0 -> broadcast the 128 bytes by any means
   -> smcmi_comm_init on every rank (RCCL communicator over xGMI) -> smcmi_run_sharded: the loop of smcmi_run with in-stream
   all-reduces of the stage sums and an all-gather on resample stages.  smcmi_run_group drives several handles of ONE process
   in lock step (single-process multi-shard / multi-GPU; host-mediated sums).
   n_para <= 16 with device likelihood families runs on the two-launch stage (per-stage sums through the peer mailbox when every
   rank could map and test it, runs of stages without resampling as one persistent launch per rank); a handle with a registered
   likelihood callback (smcmi_set_likelihood_callback on EVERY rank, src/smc_main.jl:472-476 `parallel = true`) scores the
   proposals it holds through that callback inside the same call.  A hand-over that runs out (SMCMI_ERR_TIMEOUT: a GPU shared
   after the residency test, a stalled peer) voids a run; persistent-segment time-outs are repeated as launches from a
   device-side snapshot before the call returns. */
int smcmi_comm_unique_id(uint8_t *id_out /* 128 bytes */);
int smcmi_comm_init(smcmi_handle *h, int32_t rank, int32_t world, const uint8_t *id);
int smcmi_run_sharded(smcmi_handle *h, const smcmi_run_config *rc, smcmi_result *res);
/* Host-mediated communicator: the collectives of smcmi_run_sharded carried by caller-supplied functions on HOST buffers of doubles -
   whatever transport the host already has (MPI, torch.distributed / gloo, Julia's Distributed) - instead of RCCL.  Meant for ranks
   that share one GPU or have no RCCL (multi-process tests on one box, bring-up); the reference's counterpart is the serialisation
   `@distributed` does per stage (src/smc_main.jl:472-476).  Semantics (all blocking, called on the thread that called smcmi_run_sharded,
   same sequence on every rank; return 0 on success):
     allgather : recv[r * count .. (r+1) * count) = rank r's send[0 .. count)
     alltoallv : send[send_displs[p] .. + send_counts[p]) goes to rank p, recv[recv_displs[p] .. + recv_counts[p]) comes from rank p
                 (counts in doubles; the own rank's counts are 0).  May be NULL: resample stages then all-gather the shard clouds.
     barrier   : returns once every rank has called it
   The peer mailbox (below) works with this communicator as with RCCL: the tables' IPC handles travel through allgather. */
typedef struct {
    int (*allgather)(const double *send, double *recv, int64_t count, void *user);
    int (*alltoallv)(const double *send, const int64_t *send_counts, const int64_t *send_displs, double *recv,
                     const int64_t *recv_counts, const int64_t *recv_displs, void *user);
    int (*barrier)(void *user);
    void *user;
} smcmi_host_comm;
int smcmi_comm_init_host(smcmi_handle *h, int32_t rank, int32_t world, const smcmi_host_comm *comm);
/* Peer mailbox (n_para <= 10, at most 8 handles): the two per-stage hand-overs of a sharded run (correction sums, mutation sums:
   8 x 70 and 8 x 34 doubles) written straight into every peer's fine-grained table over xGMI instead of two all-gathers.
   smcmi_run_sharded sets it up by itself on its first call - it exchanges the tables' IPC handles through the communicator, runs
   256 test exchanges on every rank and keeps the all-gathers unless all of that succeeded everywhere (SMCMI_MAILBOX=0, read at every
   run, keeps the all-gathers).  Results are the same bits either way.
   The three calls below do the same by hand for callers that exchange the 64-byte handles themselves (and for the tests):
   export -> all-gather the handles by any means -> import (rank-ordered, 64 bytes each) -> selftest on all ranks at once
   (errors_out = mismatches + time-outs of `rounds` exchanges with every peer). */
int smcmi_mailbox_export(smcmi_handle *h, uint8_t *handle_out /* 64 bytes */);
int smcmi_mailbox_import(smcmi_handle *h, int32_t rank, int32_t world, const uint8_t *all_handles /* world x 64 bytes */);
int smcmi_mailbox_selftest(smcmi_handle *h, int32_t rank, int32_t world, int32_t rounds, int32_t *errors_out);
int smcmi_mailbox_active(smcmi_handle *h, int32_t *active_out);   /* 1: the last sharded run handed its per-stage sums over through the mailbox */
int smcmi_run_group(smcmi_handle **hs, int32_t n, const smcmi_run_config *rc, smcmi_result *res);
/* parity aid: compute_proposal_densities(para_draw, para_subset, d_subset = MvNormal(mu, Sigma), c, alpha) (src/helpers.jl:128-164;
   quirk Q1: the diagonal component's density uses the unscaled Σ_ii) evaluated by the dense mixture code of the alpha < 1 mutation
   kernels on the current device, for one block of d <= 16 entries; Sigma row-major d x d.  q0 / q1 as the reference returns them. */
int smcmi_debug_proposal_densities(const double *para_draw, const double *para_subset, const double *mu, const double *Sigma,
                                   int32_t d, double c, double alpha, double *q0, double *q1);

#ifdef __cplusplus
}
#endif
#endif
