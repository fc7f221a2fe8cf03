/*
 * mcpower_hip.h -- C ABI of the MI355X-native Monte-Carlo power engine.
 *
 * This is the drop-in boundary for ONE path of pawlenartowicz/MCPower:
 * `engine-core`'s simulate loop.  The reference exposes that path only as a
 * Rust API (`engine-core` is an rlib with `#![forbid(unsafe_code)]`,
 * crates/engine-core/Cargo.toml:12-13), so every entry point below cites the
 * Rust item it replaces; INTEGRATION.md shows the `extern "C"` block a
 * maintainer adds on the Rust side.
 *
 *   reference item                                  (file:line)               C entry point
 *   ----------------------------------------------  ------------------------  -------------------------
 *   run_batch(spec, sample_sizes, n_sims, seed,     batch.rs:130              mcp_run_batch
 *             sink) -> Result<BatchResult,_>
 *   run_batch_st (sequential, bit-identical)        batch.rs:157              mcp_run_batch (same result
 *                                                                             by construction; the GPU
 *                                                                             path has no thread order)
 *   run_batch_st_capture (+ squared statistics)     batch.rs:180              mcp_run_batch + MCP_WANT_STAT_CAPTURE
 *   set_n_threads(n)                                batch.rs:93               mcp_set_device
 *   SimulationSpec                                  spec.rs:251-417           mcp_spec
 *   ScenarioPerturbations                           spec.rs:118-237           mcp_scenario
 *   ClusterSpec / ClusterSizing                     engine-contract/src/
 *                                                   generation.rs:121-171     mcp_cluster
 *   BatchResult / ResultShape                       spec.rs:535-600           mcp_batch_result
 *   EngineError                                     spec.rs:20-29             mcp_status + mcp_last_error
 *   ProgressSink::report / is_cancelled             spec.rs:38-52             mcp_progress_fn
 *   aggregate_batch counters (what a multi-GPU      orchestrator/src/
 *     merge pools)                                  aggregation.rs:61-330     mcp_counter_layout + counters[]
 *   merge_power_results (pool per-worker counters   orchestrator/src/         mcp_comm_unique_id / mcp_comm_init /
 *     on the host; here: one RCCL all-reduce of     merge.rs:26-70            mcp_comm_destroy + MCP_ALLREDUCE_COUNTERS,
 *     the device-resident counter vector)                                     mcp_shard_range
 *   introspect::capture_sim0_data (debug: the       introspect.rs:104         mcp_generate_design
 *     generated design/outcome of chosen sims)
 *   introspect::fit_provided_data (debug: fit       introspect.rs:174         mcp_fit_design
 *     caller-provided bytes with the shipped solver)
 *
 * Ownership: every pointer inside `mcp_spec` is borrowed and read-only for the
 * duration of the call (the reference clones the spec, batch.rs:336).  Result
 * buffers are allocated by the library and released with mcp_free_result.
 *
 * Threading: entry points are re-entrant; the only process-global state is the
 * selected device (mcp_set_device) and a thread-local error string.
 *
 * Determinism: results are a pure function of
 * (spec, sample_sizes, n_sims, base_seed, sim_offset) -- independent of the
 * number of GPUs the replicate range is split over (batch.rs:2734-2752).
 */
#ifndef MCPOWER_HIP_H
#define MCPOWER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCP_ABI_VERSION 3u

/* ---- status codes (EngineError, spec.rs:20-29) --------------------------- */
typedef enum mcp_status {
  MCP_OK = 0,
  MCP_ERR_INVALID_SPEC = 1,       /* EngineError::InvalidSpec(String)      */
  MCP_ERR_CORRELATION_NOT_PSD = 2,/* EngineError::CorrelationNotPSD        */
  MCP_ERR_RANK_DEFICIENT = 3,     /* EngineError::RankDeficient(u32)       */
  MCP_ERR_CANCELLED = 4,          /* EngineError::Cancelled                */
  MCP_ERR_UNSUPPORTED = 5,        /* valid spec, feature not on the device path yet: fails loudly, never falls back to a CPU path */
  MCP_ERR_DEVICE = 6              /* HIP runtime error / no gfx950 device  */
} mcp_status;

/* ---- enums mirrored from engine-contract --------------------------------- */
typedef enum mcp_distribution {   /* engine_contract::Distribution */
  MCP_DIST_NORMAL = 0,
  MCP_DIST_BINARY = 1,
  MCP_DIST_RIGHT_SKEWED = 2,
  MCP_DIST_LEFT_SKEWED = 3,
  MCP_DIST_HIGH_KURTOSIS = 4,
  MCP_DIST_UNIFORM = 5,
  MCP_DIST_UPLOADED_DATA = 6,
  MCP_DIST_UPLOADED_BINARY = 7,
  MCP_DIST_UPLOADED_FACTOR = 8
} mcp_distribution;

typedef enum mcp_residual_dist {  /* engine_contract::ResidualDist */
  MCP_RESID_NORMAL = 0,
  MCP_RESID_HIGH_KURTOSIS = 1,
  MCP_RESID_RIGHT_SKEWED = 2,
  MCP_RESID_LEFT_SKEWED = 3,
  MCP_RESID_UNIFORM = 4
} mcp_residual_dist;

typedef enum mcp_outcome_kind { MCP_OUTCOME_CONTINUOUS = 0, MCP_OUTCOME_BINARY = 1, MCP_OUTCOME_COUNT = 2 } mcp_outcome_kind;
typedef enum mcp_link_kind { MCP_LINK_CANONICAL = 0, MCP_LINK_PROBIT = 1 } mcp_link_kind;
/* engine-contract estimator.rs:39-56: fixed-effect Wald SE of the clustered GLM; Rx is the default */
typedef enum mcp_wald_se { MCP_WALD_SE_RX = 0, MCP_WALD_SE_HESSIAN = 1 } mcp_wald_se;
typedef enum mcp_estimator { MCP_EST_OLS = 0, MCP_EST_GLM = 1, MCP_EST_MLE = 2 } mcp_estimator;
typedef enum mcp_correction {     /* engine_contract::CorrectionMethod */
  MCP_CORR_NONE = 0,
  MCP_CORR_BONFERRONI = 1,
  MCP_CORR_HOLM = 2,
  MCP_CORR_BH = 3,
  MCP_CORR_TUKEY_HSD = 4
} mcp_correction;
typedef enum mcp_cluster_sizing { MCP_CLUSTER_FIXED_CLUSTERS = 0, MCP_CLUSTER_FIXED_SIZE = 1 } mcp_cluster_sizing;

/* ---- ScenarioPerturbations (spec.rs:118-176) ------------------------------ */
typedef struct mcp_scenario {
  double heterogeneity;
  double heteroskedasticity_ratio;      /* 1.0 = off */
  double correlation_noise_sd;
  double distribution_change_prob;
  const int32_t* new_distributions;     /* mcp_distribution menu */
  uint32_t n_new_distributions;
  double residual_change_prob;
  const int32_t* residual_dists;        /* mcp_residual_dist menu */
  uint32_t n_residual_dists;
  double residual_df;
  uint8_t sampled_factor_proportions;
  uint8_t truth_start;
  uint8_t has_lme;                      /* scenario.lme.is_some() */
  uint8_t _pad0;
  int32_t lme_random_effect_dist;       /* mcp_residual_dist */
  double lme_random_effect_df;
  double lme_icc_noise_sd;
} mcp_scenario;

/* ---- ClusterSpec (engine-contract/src/generation.rs:121-428) ----------------
 * Primary grouping (random intercept + optional random slopes) and up to two
 * extra groupings, crossed with or nested within the primary. */
#define MCP_MAX_RE_SLOPES 2u          /* random slopes per grouping: q = 1 + n_slopes <= 3 */
#define MCP_MAX_EXTRA_GROUPINGS 2u

typedef enum mcp_grouping_relation {  /* GroupingRelation, generation.rs:244-252 */
  MCP_GROUPING_CROSSED = 0,           /* count = n_clusters */
  MCP_GROUPING_NESTED_WITHIN = 1      /* count = n_per_parent */
} mcp_grouping_relation;

/* the SlopeTerm list of one grouping (generation.rs:196-241), already resolved to kernel design columns
 * (spec.cluster_slope_design_cols / spec.extra_slope_cols, contract_adapter.rs) */
typedef struct mcp_re_slopes {
  uint32_t n_slopes;
  uint32_t design_cols[MCP_MAX_RE_SLOPES];   /* kernel design column of each slope (intercept = column 0) */
  uint32_t _pad;
  double variance[MCP_MAX_RE_SLOPES];
  double re_corr[(1 + MCP_MAX_RE_SLOPES) * (1 + MCP_MAX_RE_SLOPES)]; /* re_correlation_matrix(): row-major q x q over [intercept, slopes...], q = 1 + n_slopes */
} mcp_re_slopes;

typedef struct mcp_grouping {         /* GroupingSpec */
  int32_t relation;                   /* mcp_grouping_relation */
  uint32_t count;
  double tau_squared;
  mcp_re_slopes slopes;
} mcp_grouping;

typedef struct mcp_cluster {
  uint8_t present;                      /* spec.cluster.is_some() */
  uint8_t _pad[3];
  int32_t sizing;                       /* mcp_cluster_sizing */
  uint32_t sizing_value;                /* n_clusters or cluster_size */
  uint32_t n_extra_groupings;
  double tau_squared;
  mcp_re_slopes slopes;                 /* primary random slopes */
  mcp_grouping extra[MCP_MAX_EXTRA_GROUPINGS];
} mcp_cluster;

/* ---- SimulationSpec (spec.rs:251-417), field for field -------------------- */
typedef struct mcp_spec {
  uint32_t abi_version;                 /* MCP_ABI_VERSION */
  uint32_t n_non_factor;
  uint32_t n_factor_dummies;
  const double* correlation;            /* n_non_factor^2, column-major */
  const int32_t* var_types;             /* n_non_factor mcp_distribution */
  const uint8_t* var_pinned;            /* n_var_pinned entries (may be 0) */
  uint32_t n_var_pinned;
  const double* var_params;             /* n_var_params entries */
  uint32_t n_var_params;
  const double* upload_normal;          /* (rows, cols) row-major */
  uint32_t upload_normal_rows, upload_normal_cols;
  const double* upload_data;
  uint32_t upload_data_rows, upload_data_cols;
  const int32_t* bootstrap_frame_map;   /* -1 = None */
  uint32_t n_bootstrap_frame_map;
  const uint32_t* between_var_indices;
  uint32_t n_between_var_indices;
  const int32_t* factor_n_levels;
  uint32_t n_factors;
  const double* factor_proportions;     /* sum(factor_n_levels) */
  uint32_t n_factor_proportions;
  const int8_t* factor_sampled;         /* -1 inherit, 0 exact, 1 sampled */
  uint32_t n_factor_sampled;
  const double* effect_sizes;           /* 1 + n_non_factor + n_factor_dummies + n_interactions */
  uint32_t n_effect_sizes;
  const uint32_t* target_indices;
  uint32_t n_target_indices;
  const uint32_t* fit_columns;
  uint32_t n_fit_columns;
  const uint32_t* contrast_pairs;       /* 2 * n_contrast_pairs: (positive, negative) */
  uint32_t n_contrast_pairs;
  const uint32_t* interaction_components; /* CSR payload */
  const uint32_t* interaction_offsets;  /* n_interactions + 1 */
  uint32_t n_interactions;
  int32_t correction_method;            /* mcp_correction */
  double alpha;                         /* crit_values.alpha */
  double posthoc_alpha;                 /* NaN = None */
  int32_t heteroskedasticity_driver;    /* -1 = None (linear predictor) */
  int32_t residual_dist;                /* mcp_residual_dist */
  uint8_t residual_pinned;
  uint8_t report_overall;
  uint8_t wald_se;                      /* mcp_wald_se: clustered GLM only (spec.rs:359-362); 0 = Rx, the default */
  uint8_t nagq;                         /* spec.rs:363-366: 0 / 1 = Laplace; odd k <= 25 = adaptive Gauss-Hermite (random intercept only) */
  int32_t outcome_kind;                 /* mcp_outcome_kind */
  int32_t link;                         /* mcp_link_kind */
  int32_t estimator;                    /* mcp_estimator */
  double intercept;                     /* metadata only (spec.rs:367-371) */
  const uint32_t* posthoc_factor_index; /* one per posthoc block */
  const uint32_t* posthoc_target_offsets; /* n_posthoc + 1 */
  const uint32_t* posthoc_target_indices;
  uint32_t n_posthoc;
  double max_failed_fraction;           /* evaluated by the caller */
  uint32_t factor_min_level_count;
  uint32_t _pad2;
  mcp_cluster cluster;
  mcp_scenario scenario;
} mcp_spec;

/* ---- counters: the additive reduction `aggregate_batch` performs
 *      (orchestrator/src/aggregation.rs:84-284).  `counters` is a flat
 *      uint64 vector; per sample-size index the block layout is given by
 *      mcp_counter_layout (offsets relative to `n_idx * stride`).  This vector
 *      is what ranks all-reduce (sum) over RCCL. ----------------------------- */
typedef struct mcp_counter_layout {
  uint32_t stride;          /* uint64 words per sample-size index */
  uint32_t k_unc;           /* [n_targets] success counts, uncorrected */
  uint32_t k_cor;           /* [n_targets] success counts, corrected */
  uint32_t converged;       /* [1] */
  uint32_t overall;         /* [1] (0 when report_overall is false) */
  uint32_t hist_unc;        /* [n_targets + n_posthoc_contrasts + 1] joint-success histogram */
  uint32_t hist_cor;        /* same */
  uint32_t posthoc_unc;     /* [n_posthoc_contrasts] */
  uint32_t posthoc_cor;     /* [n_posthoc_contrasts] */
  uint32_t excl_sparse;     /* [n_factors] factor_excluded == 1 */
  uint32_t excl_sep;        /* [n_factors] factor_excluded == 2 */
  uint32_t joint_unc;       /* [1] Mle joint Wald chi2 */
  uint32_t joint_cor;       /* [1] */
  uint32_t boundary_1;      /* [1] boundary_hit == 1 */
  uint32_t boundary_2;      /* [1] boundary_hit == 2 */
  uint32_t n_sims;          /* [1] replicates folded into this block */
} mcp_counter_layout;

/* ---- BatchResult (spec.rs:553-600) --------------------------------------- */
typedef struct mcp_batch_result {
  /* ResultShape */
  uint32_t n_sims, n_sample_sizes, n_targets, n_factors;
  uint32_t n_posthoc_contrasts, n_variance_components;
  /* sim-major tensors; NULL when MCP_WANT_PER_SIM is not requested */
  uint8_t* uncorrected;        /* (n_sims, n_ss, n_targets) */
  uint8_t* corrected;
  uint8_t* posthoc_unc;        /* (n_sims, n_ss, n_posthoc_contrasts) or NULL */
  uint8_t* posthoc_cor;
  uint8_t* converged;          /* (n_sims, n_ss) */
  uint8_t* boundary_hit;
  uint64_t* pinned_components;
  uint8_t* joint_unc;
  uint8_t* joint_cor;
  uint8_t* overall;            /* NULL unless spec.report_overall */
  uint8_t* factor_excluded;    /* (n_sims, n_ss, n_factors) or NULL */
  double* tau_squared_hat;     /* NaN except GLMM fits */
  double* stat_capture;        /* (n_sims, n_targets) squared statistics at grid point 0, NaN if not converged; NULL unless MCP_WANT_STAT_CAPTURE */
  /* reduced form: always filled */
  uint64_t* counters;          /* n_sample_sizes * layout.stride */
  mcp_counter_layout layout;
  /* timing of the device work inside the call (HIP events on the launch stream) */
  double device_ms;            /* all kernels of the call */
  double kernel_ms;            /* dominant kernel only, summed over launches (fused simulate kernel; fit kernel when staged) */
  double aux_kernel_ms;        /* staged pipeline: generator kernel, summed over launches; 0 otherwise */
  uint32_t n_kernel_launches;
  uint32_t _pad;
  /* when the first dominant-kernel launch of the call began and the last one ended, in ms since this host thread's
   * first call: batches in flight overlap on the device (mcp_run_batch_submit), so a caller that wants the kernel-busy
   * time of a region takes the union of these intervals instead of adding kernel_ms */
  double kernel_begin_ms, kernel_end_ms;
} mcp_batch_result;

/* progress / cancel: ProgressSink::report(current,total)->bool (spec.rs:41);
 * return 0 to cancel.  Called from the host thread between device chunks. */
typedef int (*mcp_progress_fn)(uint64_t current, uint64_t total, void* user);

/* option bits */
#define MCP_WANT_PER_SIM      0x1u   /* fill the sim-major u8 tensors (BatchResult parity) */
#define MCP_WANT_STAT_CAPTURE 0x2u   /* run_batch_st_capture (batch.rs:180) */
#define MCP_PIPELINE_STAGED   0x4u   /* generator -> HBM design -> fit kernels instead of the fused kernel (OLS, <= 8
                                         design columns; wider designs and the other estimators ignore the flag) */
#define MCP_SEED_PER_REPLICATE 0x10u /* validation aid (clustered GLM only): replicate i is replicate 0 of its OWN call seed,
                                        key_i = splitmix64(splitmix64(base_seed + sim_offset + i)) -- the draw sequence of the
                                        reference's per-draw harness (validation/oracle2_power.R:38-66: `.debug_seed = base + k`,
                                        create_data -> load_data), so that its frozen per-draw rates are same-data comparisons */
#define MCP_ALLREDUCE_COUNTERS 0x8u  /* pool `counters` over the ranks of this process's communicator: one
                                        ncclAllReduce(sum, uint64) of the device-resident vector on the launch stream,
                                        before the call's single device-to-host copy (merge.rs:26-70).  Every rank of
                                        the communicator must make the same call (same spec, sample sizes and flags;
                                        its own sim_offset / n_sims shard; a shard may be EMPTY, n_sims = 0, when
                                        there are fewer replicates than ranks: that rank launches nothing, joins the
                                        collective with zeros and returns the pooled counters).  Per-sim tensors stay
                                        local.  A rank that fails BEFORE its collective (device error, invalid spec)
                                        strands its peers, as in any collective program: validate with
                                        mcp_build_crit_table on every rank first, and treat such a failure as fatal
                                        for the job.  A progress
                                        callback that cancels on ANY rank cancels the call on EVERY rank: the verdict
                                        rides in the reduced payload, so all ranks leave the collective together */

typedef struct mcp_run_options {
  uint32_t flags;
  uint32_t chunk_sims;       /* replicates per device launch; 0 = library default */
  uint64_t sim_offset;       /* first sim_id of this call: this rank's shard is [sim_offset, sim_offset + n_sims) */
  void* hip_stream;          /* hipStream_t to launch on; NULL = the library's own stream */
} mcp_run_options;

/* ---- entry points --------------------------------------------------------- */

/* ABI version of the loaded library. */
uint32_t mcp_abi_version(void);

/* Thread-local message of the last non-OK status on this thread. */
const char* mcp_last_error(void);

/* Select the HIP device used by subsequent calls of this process (replaces
 * batch.rs:93 set_n_threads: one process per GPU). */
int mcp_set_device(int device_index);

/* Number of visible HIP devices (0 when there is none). */
int mcp_device_count(void);

/* run_batch / run_batch_st / run_batch_st_capture (batch.rs:130,157,180).
 * `sample_sizes` strictly increasing, [0] > 0 (batch.rs:236-253).
 * On MCP_OK `*out` is filled and must be released with mcp_free_result. */
int mcp_run_batch(const mcp_spec* spec, const uint32_t* sample_sizes, uint32_t n_sample_sizes,
                  uint32_t n_sims, uint64_t base_seed, const mcp_run_options* opts,
                  mcp_progress_fn progress, void* progress_user, mcp_batch_result* out);

void mcp_free_result(mcp_batch_result* r);

/* Asynchronous form of mcp_run_batch for back-to-back batches (no reference counterpart: the reference's
 * run_batch blocks, batch.rs:130; its callers overlap work by calling it from several host threads).
 * `submit` validates, lowers and enqueues the whole batch on the launch stream and returns at once with a
 * ticket; `collect` waits for that batch and fills `*out` (counters, layout, shape, timings -- the reduced
 * form: MCP_WANT_PER_SIM / MCP_WANT_STAT_CAPTURE are rejected, there is no progress callback).  Up to four
 * batches may be in flight per host thread, each on a stream of its own: the host-side work of batch i+1 hides
 * behind the kernels of batch i, and the partly filled last round of batch i's workgroups shares the chip with the
 * first round of batch i+1 (pass opts->hip_stream to force one in-order stream).  Results do not depend on the
 * order (every batch owns its buffers).  A ticket is collected exactly once, on the thread
 * that submitted it; the result is released with mcp_free_result. */
int mcp_run_batch_submit(const mcp_spec* spec, const uint32_t* sample_sizes, uint32_t n_sample_sizes,
                         uint32_t n_sims, uint64_t base_seed, const mcp_run_options* opts, uint64_t* ticket_out);
int mcp_run_batch_collect(uint64_t ticket, mcp_batch_result* out);

/* ---- multi-GPU: one process per GPU, ONE collective -------------------------------------------------------------
 * The reference pools per-worker counters on the host (merge_power_results, orchestrator/src/merge.rs:26-70; browser
 * worker pool ports/wasm/src/seeds.ts:98-103).  Here each process owns one MI355X; rank r of W simulates the replicate
 * slice mcp_shard_range gives it under the SAME base_seed (union bit-identical to a one-GPU run) and the counters are
 * summed by RCCL over xGMI while still in HBM.  Bring-up, once per process:
 *   rank 0:  mcp_comm_unique_id(&id);  ...host broadcasts the 128 bytes over its own channel...
 *   all:     mcp_set_device(local_rank);  mcp_comm_init(rank, world, &id);          (collective)
 *   per run: mcp_run_batch[_submit](..., opts.flags |= MCP_ALLREDUCE_COUNTERS, opts.sim_offset / n_sims = the shard)
 *   end:     mcp_comm_destroy();                                                    (collective)
 * RCCL is bound at the first of these calls (librccl.so.1); a process that never calls them never loads it. */
#define MCP_COMM_ID_BYTES 128u
typedef struct mcp_comm_id { uint8_t bytes[MCP_COMM_ID_BYTES]; } mcp_comm_id;   /* an ncclUniqueId, verbatim */

/* Create the rendezvous id of a new communicator (rank 0 only).  Needs the device: MCP_ERR_DEVICE without one. */
int mcp_comm_unique_id(mcp_comm_id* id_out);
/* Join the communicator as `rank` of `world` on the device chosen with mcp_set_device; returns once every rank has
 * joined and a one-word all-reduce has crossed the rings.  One communicator per process. */
int mcp_comm_init(int rank, int world, const mcp_comm_id* id);
/* Leave and free the communicator (waits for the device first).  No-op without one. */
int mcp_comm_destroy(void);
/* Host-only: bind RCCL now (librccl.so.1 beside the HIP runtime this library uses, or the file MCP_RCCL_PATH names).
 * MCP_OK, or MCP_ERR_DEVICE with the loader's message in mcp_last_error() -- what mcp_comm_init would fail with. */
int mcp_comm_probe(void);
int mcp_comm_rank(void);    /* -1 without a communicator */
int mcp_comm_world(void);   /*  0 without a communicator */
/* The sharding rule: rank r of W owns sim_ids [r*S/W, (r+1)*S/W) of a run of S replicates (host-only arithmetic). */
int mcp_shard_range(uint64_t n_sims_total, int rank, int world, uint64_t* offset_out, uint64_t* count_out);

/* Host-only part of run_batch (no device): validation (batch.rs:208-330) and
 * the critical-value table (critvals.rs:827-1012).  `crit_out` receives, per
 * sample size, [t_crit_sq_uncorrected, correction_t_crit_sq[0..n_targets),
 * posthoc_t_crit_sq, joint_t_crit_sq, overall_crit] = n_targets + 4 doubles. */
int mcp_build_crit_table(const mcp_spec* spec, const uint32_t* sample_sizes, uint32_t n_sample_sizes,
                         double* crit_out);

/* Scalar quantile functions of critvals.rs (t_ppf :211, norm_ppf :33,
 * chi2_ppf :356, f_ppf :450), exported so hosts can reuse the exact thresholds. */
double mcp_t_ppf(double p, double df);
double mcp_norm_ppf(double p);
double mcp_chi2_ppf(double p, double k);
double mcp_f_ppf(double p, double dfn, double dfd);

/* introspect::capture_sim0_data (introspect.rs:104) generalised to a range of
 * replicates: writes the generated design (column-major f32, max_n x p per
 * replicate, replicate-major) and outcome (max_n f32 per replicate) of
 * sim_id in [sim_offset, sim_offset + n_sims) into caller-provided HOST buffers. */
int mcp_generate_design(const mcp_spec* spec, uint32_t max_n, uint32_t n_sims, uint64_t base_seed,
                        uint64_t sim_offset, float* design_out, float* outcome_out);

/* introspect::fit_provided_data (introspect.rs:174) for the OLS arm: fit
 * caller-provided designs on the device with the shipped solver.
 * design: n_fits x (p columns x n rows) column-major f32; outcome: n_fits x n.
 * Outputs per fit: beta[p], se[p], t_sq[p], rss, sst, converged. */
int mcp_fit_design(const float* design, const float* outcome, uint32_t n_fits, uint32_t n, uint32_t p,
                   double* beta_out, double* se_out, double* t_sq_out, double* rss_out, double* sst_out,
                   uint8_t* converged_out);

/* Debug probe of the frozen f32 inverse-normal-CDF kernel (rng.rs:354):
 * out[i] = norm_inv_cdf_f32(u32_to_unit_f32(first_word + i * stride)), i < n.
 * Lets a host test sweep all 2^23 reachable uniforms (rng.rs:539). */
int mcp_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MCPOWER_HIP_H */
