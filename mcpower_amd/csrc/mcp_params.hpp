// mcp_params.hpp -- launch-constant block shared by host lowering and kernels.
//
// The host lowers a validated mcp_spec (reference SimulationSpec, spec.rs:251)
// into this flat block once per run_batch call; kernels read it through scalar
// (wave-uniform) loads.  Column layout is the reference's fixed kernel layout
// [0 intercept | 1..K continuous | factor dummies | interactions]
// (data_gen.rs:79-83).
#pragma once
#include <stdint.h>

namespace mcp {

constexpr int MAX_P = 16;        // design columns incl. intercept handled by the register-resident OLS kernels
constexpr int MAX_P_WAVE = 8;    // ... by the wave-per-replicate GLM kernels (Mle: 6)
constexpr int MAX_K = 15;        // continuous predictors
constexpr int MAX_GRID = 256;    // sample-size grid points per call (64 until round 3; only SimParams::ns is sized by it)
constexpr int MAX_T = 16;        // tests per fit (marginal targets + contrast pairs)
constexpr int MAX_P_FUSED = 12;   // ... by the fused lane-per-replicate OLS kernel (P = 9..12: Cholesky in place, <= 12 tests)
constexpr int MAX_T_WAVE = 8;    // ... in the GLM / LMM kernels (no contrast pairs there, targets <= P - 1)
constexpr int MAX_FACTORS = 4;
constexpr int MAX_LEVELS = 16;   // sum of levels over all factors
constexpr int MAX_PC = 28;       // post-hoc pairwise contrasts per fit (<= 7 dummies: one 8-level factor = C(8,2))
// Same-address atomics serialise in L2 at ~90 per microsecond (MI355X_MICROARCH.md, "dequeue"); with one
// atomic per wave per counter a 10^6-replicate launch would spend ~1.8 ms there.  Spreading the adds over
// 256 copies, each on its own cache lines, takes that off the critical path; a tiny kernel sums the copies.
constexpr int CTR_COPIES = 256;

constexpr int LMG_MAX_TERMS = 3;   // primary + MCP_MAX_EXTRA_GROUPINGS
constexpr int LMG_MAX_SLOPES = 2;  // MCP_MAX_RE_SLOPES
constexpr int LMG_MAX_THETA = 12;
constexpr uint32_t SCEN_CLASSES = 8192;  // transform class of a replicate: 6 + 6 marginal bits, 1 residual bit

struct SimParams {
  uint64_t base_seed;
  uint64_t sim_begin;      // sim_id of lane 0 of block 0
  uint32_t seed_per_replicate;  // MCP_SEED_PER_REPLICATE: every replicate is replicate 0 of its own call seed (replicate_key)
  uint32_t pad_seed_;
  uint32_t n_sims;         // replicates in this launch
  uint32_t out_base;       // row of the first replicate in the per-sim output tensors
  uint32_t max_n;
  uint32_t n_grid;
  uint32_t ns[MAX_GRID];   // strictly increasing sample sizes

  uint32_t P;              // total design columns
  uint32_t K;              // continuous predictors (columns 1..K)
  uint32_t ident_L;        // 1 when the correlation Cholesky factor is exactly I
  double L[MAX_K * MAX_K]; // row-major lower factor of the correlation (data_gen.rs:998-1005)
  double beta[MAX_P];      // effect_sizes

  // factors (data_gen.rs:509-551)
  uint32_t n_factors;
  uint32_t f_levels[MAX_FACTORS];
  uint32_t f_col[MAX_FACTORS];       // first dummy column of factor f
  uint32_t f_prop_off[MAX_FACTORS];
  uint32_t f_sampled[MAX_FACTORS];   // 1 = categorical draw on the sequential stream
  uint32_t f_seq_slot[MAX_FACTORS];  // ordinal of this factor among the sampled ones
  double f_props[MAX_LEVELS];
  uint32_t n_seq_pre;                // sequential-stream draws consumed before the row loop
  uint32_t n_seq_per_row;            // sequential-stream draws consumed per row
  uint32_t col_factor[MAX_P];        // owning factor of a dummy column (else 0xffffffff)
  uint32_t col_level[MAX_P];         // level (>=1) a dummy column indicates
  uint32_t inter_mask[MAX_P];        // component-column bitmask of an interaction column (else 0)

  // tests (batch.rs:1045-1095): row t of `contrast` is the vector c with c'beta tested
  uint32_t n_tests;
  double contrast[MAX_T * MAX_P];
  int32_t correction;                // mcp_correction
  uint32_t report_overall;
  uint32_t crit_stride;              // n_tests + 4

  // cluster random intercept (data_gen.rs:146-161, generation.rs:125-158)
  uint32_t has_cluster;
  uint32_t cluster_fixed_clusters;   // 1: row i -> i % v ; 0: row i -> i / v
  uint32_t cluster_v;
  uint32_t n_clusters_max;           // clusters present at max_n
  double tau;                        // sqrt(max(tau_squared, 0))
  uint32_t between_mask;             // columns broadcast from the cluster's representative row (data_gen.rs:606-636)
  uint32_t _pad_between;

  // outcome
  uint32_t outcome_kind;             // mcp_outcome_kind
  uint32_t estimator;                // mcp_estimator
  uint32_t glm_family;               // 0 Binary/logit, 1 Binary/probit, 2 Count/log (mixed_workspace::glmm_family)
  uint32_t _pad_family;

  // device pointers
  const uint8_t* fixed_levels;       // [max_n][n_factors] levels of the deterministic allocation walk
  const double* crit;                // [n_grid][crit_stride]
  uint8_t* unc;                      // [n_sims_total][n_grid][n_tests] or null
  uint8_t* cor;
  uint8_t* conv;                     // [n_sims_total][n_grid] or null
  uint8_t* overall;
  double* stat;                      // [n_sims_total][n_tests] or null
  uint8_t* boundary;                 // [n_sims_total][n_grid] boundary_hit (Mle arm) or null
  uint8_t* joint;                    // [n_sims_total][n_grid] joint Wald chi2 decision (Mle arm) or null
  unsigned long long* counters;      // [CTR_COPIES][n_grid][ctr_stride]: replicated, a wave adds into copy (wave id % CTR_COPIES)
  uint32_t ctr_copy_stride;          // u64 words between copies (a multiple of 16 = 128 B, so copies never share a line)
  uint32_t _pad_ctr;
  uint32_t ctr_stride, ctr_k_unc, ctr_k_cor, ctr_conv, ctr_overall, ctr_hist_unc, ctr_hist_cor, ctr_n_sims;
  uint32_t ctr_joint_unc, ctr_joint_cor, ctr_boundary_1, ctr_boundary_2;
  uint32_t truth_start;              // scenario.truth_start (Mle start theta)
  uint32_t _pad_tail;

  // ---- general DGP ("scenario") path: per-replicate design drawn from the scenario stream
  //      (data_gen.rs:990-1093, scenarios.rs:50-186), non-normal marginals / residuals, per-study
  //      beta jitter and heteroskedastic residual scaling (data_gen.rs:709-776, 876-888) ----
  uint32_t scen_active;            // 1: kernels take the general path and read the per-replicate design records
  uint32_t scen_optimistic;        // scenario.is_optimistic(): records are the spec's own L / types for every replicate
  uint32_t resid_extra_cols;       // 0, or df_int + 1: CLASS_RESID normal columns 0..df_int are generated
  uint32_t resid_may_uniform;      // a Uniform residual can occur: the slot-0 uniform is kept next to the normal
  uint32_t resid_spec_dist;        // spec.residual_dist
  uint32_t residual_pinned;
  double resid_df;                 // max(scenario.residual_df, 3)
  double het_total;                // scenario.heterogeneity (0 = off)
  uint32_t apply_hsk;
  int32_t hsk_driver;              // -1 = linear predictor
  double hsk_gamma, hsk_center, hsk_inv_std, hsk_inv_norm;
  double corr_noise_sd, dist_change_prob, resid_change_prob;
  uint32_t n_new_dist, n_resid_dists;
  int32_t new_dist[8], resid_dists[8];
  uint32_t var_types[MAX_K];
  uint32_t var_pinned[MAX_K];
  double var_params[MAX_K];
  double corr[MAX_K * MAX_K];      // base correlation, column-major
  double* scen_L;                  // [K(K+1)/2][scen_stride] row-major lower factor entries per replicate
  uint32_t* scen_meta;             // [scen_stride]: 3 bits per predictor marginal (K <= 9) | residual dist << 28 | status << 31
  double* scen_beta;               // [P][scen_stride] per-study effects (heterogeneity > 0) or null
  uint32_t scen_stride;
  uint32_t _pad_scen;
  // divergence-aware replicate order of the lane-per-replicate general-path kernel (scenario_kernels.hip): replicates
  // counted per transform class, then scattered so that the lanes of a wave share the class; null = identity
  uint32_t* scen_hist;             // [SCEN_CLASSES + 1] class counts -> exclusive offsets, or null
  uint32_t* scen_perm;             // [scen_stride] lane -> local replicate index, or null
  const double* t3_table;          // 2048-knot standardized t(3) PPF table (marginals/t3.rs) or null
  // uploaded data (data_gen.rs:552-604, :1243-1277): NORTA quantile table and the strict-bootstrap frame
  const double* upload_normal;     // [upload_rows][K] sorted standardised values per non-factor column, or null
  uint32_t upload_rows;
  uint32_t bootstrap;              // 1: strict mode, one source row of `upload_data` per design row
  const double* upload_data;       // [upload_data_rows][upload_data_cols] raw frame, or null
  uint32_t upload_data_rows, upload_data_cols;
  int32_t boot_map[MAX_K + MAX_FACTORS];  // frame column of non-factor j / factor K + f, -1 = synthetic
  uint32_t n_sampled_factors;      // the bootstrap uniform follows the per-row categorical draws
  // LME scenario knobs (data_gen.rs:146-161, 1069-1076): random-effect shape and per-replicate ICC jitter
  uint32_t lme_re_dist;            // mcp_residual_dist of the cluster random intercept (Normal without scenario.lme)
  uint32_t re_draws_per_cluster;   // sequential-stream draws one cluster effect consumes (1, df, or 1 + df)
  double lme_re_df;
  double lme_icc_noise_sd;
  double tau_squared;              // spec cluster.tau_squared
  double* scen_tau_sq;             // [scen_stride] realised tau^2 per replicate (general path with a cluster) or null

  // ---- reduced-design path (batch.rs:925-1213, 2418-2562): per-(replicate, N) sparse-factor exclusion, a
  //      test_formula subset of the columns, post-hoc pairwise contrasts.  A kept-column bitmask selects the
  //      fitted model; every df-dependent threshold of the round comes from the row built at p_red columns ----
  uint32_t red_active;             // 1: the OLS launch takes the reduced-design kernel
  uint32_t min_level_count;        // factor_min_level_count (0 = exclusion off)
  uint32_t keep_mask[16];          // kept columns for each combination of per-factor exclusion flags (bit f = factor f)
  uint32_t n_posthoc;              // post-hoc contrasts in total (all blocks, batch.rs:2189-2223 order)
  uint8_t ph_pos[MAX_PC];          // column carrying +1
  uint8_t ph_neg[MAX_PC];          // column carrying -1, 0xff when the pair is against the reference level
  uint8_t test_pos[MAX_T];         // main tests in the same (+column, -column | 0xff) form
  uint8_t test_neg[MAX_T];
  const uint16_t* red_row;         // [n_grid][16]: threshold row of (N, flag combination); 0xffff = cannot occur
  const double* crit_red;          // rows of crit_stride doubles: [t_unc^2, corr[nt], posthoc, joint, overall, posthoc corr[npc]]
  uint8_t* posthoc_unc;            // [n_sims_total][n_grid][n_posthoc] or null
  uint8_t* posthoc_cor;
  uint8_t* factor_excluded;        // [n_sims_total][n_grid][n_factors] or null
  uint32_t ctr_posthoc_unc, ctr_posthoc_cor, ctr_excl_sparse, ctr_excl_sep;
  double joint_crit_k[MAX_T + 1];  // chi2_ppf(1 - alpha, k): joint Wald threshold over k kept targets (Mle arm)

  // LMM split path: per-(replicate, grid point) cluster sums between the prepare and optimise kernels
  double* lmm_rec;                 // [field][n_grid][lmm_rec_stride]; the general-path OLS launch parks its sufficient
                                   // statistics here in the same layout between its row and fit kernels (ols_kernels.hip, SPLIT)
  uint32_t* lmm_rec_flags;         // [n_grid][lmm_rec_stride]: exclusion flag bits | LMM_FLAG_BALANCED
  uint32_t lmm_rec_stride;         // replicates per chunk, padded to 64
  uint32_t lmg_stream;             // general LMM, single-size call without cluster-level columns or factors: the design
                                   // streams through a 256-row LDS window instead of staying resident (lmm_general_impl.hpp)

  // ---- general mixed model: random slopes, crossed / nested extra groupings (generation.rs:196-427,
  //      data_gen.rs:163-284, 812-842; mixed_workspace.rs:57-198).  Term 0 = primary grouping ----
  uint32_t lmg_active;
  uint32_t lmg_n_terms;              // 1 + extra groupings
  uint32_t lmg_q[LMG_MAX_TERMS];     // random-effect width per term (1 + slopes)
  uint32_t lmg_levels[LMG_MAX_TERMS];  // levels at max_n
  uint32_t lmg_relation[LMG_MAX_TERMS];  // term >= 1: 0 crossed, 1 nested within the primary
  uint32_t lmg_count[LMG_MAX_TERMS];   // term >= 1: n_clusters (crossed) / n_per_parent (nested)
  uint32_t lmg_cols[LMG_MAX_TERMS][LMG_MAX_SLOPES];  // design column of each slope
  uint32_t lmg_roff[LMG_MAX_TERMS];  // first column of term t >= 1 inside the dense "rest" block
  uint32_t lmg_theta_off[LMG_MAX_TERMS];
  uint32_t lmg_n_theta, lmg_R, lmg_W;  // theta length; rest width; W = R + P + 1
  uint32_t lmg_seq[2 * LMG_MAX_TERMS];   // first sequential-stream word of [intercepts of term t | slopes of term t]
  uint32_t _pad_lmg;
  // fit view of the general LMM (lmm_general_impl.hpp): the likelihood does not care which grouping is "primary", so the
  // kernel eliminates, block by block, the grouping with the MOST levels that still has block structure -- a grouping
  // nested in the primary one when there is such (its parents then join the dense "rest" columns: lmm_nested 20 + 80
  // levels: dense width 86 -> 26).  lmg_elim = 0 is the generator's own order; theta stays in declaration order.
  uint32_t lmg_elim;                  // term whose levels are eliminated one block each
  uint32_t lmg_cpe;                   // eliminated levels per primary level (1 when lmg_elim = 0)
  uint32_t lmg_n_rest;                // = lmg_n_terms - 1
  uint32_t lmg_frest[LMG_MAX_TERMS];  // the other terms, declaration order
  uint32_t lmg_froff[LMG_MAX_TERMS];  // by term id: first column inside the dense rest block
  uint32_t lmg_frel[LMG_MAX_TERMS];   // by term id: 0 = every eliminated level meets every level of it; 1 = only those of its own primary level
  uint32_t lmg_fcnt[LMG_MAX_TERMS];   // by term id: levels per primary level (1 for the primary grouping itself)
  uint32_t lmg_fR, lmg_fW;            // rest width and W = fR + P + 1 in this view
  // compact strips: an eliminated level only meets the columns of its own primary level in a grouped term (frel = 1), every
  // column of the others, X and y -- its strip in Z0'[Zr | X | y] (gzw) and in the eliminated factor (B) is stored that wide
  // (lmm_nested: 7 of 26 columns).  lmg_Wc = lmg_fW and identity positions when lmg_compact = 0 (reduced rounds need full strips)
  uint32_t lmg_compact, lmg_Wc;
  uint32_t lmg_coff[LMG_MAX_TERMS];   // by term id: first compact position of the term's columns
  uint32_t lmg_coff_xy;               // first compact position of X | y
  double lmg_tau[LMG_MAX_TERMS];     // sqrt(max(tau^2, 0)) per term
  double lmg_chol[LMG_MAX_TERMS][9]; // row-major lower chol(diag(tau) R diag(tau)) per term: the slope draws
  double lmg_tau_sq[LMG_MAX_TERMS];  // spec tau^2 per term (the ICC jitter of the general path starts from it)
  double lmg_slope_sd[LMG_MAX_TERMS][LMG_MAX_SLOPES];  // sqrt(slope variance)
  double lmg_recorr[LMG_MAX_TERMS][9];                 // re_correlation_matrix(), row-major q x q
  double lmg_theta_truth[LMG_MAX_THETA];
  uint16_t* pinned;                  // [n_sims_total][n_grid] pinned-component bits (spec.rs:566-573) or null
  double* tau_hat;                   // [n_sims_total][n_grid] tau-hat^2 of a clustered-GLM fit (spec.rs:589-593) or null
  uint32_t glmm_nagq;                // clustered GLM, random intercept: 1 = Laplace, odd k <= 25 = adaptive Gauss-Hermite (spec.nagq)
  double glmm_gz[25], glmm_gw[25];   // the k-point rule for the standard normal weight (nodes descending, sum w = 1)
  uint32_t glmm_hessian;             // clustered GLM: Wald SE from the finite-difference Hessian (spec.wald_se)
  uint32_t glmm_active;              // Glm estimator + cluster: the random-intercept GLMM kernel
  uint32_t glmm_keep;                // fitted columns of the clustered GLM: all, or the test_formula subset
  uint32_t glmm_pirls_cap;           // joint-mode kernel: Fisher steps per evaluation (60; MCP_DEBUG_GLMM_PIRLS_CAP lowers it in tests)

  // staged pipeline: design tile [P columns: x1..x_{P-1}, y][max_n][tile_sims] f32
  float* tile;
  uint32_t tile_sims;
};

}  // namespace mcp
