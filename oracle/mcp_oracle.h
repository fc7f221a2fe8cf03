/*
 * mcp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's simulate loop
 * (pawlenartowicz/MCPower, crates/engine-core) used only as the checker for the
 * HIP path: by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg.  Nothing under mcpower_amd/ may include, link or call it.
 *
 * It shares the POD layouts of include/mcpower_hip.h (mcp_spec,
 * mcp_batch_result) so a test can hand the same spec to both sides; it shares
 * no code with the product.
 *
 * Parity pins (see oracle/README.md):
 *   RNG / uniform / normal / categorical / scenario stream -- bit-exact against
 *     crates/engine-core/tests/{rng_byte_identity.rs,golden_rng.rs};
 *   full DGP (design + outcome)  -- bit-exact against validation/data/NAME.rds
 *     (introspect::capture_sim0_data output) for OLS / logit / LME cases;
 *   OLS fit + critical values    -- 1e-11 rel / 1e-9 abs against NAME.golden.rds
 *     (R lm / qt on the same bytes).
 */
#ifndef MCP_ORACLE_H
#define MCP_ORACLE_H

#include <stdint.h>
#include "../include/mcpower_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG primitives (rng.rs:22-38, rand_philox 0.1.0) -------------------- */
uint64_t mcpo_splitmix64(uint64_t z);
uint64_t mcpo_pcg_mix64(uint64_t base_seed, uint64_t sim_id);
void mcpo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float mcpo_u32_to_unit_f32(uint32_t w);
float mcpo_norm_inv_cdf_f32(float u);

/* sequential SimRng (rng.rs:51-128) */
typedef struct mcpo_simrng {
  uint32_t key[2];
  uint64_t counter;
  uint32_t buf[4];
  uint32_t buf_pos;
} mcpo_simrng;
void mcpo_simrng_new(mcpo_simrng* r, uint64_t base_seed, uint64_t sim_id);
uint32_t mcpo_simrng_next_u32(mcpo_simrng* r);
float mcpo_simrng_next_uniform(mcpo_simrng* r);
float mcpo_simrng_next_normal(mcpo_simrng* r);
uint32_t mcpo_simrng_next_categorical(mcpo_simrng* r, const double* probs, uint32_t n);

/* planar fills (rng.rs:154-261) */
void mcpo_fill_words(const uint32_t key[2], uint32_t cls, uint32_t col, uint32_t* out, uint32_t n);
void mcpo_fill_uniform_column(const uint32_t key[2], uint32_t cls, uint32_t col, float* out, uint32_t n);
void mcpo_fill_normal_column(const uint32_t key[2], uint32_t cls, uint32_t col, float* out, uint32_t n);

void mcpo_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out);

/* golden-test helpers: n draws of one stream kind for (base_seed, sim_id);
 * kind 0 = next_normal, 1 = next_uniform, 2 = scenario_rng next_uniform */
void mcpo_stream_f32(uint64_t base_seed, uint64_t sim_id, int kind, float* out, uint32_t n);
void mcpo_stream_categorical(uint64_t base_seed, uint64_t sim_id, const double* probs, uint32_t n_probs,
                             uint32_t* out, uint32_t n);

/* distributions.rs:32-97 (libm exp in place of glmm's owned exp: <= 2^-52 abs on phi) */
double mcpo_phi(double z);
double mcpo_marginal_uniform(double z, double a, double b);

/* ---- critical values (critvals.rs) --------------------------------------- */
double mcpo_norm_ppf(double p);
double mcpo_t_ppf(double p, double df);
double mcpo_chi2_ppf(double p, double k);
double mcpo_f_ppf(double p, double dfn, double dfd);
/* per sample size: [t_crit_sq_unc, correction[0..n_targets), posthoc, joint, overall] */
int mcpo_build_crit_table(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, double* out);
int mcpo_build_crit_table_p(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t p_total,
                            const double* tukey_k, double* out);
int mcpo_build_posthoc_corr(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t p_total, double* out);
double mcpo_q_tukey_ppf(double p, double k, double df);

/* correction.rs:16-92 */
void mcpo_apply_correction(int method, const double* t_sq, uint32_t t, double crit_unc,
                           const double* crit_sq, uint8_t* out);

/* ---- DGP (data_gen.rs:87-925) --------------------------------------------
 * x_out: max_n x p column-major f32; y_out: max_n f32; cluster_ids_out may be NULL. */
int mcpo_generate_sim_data(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n,
                           float* x_out, float* y_out, uint32_t* cluster_ids_out);
/* same, plus the level ids of the extra groupings: extra_ids_out is n_extra_groupings x max_n (may be NULL) */
int mcpo_generate_sim_data_ids(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n,
                               float* x_out, float* y_out, uint32_t* cluster_ids_out, uint32_t* extra_ids_out);

/* ---- general LMM on provided bytes (fit_lmm_general.c): random slopes, crossed / nested groupings --------
 * ids: n_terms x ld level ids (term 0 = primary); slope_cols: n_terms x MCP_MAX_RE_SLOPES design columns;
 * elim: the term whose levels are eliminated block by block (0 = the primary; the result does not depend on it beyond rounding);
 * theta_start may be NULL (blind start); vars_out: n_terms x 9, sigma^2 Lambda Lambda' per term, row-major 3 x 3 */
int mcpo_lmm_general_fit(const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, uint32_t n_terms,
                         const uint32_t* term_q, const uint32_t* term_levels, const uint32_t* ids,
                         const uint32_t* slope_cols, uint32_t elim, const double* theta_start, double* beta, double* se, double* z_sq,
                         double* theta_hat, double* sigma_sq, double* vars_out, uint8_t* converged,
                         uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval);

/* ---- random-intercept GLMM (Laplace) on provided bytes (fit_glmm.c); family 0 logit, 1 probit, 2 Poisson ---- */
int mcpo_glmm_fit(int family, const float* x, const float* y, const uint32_t* cluster_ids, uint32_t n, uint32_t ld,
                  uint32_t p, uint32_t n_clusters, double theta_start, const double* beta_start, double* beta, double* se,
                  double* z_sq, double* theta_hat, double* tau_sq, uint8_t* converged, uint8_t* boundary_hit,
                  uint32_t* n_eval, uint32_t mode /* bit 0: Wald SE mode "hessian"; bits 8..15: nagq */);
void mcpo_gh_rule(uint32_t k, double* z, double* w);

/* ---- GLMM with any carried random-effect structure (fit_glmm_general.c): oracle only ---- */
int mcpo_glmm_general_fit(int family, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, uint32_t n_terms,
                          const uint32_t* term_q, const uint32_t* term_levels, const uint32_t* ids, const uint32_t* slope_cols,
                          const double* theta_start, const double* beta_start, double* beta, double* se, double* theta_hat,
                          double* vars_out, uint8_t* converged, uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval,
                          uint32_t mode);

/* test hook: Fisher-step cap of fit_glmm_general.c (default 60 = GG_PIRLS_ITERS; out-of-range values restore it) */
void mcpo_set_glmm_general_pirls_cap(int cap);

/* ---- GLMM with random slopes on the primary grouping (fit_glmm_slopes.c): q = 1 + slopes <= 3 ---- */
int mcpo_glmm_slopes_fit(int family, const float* x, const float* y, const uint32_t* cluster_ids, uint32_t n, uint32_t ld,
                         uint32_t p, uint32_t n_clusters, uint32_t q, const uint32_t* slope_cols, const double* theta_start,
                         const double* beta_start, double* beta, double* se, double* theta_hat, double* vars_out,
                         uint8_t* converged, uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval, uint32_t mode);

/* ---- OLS fit on provided bytes (glmm fit_suff_stats_t_sq, inferred from
 *      batch.rs:958-1142, posthoc.rs:62-99, introspect.rs:280-284) ----------- */
int mcpo_ols_fit(const float* x_colmajor, const float* y, uint32_t n_rows, uint32_t ld, uint32_t p,
                 double* beta, double* se, double* t_sq, double* rss, double* sst, uint8_t* converged);

/* ---- logistic IRLS on provided bytes (glmm glm_irls_fit; published Fisher scoring, see fit_glm.c) ---- */
int mcpo_glm_fit(const float* x_colmajor, const float* y, uint32_t n_rows, uint32_t ld, uint32_t p,
                 const double* beta_start, double* beta, double* se, double* z_sq, double* deviance,
                 double* deviance_null, uint8_t* converged, uint32_t* n_iter);
/* family: 0 = Binary/logit, 1 = Binary/probit, 2 = Count/log (Poisson) */
int mcpo_glm_fit_family(int family, const float* x_colmajor, const float* y, uint32_t n_rows, uint32_t ld, uint32_t p,
                        const double* beta_start, double* beta, double* se, double* z_sq, double* deviance,
                        double* deviance_null, uint8_t* converged, uint32_t* n_iter);

/* ---- random-intercept REML on provided bytes (glmm fit_on; closed-form profiled deviance, see fit_lmm.c) ---- */
int mcpo_lmm_fit(const float* x_colmajor, const float* y, const uint32_t* cluster_ids, uint32_t n_rows, uint32_t ld,
                 uint32_t p, uint32_t n_clusters, double theta_start, double* beta, double* se, double* z_sq,
                 double* theta_hat, double* sigma_sq, double* tau_sq, uint8_t* converged, uint8_t* boundary_hit,
                 uint32_t* n_eval);

/* ---- run_batch (batch.rs:199-776) ----------------------------------------
 * Same contract as mcp_run_batch; `n_threads` >= 1 splits the replicate range
 * over pthreads (results identical for any thread count). */
int mcpo_run_batch(const mcp_spec* spec, const uint32_t* sample_sizes, uint32_t n_sample_sizes,
                   uint32_t n_sims, uint64_t base_seed, const mcp_run_options* opts, uint32_t n_threads,
                   mcp_batch_result* out);
void mcpo_free_result(mcp_batch_result* r);
const char* mcpo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
