/* oracle_internal.h -- CPU ORACLE (test infrastructure): shared internals. */
#ifndef MCP_ORACLE_INTERNAL_H
#define MCP_ORACLE_INTERNAL_H
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "mcp_oracle.h"

#define MCPO_MAX_K 16      /* continuous predictors */
#define MCPO_MAX_P 40      /* design columns */
#define MCPO_MAX_LEVELS 64 /* sum of factor levels */

typedef struct mcpo_ws {
  float* x;          /* max_n x p, column-major */
  float* y;          /* max_n */
  float* z;          /* max_n x k planar normals */
  float* resid;      /* max_n */
  float* resid_g;    /* max_n */
  double* acc;       /* max_n (mix_acc) */
  double* eta;       /* max_n (eta_acc) */
  float* cluster_u;  /* cap_clusters */
  uint32_t* cluster_ids;
  uint32_t cap_clusters;
  float* slope_u;                                 /* cap_clusters x MCP_MAX_RE_SLOPES: primary slope draws, [c * n_slopes + k] */
  float* extra_u[MCP_MAX_EXTRA_GROUPINGS];        /* levels x q_g */
  uint32_t* extra_ids[MCP_MAX_EXTRA_GROUPINGS];   /* max_n */
  size_t cap_extra[MCP_MAX_EXTRA_GROUPINGS];
} mcpo_ws;
int mcpo_ws_size_extras(mcpo_ws* ws, const mcp_spec* spec, uint32_t max_n);
uint32_t mcpo_extra_n_levels_at(const mcp_cluster* c, uint32_t g, uint32_t n);
uint32_t mcpo_extra_level_of_row(const mcp_cluster* c, uint32_t g, uint32_t i);

mcpo_ws* mcpo_ws_new(uint32_t max_n, uint32_t p, uint32_t k, uint32_t cap_clusters);
void mcpo_ws_free(mcpo_ws* ws);
int mcpo_generate_ws(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n, mcpo_ws* ws);
int mcpo_chol_colmajor(double* a, uint32_t k);
int mcpo_spec_supported(const mcp_spec* spec);
uint32_t mcpo_cluster_of_row(const mcp_cluster* c, uint32_t i);
uint32_t mcpo_n_clusters_at(const mcp_cluster* c, uint32_t n);
void mcpo_compute_het_coeffs(const mcp_spec* spec, double* lp_mean, double* lp_std, double* col_mean, double* col_std);

/* general mixed model (fit_lmm_general.c): one term per grouping, primary first */
#define MCPO_GEN_MAX_TERMS (1 + MCP_MAX_EXTRA_GROUPINGS)
typedef struct mcpo_re_term {
  uint32_t q, n_levels;
  const uint32_t* ids;                  /* level id per row */
  uint32_t cols[MCP_MAX_RE_SLOPES];     /* design column of each slope */
} mcpo_re_term;
typedef struct mcpo_gen_fit {
  int converged;
  uint8_t boundary_hit, at_upper;
  uint64_t pinned;                      /* bit k: diagonal variance component k at 0, order [primary intercept, slopes.., extra_1 ..] */
  uint32_t n_eval, n_theta, p;
  double theta[12];
  double dev, sigma_sq;
  double beta[MCPO_MAX_P];
  double L[MCPO_MAX_P * MCPO_MAX_P];    /* row-major lower Cholesky factor of RX'RX (= X'V^-1X up to sigma^2) */
} mcpo_gen_fit;
void mcpo_slope_cholesky(const mcp_re_slopes* sl, double tau0, double* l);
uint32_t mcpo_truth_theta(const mcp_cluster* c, int include_slopes, double* theta);
uint32_t mcpo_elim_term(const mcp_cluster* c, uint32_t max_n, int equal_q);
int mcpo_lmm_general_fit_terms(const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                               const mcpo_re_term* terms, uint32_t n_terms, uint32_t elim, const double* theta_start, mcpo_gen_fit* f);

/* GLM family kernel shared by fit_glm.c and fit_glmm.c: IRLS weight, working response, deviance contribution */
void mcpo_fam_eval(int fam, double eta, double yi, double* w_out, double* z_out, double* dev_out);
int mcpo_glmm_fit_ids(int fam, const float* x, const float* y, const uint32_t* ids, uint32_t n, uint32_t ld, uint32_t p,
                      uint32_t ncl, double theta0, const double* beta0, uint32_t mode, mcpo_gen_fit* f);
/* mode bits of the GLMM fits: Wald SE from the finite-difference Hessian instead of "rx" */
#define MCPO_MODE_HESSIAN 1u  /* bits 8..15: nagq (0 / 1 = Laplace) */
void mcpo_gh_rule(uint32_t k, double* z, double* w);
int mcpo_glmm_general_fit_terms(int fam, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                                const mcpo_re_term* terms, uint32_t n_terms, uint32_t elim, const double* theta0, const double* beta0,
                                uint32_t mode, mcpo_gen_fit* f);
/* EXPERIMENT KNOBS (tools/oracle2_perdraw.py; DESIGN.md section 5, "oracle2 experiments"): environment variable
 * MCPO_KNOB_<NAME>, read at every call (the harness sets them per worker process); unset = the shipped constant, so the
 * checker's results are unchanged unless a knob is exported.  Never read by the product. */
#include <stdlib.h>
static inline double mcpo_knob(const char* name, double dflt) {
  char key[64] = "MCPO_KNOB_";
  strncat(key, name, sizeof key - 11);
  const char* v = getenv(key);
  return v && *v ? strtod(v, NULL) : dflt;
}
typedef double (*mcpo_dev_fn)(void* ctx, double* par);
int mcpo_fd_hessian_factor(mcpo_dev_fn fn, void* ctx, const double* par, uint32_t nth, uint32_t p, const uint8_t* theta_free,
                           double fx, double* l_out);

/* Stationarity check of a simplex result; it replaces the unconditional restart of the four mixed-model optimisers
 * (a restart re-shrinks a fresh simplex from 0.05 to the tolerance: ~160 evaluations that, in 2 000 of 2 000 replicates
 * of the reference bench's glmm_intercept, moved no decision).  Central differences of the objective along every
 * coordinate with lme4's deriv12 step h = 1e-4 (h = par_k for a bounded parameter closer than that to 0, the single
 * probe +h for one pinned AT 0): the point is accepted when no probe is lower than f0 by more than 1e-8 and the Newton
 * decrease g^2 / (2 H) each coordinate predicts stays below 1e-8 -- the same 1e-8 the restart loop calls "no progress".
 * 2 d evaluations.  A rejected point takes the restart loop as before.  lower0[k] != 0: parameter k is bounded below
 * by 0 (a diagonal of the covariance template).  Probes are issued k ascending, +h then -h. */
#define MCPO_STAT_H 1e-4
#define MCPO_STAT_TOL 1e-8
#define MCPO_STAT_MAX_DIM 32
static inline int mcpo_stationary(mcpo_dev_fn fn, void* ctx, const double* par, uint32_t d, const uint8_t* lower0, double f0) {
  double x[MCPO_STAT_MAX_DIM];
  if (d > MCPO_STAT_MAX_DIM || !isfinite(f0)) return 0;
  for (uint32_t k = 0; k < d; ++k) {
    double h = MCPO_STAT_H;
    const int bounded = lower0 && lower0[k];
    memcpy(x, par, sizeof(double) * d);
    if (bounded && par[k] == 0.0) {
      x[k] = h;
      const double fp = fn(ctx, x);
      if (!(fp >= f0 - MCPO_STAT_TOL)) return 0;
      continue;
    }
    if (bounded && par[k] < h) h = par[k];
    x[k] = par[k] + h;
    const double fp = fn(ctx, x);
    memcpy(x, par, sizeof(double) * d);
    x[k] = par[k] - h;
    const double fm = fn(ctx, x);
    if (!(isfinite(fp) && isfinite(fm)) || fp < f0 - MCPO_STAT_TOL || fm < f0 - MCPO_STAT_TOL) return 0;
    const double g = (fp - fm) / (2.0 * h), H = (fp - 2.0 * f0 + fm) / (h * h);
    if (H > 0.0 ? !(g * g <= 2.0 * MCPO_STAT_TOL * H) : !(fabs(fp - fm) <= 2.0 * MCPO_STAT_TOL)) return 0;
  }
  return 1;
}

int mcpo_glmm_slopes_fit_ids(int fam, const float* x, const float* y, const uint32_t* ids, uint32_t n, uint32_t ld,
                             uint32_t p, uint32_t ncl, uint32_t q, const uint32_t* slope_cols, const double* theta0,
                             const double* beta0, uint32_t mode, mcpo_gen_fit* f);

/* OLS sufficient statistics + fit (fit_ols.c) */
typedef struct mcpo_olsfit_t {
  uint32_t p;
  int converged;
  double beta[MCPO_MAX_P];
  double L[MCPO_MAX_P * MCPO_MAX_P]; /* row-major lower Cholesky factor of X'X */
  double sigma_sq, rss, sst, df_resid;
} mcpo_olsfit_t;

typedef struct mcpo_suff {
  uint32_t p;
  double xtx[MCPO_MAX_P * MCPO_MAX_P]; /* row-major, lower triangle valid */
  double xty[MCPO_MAX_P];
  double yty, sum_y;
  uint64_t n_rows;
} mcpo_suff;

void mcpo_suff_reset(mcpo_suff* s, uint32_t p);
void mcpo_suff_add_rows(mcpo_suff* s, const float* x, const float* y, uint32_t ld, uint32_t from, uint32_t to);
void mcpo_fit_suff(const mcpo_suff* s, mcpo_olsfit_t* fit);
/* squared statistic of c'beta; NaN when not well formed (posthoc.rs:86-99) */
double mcpo_contrast_t_sq(const mcpo_olsfit_t* fit, const double* c);

uint32_t mcpo_fixed_level_next(const double* probs, uint32_t* counts, uint32_t n_levels);

/* sparse-factor exclusion / reduced designs (exclusion.c) */
typedef struct mcpo_red_crit {
  uint32_t n, p_red;
  double* row;  /* nt + 4: [t_unc^2, corr[0..nt), posthoc, joint, overall] at p_red fitted columns */
  double* post; /* npc post-hoc correction thresholds */
} mcpo_red_crit;
typedef struct mcpo_red_cache {
  mcpo_red_crit* e;
  uint32_t count, cap;
} mcpo_red_cache;
void mcpo_update_factor_exclusions(const float* x, uint32_t ld, const mcp_spec* spec, uint32_t from, uint32_t to,
                                   uint32_t* prefix_counts, uint8_t* flags);
uint32_t mcpo_build_exclusion_remap(const mcp_spec* spec, const uint8_t* flags, uint32_t* kept_cols,
                                    int32_t* col_remap);
int mcpo_smallest_min_count_factor(const mcp_spec* spec, const uint32_t* prefix_counts, const uint8_t* flags);
const mcpo_red_crit* mcpo_reduced_crit(mcpo_red_cache* cache, const mcp_spec* spec, uint32_t n, uint32_t p_red);
void mcpo_red_cache_free(mcpo_red_cache* cache);
void mcpo_set_error(const char* fmt, ...);
#endif
