/*
 * oracle/dgp.c -- CPU ORACLE (test infrastructure): per-replicate data
 * generating process.
 *
 * Follows /root/reference/crates/engine-core/src/data_gen.rs
 *   chol_lower :55, generate_sim_data :87-925, fixed_level_next :943,
 *   populate_design :990-1093, apply_marginal(_column) :1106-1250,
 *   draw_residual :1289-1320,
 * scenarios.rs (scenario_rng :50, perturb_correlation :67, perturb_var_types
 * :127, pick_residual :163), correlation.rs (factor_only :48,
 * psd_repair_and_factor :82), distributions.rs (sample_t :78),
 * marginals/t3.rs (T3PpfTable :44-96), spec.rs (compute_het_coeffs :436-515)
 * and engine-contract/src/generation.rs:125-158 (ClusterSizing row maps).
 *
 * Pinned bit-exact against validation/data/*.rds (design + outcome of sim 0)
 * for the Normal-marginal / Normal-residual OLS, logit and random-intercept
 * cases.  NOT bitwise (documented "parity unpinned" below the f64 transform
 * plane): paths that go through glmm's owned exp/ln/sigmoid/phi kernels
 * (<= 1-2 ULP of libm, distributions.rs:21-23) -- the non-normal marginals,
 * heteroskedastic scaling and the logit draw use libm here -- and the scenario
 * PSD repair (faer self_adjoint_eigen replaced by a cyclic Jacobi sweep).
 * Unsupported (returns MCP_ERR_UNSUPPORTED): uploaded-data marginals,
 * bootstrap frame maps.
 */
#include <pthread.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mcp_oracle.h"
#include "oracle_internal.h"

#define CLASS_XNORM 1u /* rng.rs:141 */
#define CLASS_RESID 2u /* rng.rs:147 */
#define STREAM_TAG_SCENARIO 0x5C5C5C5C5C5C5C5CULL
#define STREAM_TAG_HET 0x4848484848484848ULL
#define FLOAT_NEAR_ZERO 1e-30 /* lib.rs:46 */

static const double EXP_CAP = 6.95925599364711;
static const double EXP_CENSORED_MEAN = 0.9990501970288289;
static const double EXP_CENSORED_STD = 0.9933676327697134;
static const double SQRT3 = 1.7320508075688772;

/* ---- t(3) PPF table (marginals/t3.rs:44-96) -------------------------------- */
#define T3_RES 2048
static double g_t3[T3_RES];
static pthread_once_t g_t3_once = PTHREAD_ONCE_INIT;
static const double T3_MIN = 0.00121, T3_MAX = 0.99879;

/* built once per process, by whichever replicate thread asks first: the unguarded lazy build it replaces let two
 * threads of one batch fill and rescale the table concurrently (found by tests/test_fuzz_parity.py) */
static void t3_build_once(void) {
  double denom = (double)(T3_RES - 1);
  for (int i = 0; i < T3_RES; ++i) {
    double p = T3_MIN + (T3_MAX - T3_MIN) * (double)i / denom;
    g_t3[i] = mcpo_t_ppf(p, 3.0);
  }
  double du = (T3_MAX - T3_MIN) / denom;
  double m1 = T3_MIN * g_t3[0] + (1.0 - T3_MAX) * g_t3[T3_RES - 1];
  double m2 = T3_MIN * g_t3[0] * g_t3[0] + (1.0 - T3_MAX) * g_t3[T3_RES - 1] * g_t3[T3_RES - 1];
  for (int i = 0; i + 1 < T3_RES; ++i) {
    double a = g_t3[i], b = g_t3[i + 1];
    m1 += du * (a + b) / 2.0;
    m2 += du * (a * a + a * b + b * b) / 3.0;
  }
  double sd = sqrt(m2 - m1 * m1);
  for (int i = 0; i < T3_RES; ++i) g_t3[i] /= sd;
}

static void t3_build(void) { pthread_once(&g_t3_once, t3_build_once); }

static double t3_lookup(double pct) {
  double p = pct < T3_MIN ? T3_MIN : (pct > T3_MAX ? T3_MAX : pct);
  double scale = (double)(T3_RES - 1) / (T3_MAX - T3_MIN);
  double idx = (p - T3_MIN) * scale;
  size_t ii = (size_t)floor(idx);
  double frac = idx - (double)ii;
  if (ii + 1 >= T3_RES) return g_t3[T3_RES - 1];
  return g_t3[ii] * (1.0 - frac) + g_t3[ii + 1] * frac;
}

/* ---- small dense helpers --------------------------------------------------- */
/* Lower Cholesky, column-major k x k, in place; upper triangle zeroed.
 * correlation.rs:48 delegates to faer llt; for these tiny matrices the
 * left-looking inner-product order below reproduced faer bit-for-bit on every
 * correlated fixture (ols_corr_*, ols_interaction_a, glm_two_a). */
int mcpo_chol_colmajor(double* a, uint32_t k) {
  for (uint32_t j = 0; j < k; ++j) {
    double d = a[j * k + j];
    for (uint32_t t = 0; t < j; ++t) d -= a[t * k + j] * a[t * k + j];
    if (!(d > 0.0)) return -1;
    double ljj = sqrt(d);
    a[j * k + j] = ljj;
    for (uint32_t i = j + 1; i < k; ++i) {
      double s = a[j * k + i];
      for (uint32_t t = 0; t < j; ++t) s -= a[t * k + i] * a[t * k + j];
      a[j * k + i] = s / ljj;
    }
  }
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < j; ++i) a[j * k + i] = 0.0;
  return 0;
}

/* cyclic Jacobi eigen-decomposition of a symmetric k x k (column-major);
 * eigenvectors in the columns of v.  Stand-in for faer self_adjoint_eigen
 * (correlation.rs:128): not bitwise, validated statistically. */
static void jacobi_eig(double* a, double* v, double* w, uint32_t k) {
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < k; ++j) v[j * k + i] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0;
    for (uint32_t i = 0; i < k; ++i)
      for (uint32_t j = i + 1; j < k; ++j) off += a[j * k + i] * a[j * k + i];
    if (off < 1e-30) break;
    for (uint32_t p = 0; p < k; ++p)
      for (uint32_t q = p + 1; q < k; ++q) {
        double apq = a[q * k + p];
        if (fabs(apq) < 1e-300) continue;
        double theta = (a[q * k + q] - a[p * k + p]) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (uint32_t r = 0; r < k; ++r) {
          double arp = a[p * k + r], arq = a[q * k + r];
          a[p * k + r] = c * arp - s * arq;
          a[q * k + r] = s * arp + c * arq;
        }
        for (uint32_t r = 0; r < k; ++r) {
          double apr = a[r * k + p], aqr = a[r * k + q];
          a[r * k + p] = c * apr - s * aqr;
          a[r * k + q] = s * apr + c * aqr;
        }
        for (uint32_t r = 0; r < k; ++r) {
          double vrp = v[p * k + r], vrq = v[q * k + r];
          v[p * k + r] = c * vrp - s * vrq;
          v[q * k + r] = s * vrp + c * vrq;
        }
      }
  }
  for (uint32_t i = 0; i < k; ++i) w[i] = a[i * k + i];
}

/* correlation.rs:82-190 */
static int psd_repair_and_factor(double* m, uint32_t k) {
  double tmp[MCPO_MAX_K * MCPO_MAX_K];
  memcpy(tmp, m, sizeof(double) * k * k);
  if (mcpo_chol_colmajor(tmp, k) == 0) {
    memcpy(m, tmp, sizeof(double) * k * k);
    return 0;
  }
  double sym[MCPO_MAX_K * MCPO_MAX_K], v[MCPO_MAX_K * MCPO_MAX_K], w[MCPO_MAX_K], rep[MCPO_MAX_K * MCPO_MAX_K];
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < k; ++i) sym[j * k + i] = (i >= j) ? m[j * k + i] : m[i * k + j];
  jacobi_eig(sym, v, w, k);
  for (uint32_t t = 0; t < k; ++t)
    if (!(w[t] > 1e-10)) w[t] = 1e-10; /* PSD_EIGENVALUE_FLOOR correlation.rs:20 */
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < k; ++j) {
      double acc = 0.0;
      for (uint32_t t = 0; t < k; ++t) acc += v[t * k + i] * w[t] * v[t * k + j];
      rep[j * k + i] = acc;
    }
  double d[MCPO_MAX_K];
  for (uint32_t i = 0; i < k; ++i) d[i] = rep[i * k + i] > 0.0 ? sqrt(rep[i * k + i]) : 1.0;
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t i = 0; i < k; ++i) rep[j * k + i] /= d[i] * d[j];
  for (uint32_t i = 0; i < k; ++i) rep[i * k + i] = 1.0;
  memcpy(m, rep, sizeof(double) * k * k);
  return mcpo_chol_colmajor(m, k);
}

uint32_t mcpo_fixed_level_next(const double* probs, uint32_t* counts, uint32_t n_levels);
static uint32_t fixed_level_next(const double* probs, uint32_t* counts, uint32_t n_levels) {
  return mcpo_fixed_level_next(probs, counts, n_levels);
}

/* data_gen.rs:943-961 */
uint32_t mcpo_fixed_level_next(const double* probs, uint32_t* counts, uint32_t n_levels) {
  double total = 0.0;
  uint32_t assigned = 0;
  for (uint32_t g = 0; g < n_levels; ++g) {
    total += probs[g];
    assigned += counts[g];
  }
  double t = (double)(assigned + 1);
  uint32_t best = 0;
  double best_def = -INFINITY;
  for (uint32_t g = 0; g < n_levels; ++g) {
    double deficit = (probs[g] / total) * t - (double)counts[g];
    if (deficit > best_def) {
      best_def = deficit;
      best = g;
    }
  }
  counts[best] += 1;
  return best;
}

/* distributions.rs:78-97 */
static double sample_t(mcpo_simrng* rng, double df) {
  if (!isfinite(df) || df <= 0.0) return (double)mcpo_simrng_next_normal(rng);
  double z = (double)mcpo_simrng_next_normal(rng);
  int64_t dfi = (int64_t)llround(df);
  if (dfi < 1) dfi = 1;
  double chi2 = 0.0;
  for (int64_t i = 0; i < dfi; ++i) {
    double g = (double)mcpo_simrng_next_normal(rng);
    chi2 += g * g;
  }
  double denom = sqrt(chi2 / df);
  return denom <= 0.0 ? z : z / denom;
}

/* data_gen.rs:1289-1320 */
static double draw_residual(mcpo_simrng* rng, int dist, double resid_df) {
  switch (dist) {
    case MCP_RESID_HIGH_KURTOSIS: {
      double df = resid_df > 3.0 ? resid_df : 3.0;
      double scale = 1.0 / sqrt(df / (df - 2.0));
      return sample_t(rng, df) * scale;
    }
    case MCP_RESID_RIGHT_SKEWED:
    case MCP_RESID_LEFT_SKEWED: {
      double df = resid_df > 3.0 ? resid_df : 3.0;
      int64_t dfi = (int64_t)llround(df);
      if (dfi < 1) dfi = 1;
      double chi2 = 0.0;
      for (int64_t i = 0; i < dfi; ++i) {
        double g = (double)mcpo_simrng_next_normal(rng);
        chi2 += g * g;
      }
      double centered = (chi2 - df) * (1.0 / sqrt(2.0 * df));
      return dist == MCP_RESID_LEFT_SKEWED ? -centered : centered;
    }
    case MCP_RESID_UNIFORM:
      return (2.0 * (double)mcpo_simrng_next_uniform(rng) - 1.0) * SQRT3;
    default:
      return (double)mcpo_simrng_next_normal(rng);
  }
}

static double ln_owned_floor(double u) {
  /* glmm ln_owned carries a 2^-11 input floor (data_gen.rs:1137-1141) */
  const double fl = 1.0 / 2048.0;
  return log(u > fl ? u : fl);
}

/* data_gen.rs:1253-1277: type-7 (linear interpolation) quantile of the sorted uploaded column `col` */
static double empirical_quantile(const mcp_spec* spec, uint32_t col, double u) {
  uint32_t u_rows = spec->upload_normal_rows, n_cols = spec->upload_normal_cols;
  if (u_rows == 0) return 0.0;
  double pos = u * (double)(u_rows - 1);
  size_t lo = (size_t)floor(pos);
  size_t hi = lo + 1 < u_rows - 1 ? lo + 1 : u_rows - 1;
  double frac = pos - (double)lo;
  double val_lo = spec->upload_normal[lo * n_cols + col], val_hi = spec->upload_normal[hi * n_cols + col];
  return val_lo + frac * (val_hi - val_lo);
}

/* data_gen.rs:1190-1250 (column twin of apply_marginal :1106) */
static int apply_marginal_column(double* col, uint32_t n, int dist, double param, const mcp_spec* spec,
                                 uint32_t var_idx) {
  switch (dist) {
    case MCP_DIST_UPLOADED_DATA: /* NORTA: u = Phi(z) clamped to [0,1], then the inverse empirical CDF */
      for (uint32_t i = 0; i < n; ++i) {
        double u = mcpo_phi(col[i]);
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
        col[i] = empirical_quantile(spec, var_idx, u);
      }
      return 0;
    case MCP_DIST_NORMAL:
    case MCP_DIST_UPLOADED_FACTOR:
      return 0;
    case MCP_DIST_BINARY:
    case MCP_DIST_UPLOADED_BINARY:
      for (uint32_t i = 0; i < n; ++i) col[i] = mcpo_phi(col[i]) < 1.0 - param ? 0.0 : 1.0;
      return 0;
    case MCP_DIST_UNIFORM:
      for (uint32_t i = 0; i < n; ++i) col[i] = -SQRT3 + (SQRT3 - (-SQRT3)) * mcpo_phi(col[i]);
      return 0;
    case MCP_DIST_RIGHT_SKEWED:
      for (uint32_t i = 0; i < n; ++i) {
        double e = fmin(-ln_owned_floor(mcpo_phi(-col[i])), EXP_CAP);
        col[i] = (e - EXP_CENSORED_MEAN) / EXP_CENSORED_STD;
      }
      return 0;
    case MCP_DIST_LEFT_SKEWED:
      for (uint32_t i = 0; i < n; ++i) {
        double e = fmin(-ln_owned_floor(mcpo_phi(col[i])), EXP_CAP);
        col[i] = (EXP_CENSORED_MEAN - e) / EXP_CENSORED_STD;
      }
      return 0;
    case MCP_DIST_HIGH_KURTOSIS:
      t3_build();
      for (uint32_t i = 0; i < n; ++i) col[i] = t3_lookup(mcpo_phi(col[i]));
      return 0;
    default:
      return MCP_ERR_UNSUPPORTED; /* uploaded data (NORTA) */
  }
}

static int scenario_is_optimistic(const mcp_scenario* s) {
  /* spec.rs:229-236 */
  return s->heterogeneity == 0.0 && s->heteroskedasticity_ratio == 1.0 && s->correlation_noise_sd == 0.0 &&
         s->distribution_change_prob == 0.0 && s->residual_change_prob == 0.0 && !s->has_lme;
}

/* spec.rs:436-515 */
void mcpo_compute_het_coeffs(const mcp_spec* spec, double* lp_mean, double* lp_std, double* col_mean,
                             double* col_std) {
  uint32_t k = spec->n_non_factor, nfd = spec->n_factor_dummies, p = 1 + k + nfd;
  for (uint32_t j = 0; j < p; ++j) col_mean[j] = col_std[j] = 0.0;
  col_mean[0] = 1.0;
  for (uint32_t j = 0; j < k; ++j) {
    if (spec->var_types[j] == MCP_DIST_BINARY) {
      double pp = j < spec->n_var_params ? spec->var_params[j] : 0.5;
      col_mean[1 + j] = pp;
      col_std[1 + j] = sqrt(fmax(pp * (1.0 - pp), 0.0));
    } else {
      col_mean[1 + j] = 0.0;
      col_std[1 + j] = 1.0;
    }
  }
  uint32_t col = 1 + k, off = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t nl = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    for (uint32_t d = 0; d + 1 < nl; ++d) {
      double pp = (off + d + 1 < spec->n_factor_proportions) ? spec->factor_proportions[off + d + 1] : 0.0;
      col_mean[col + d] = pp;
      col_std[col + d] = sqrt(fmax(pp * (1.0 - pp), 0.0));
    }
    col += nl > 0 ? nl - 1 : 0;
    off += nl;
  }
  double m = 0.0, v = 0.0;
  for (uint32_t j = 0; j < p; ++j) {
    double b = j < spec->n_effect_sizes ? spec->effect_sizes[j] : 0.0;
    m += b * col_mean[j];
  }
  for (uint32_t j = 0; j < p; ++j) {
    double b = j < spec->n_effect_sizes ? spec->effect_sizes[j] : 0.0;
    v += b * b * col_std[j] * col_std[j];
  }
  for (uint32_t j = 0; j < k; ++j)
    for (uint32_t t = j + 1; t < k; ++t) {
      double bj = spec->effect_sizes[1 + j], bt = spec->effect_sizes[1 + t];
      v += 2.0 * bj * bt * spec->correlation[t * k + j] * col_std[1 + j] * col_std[1 + t];
    }
  *lp_mean = m;
  *lp_std = sqrt(fmax(v, 0.0));
}

int mcpo_spec_supported(const mcp_spec* spec) {
  if (spec->n_non_factor > MCPO_MAX_K) return 0;
  if (spec->cluster.present) { /* capacity of the shared POD layout */
    const mcp_cluster* c = &spec->cluster;
    if (c->slopes.n_slopes > MCP_MAX_RE_SLOPES || c->n_extra_groupings > MCP_MAX_EXTRA_GROUPINGS) return 0;
    for (uint32_t g = 0; g < c->n_extra_groupings; ++g) {
      if (c->extra[g].slopes.n_slopes > MCP_MAX_RE_SLOPES) return 0;
      if (c->extra[g].relation == MCP_GROUPING_CROSSED && c->sizing != MCP_CLUSTER_FIXED_CLUSTERS) return 0; /* validate(): crossed x FixedSize */
    }
  }
  /* uploaded frames: shapes must be consistent (contract_adapter.rs:356-469 builds them) */
  if (spec->upload_normal_rows && spec->upload_normal_cols != spec->n_non_factor) return 0;
  if (spec->n_bootstrap_frame_map) {
    if (spec->n_bootstrap_frame_map != spec->n_non_factor + spec->n_factors || spec->upload_data_rows == 0) return 0;
    for (uint32_t j = 0; j < spec->n_bootstrap_frame_map; ++j)
      if (spec->bootstrap_frame_map[j] >= (int32_t)spec->upload_data_cols) return 0;
  }
  return 1;
}

uint32_t mcpo_cluster_of_row(const mcp_cluster* c, uint32_t i) {
  uint32_t v = c->sizing_value ? c->sizing_value : 1;
  return c->sizing == MCP_CLUSTER_FIXED_CLUSTERS ? i % v : i / v;
}
uint32_t mcpo_n_clusters_at(const mcp_cluster* c, uint32_t n) {
  uint32_t v = c->sizing_value ? c->sizing_value : 1;
  return c->sizing == MCP_CLUSTER_FIXED_CLUSTERS ? v : n / v;
}

/* extra_n_levels_at (generation.rs:302-317) */
uint32_t mcpo_extra_n_levels_at(const mcp_cluster* c, uint32_t g, uint32_t n) {
  const mcp_grouping* e = &c->extra[g];
  uint32_t cnt = e->count ? e->count : 1;
  if (e->relation == MCP_GROUPING_CROSSED) return cnt;
  uint32_t v = c->sizing_value ? c->sizing_value : 1;
  uint32_t parents = c->sizing == MCP_CLUSTER_FIXED_CLUSTERS ? v : (n + v - 1) / v;
  return parents * cnt;
}
/* extra_level_of_row (generation.rs:402-427): one lexicographic stride chain, primary fastest */
uint32_t mcpo_extra_level_of_row(const mcp_cluster* c, uint32_t g, uint32_t i) {
  const mcp_grouping* e = &c->extra[g];
  uint32_t bl = e->count ? e->count : 1;
  uint32_t v = c->sizing_value ? c->sizing_value : 1;
  if (c->sizing == MCP_CLUSTER_FIXED_CLUSTERS) {
    uint64_t stride = v;
    for (uint32_t h = 0; h < g; ++h) stride *= c->extra[h].count ? c->extra[h].count : 1;
    uint32_t within = (uint32_t)((i / stride) % bl);
    return e->relation == MCP_GROUPING_CROSSED ? within : (i % v) * bl + within;
  }
  return (i / v) * bl + (i % v) % bl;
}

/* u_k = sum_{j<=k} L[k][j] z_j for k >= 1, with L = chol(diag(tau) R diag(tau)) (data_gen.rs:163-214, 242-283) */
void mcpo_slope_cholesky(const mcp_re_slopes* sl, double tau0, double* l /* q*q row-major */) {
  uint32_t q = 1 + sl->n_slopes;
  double tv[1 + MCP_MAX_RE_SLOPES], d[(1 + MCP_MAX_RE_SLOPES) * (1 + MCP_MAX_RE_SLOPES)];
  tv[0] = tau0;
  for (uint32_t k = 0; k < sl->n_slopes; ++k) tv[k + 1] = sqrt(fmax(sl->variance[k], 0.0));
  for (uint32_t i = 0; i < q; ++i)
    for (uint32_t j = 0; j < q; ++j) d[i * q + j] = tv[i] * sl->re_corr[i * q + j] * tv[j];
  /* chol_lower (data_gen.rs:55-80): row-major lower factor, non-positive pivots give a zero column */
  for (uint32_t i = 0; i < q * q; ++i) l[i] = 0.0;
  for (uint32_t j = 0; j < q; ++j) {
    double sum = d[j * q + j];
    for (uint32_t t = 0; t < j; ++t) sum -= l[j * q + t] * l[j * q + t];
    double diag = sum > 0.0 ? sqrt(sum) : 0.0;
    l[j * q + j] = diag;
    for (uint32_t i = j + 1; i < q; ++i) {
      double v = d[i * q + j];
      for (uint32_t t = 0; t < j; ++t) v -= l[i * q + t] * l[j * q + t];
      l[i * q + j] = diag > 0.0 ? v / diag : 0.0;
    }
  }
}

/* generate_sim_data, data_gen.rs:87-925.  ws buffers sized for (max_n, p). */
int mcpo_generate_ws(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n, mcpo_ws* ws) {
  const uint32_t k = spec->n_non_factor, nfd = spec->n_factor_dummies, nf = spec->n_factors;
  const uint32_t p = 1 + k + nfd + spec->n_interactions;
  float* x = ws->x;
  float* y = ws->y;
  if (!mcpo_spec_supported(spec)) return MCP_ERR_UNSUPPORTED;

  /* 1. populate_design (data_gen.rs:990-1093) */
  double L[MCPO_MAX_K * MCPO_MAX_K];
  int var_types[MCPO_MAX_K];
  int resid_dist = spec->residual_dist;
  double resid_df = spec->scenario.residual_df;
  double tau_sq = spec->cluster.present ? spec->cluster.tau_squared : 0.0;
  double extra_tau_sq[MCP_MAX_EXTRA_GROUPINGS] = {0.0, 0.0};
  for (uint32_t g = 0; spec->cluster.present && g < spec->cluster.n_extra_groupings; ++g)
    extra_tau_sq[g] = spec->cluster.extra[g].tau_squared;
  if (scenario_is_optimistic(&spec->scenario)) {
    memcpy(L, spec->correlation, sizeof(double) * k * k);
    if (k && mcpo_chol_colmajor(L, k)) return MCP_ERR_CORRELATION_NOT_PSD;
    for (uint32_t j = 0; j < k; ++j) var_types[j] = spec->var_types[j];
  } else {
    mcpo_simrng srng;
    mcpo_simrng_new(&srng, base_seed, sim_id ^ STREAM_TAG_SCENARIO);
    const mcp_scenario* sc = &spec->scenario;
    double sigma[MCPO_MAX_K * MCPO_MAX_K];
    if (sc->correlation_noise_sd == 0.0) {
      memcpy(sigma, spec->correlation, sizeof(double) * k * k);
    } else { /* scenarios.rs:67-122 */
      double noise[MCPO_MAX_K * MCPO_MAX_K];
      for (uint32_t j = 0; j < k; ++j)
        for (uint32_t i = 0; i < k; ++i) noise[j * k + i] = (double)mcpo_simrng_next_normal(&srng) * sc->correlation_noise_sd;
      for (uint32_t j = 0; j < k; ++j)
        for (uint32_t i = 0; i < k; ++i) {
          double sym = 0.5 * (noise[j * k + i] + noise[i * k + j]);
          double v = spec->correlation[j * k + i] + sym;
          sigma[j * k + i] = v < -0.8 ? -0.8 : (v > 0.8 ? 0.8 : v);
        }
      for (uint32_t i = 0; i < k; ++i) sigma[i * k + i] = 1.0;
    }
    memcpy(L, sigma, sizeof(double) * k * k);
    if (k && psd_repair_and_factor(L, k)) return MCP_ERR_CORRELATION_NOT_PSD;
    /* perturb_var_types scenarios.rs:127-157 */
    for (uint32_t j = 0; j < k; ++j) var_types[j] = spec->var_types[j];
    if (sc->distribution_change_prob > 0.0 && sc->n_new_distributions > 0) {
      for (uint32_t j = 0; j < k; ++j) {
        double u = (double)mcpo_simrng_next_uniform(&srng);
        int pinned = j < spec->n_var_pinned ? spec->var_pinned[j] : 0;
        if (var_types[j] == MCP_DIST_NORMAL && !pinned && u < sc->distribution_change_prob) {
          size_t pick = (size_t)((double)mcpo_simrng_next_uniform(&srng) * (double)sc->n_new_distributions);
          if (pick > sc->n_new_distributions - 1) pick = sc->n_new_distributions - 1;
          var_types[j] = sc->new_distributions[pick];
        }
      }
    }
    /* pick_residual scenarios.rs:163-186 */
    if (sc->residual_change_prob > 0.0 && sc->n_residual_dists > 0) {
      double u = (double)mcpo_simrng_next_uniform(&srng);
      if (u < sc->residual_change_prob && spec->residual_dist == MCP_RESID_NORMAL && !spec->residual_pinned) {
        size_t pick = (size_t)((double)mcpo_simrng_next_uniform(&srng) * (double)sc->n_residual_dists);
        if (pick > sc->n_residual_dists - 1) pick = sc->n_residual_dists - 1;
        resid_dist = sc->residual_dists[pick];
      }
    }
    if (sc->has_lme && sc->lme_icc_noise_sd > 0.0) { /* data_gen.rs:1069-1076 */
      double jit = (double)mcpo_simrng_next_normal(&srng) * sc->lme_icc_noise_sd;
      tau_sq = fmax(tau_sq + jit, 0.0);
      /* one independent jitter per extra grouping, declaration order (data_gen.rs:1077-1092) */
      for (uint32_t g = 0; spec->cluster.present && g < spec->cluster.n_extra_groupings; ++g) {
        double jg = (double)mcpo_simrng_next_normal(&srng) * sc->lme_icc_noise_sd;
        extra_tau_sq[g] = fmax(spec->cluster.extra[g].tau_squared + jg, 0.0);
      }
    }
  }

  /* 2. per-sim stream; 2a. cluster random intercepts (data_gen.rs:140-161) */
  mcpo_simrng rng;
  mcpo_simrng_new(&rng, base_seed, sim_id);
  uint32_t n_clusters = 0;
  if (spec->cluster.present) {
    double tau = sqrt(fmax(tau_sq, 0.0));
    int re_dist = spec->scenario.has_lme ? spec->scenario.lme_random_effect_dist : MCP_RESID_NORMAL;
    double re_df = spec->scenario.has_lme ? spec->scenario.lme_random_effect_df : 0.0;
    n_clusters = mcpo_n_clusters_at(&spec->cluster, max_n);
    if (n_clusters > ws->cap_clusters) return MCP_ERR_INVALID_SPEC;
    for (uint32_t c = 0; c < n_clusters; ++c) ws->cluster_u[c] = (float)(draw_residual(&rng, re_dist, re_df) * tau);
    const mcp_cluster* cl = &spec->cluster;
    /* 2a'. primary random slopes: z0 recovered from the f32 intercept draw (data_gen.rs:163-214) */
    if (cl->slopes.n_slopes) {
      uint32_t nsl = cl->slopes.n_slopes, q = 1 + nsl;
      double l[(1 + MCP_MAX_RE_SLOPES) * (1 + MCP_MAX_RE_SLOPES)], z[1 + MCP_MAX_RE_SLOPES];
      mcpo_slope_cholesky(&cl->slopes, tau, l);
      for (uint32_t c = 0; c < n_clusters; ++c) {
        z[0] = tau > 0.0 ? (double)ws->cluster_u[c] / tau : 0.0;
        for (uint32_t j = 1; j < q; ++j) z[j] = draw_residual(&rng, re_dist, re_df);
        for (uint32_t k = 1; k < q; ++k) {
          double uk = 0.0;
          for (uint32_t j = 0; j <= k; ++j) uk += l[k * q + j] * z[j];
          ws->slope_u[c * nsl + (k - 1)] = (float)uk;
        }
      }
    }
    /* 2b. extra-grouping intercepts, then 2b'. their slopes -- all intercepts first (data_gen.rs:216-283) */
    for (uint32_t g = 0; g < cl->n_extra_groupings; ++g) {
      double tau_g = sqrt(fmax(extra_tau_sq[g], 0.0));
      uint32_t qg = 1 + cl->extra[g].slopes.n_slopes, nl = mcpo_extra_n_levels_at(cl, g, max_n);
      if ((size_t)nl * qg > ws->cap_extra[g]) return MCP_ERR_INVALID_SPEC;
      for (uint32_t c = 0; c < nl; ++c) ws->extra_u[g][c * qg] = (float)(draw_residual(&rng, re_dist, re_df) * tau_g);
    }
    for (uint32_t g = 0; g < cl->n_extra_groupings; ++g) {
      const mcp_re_slopes* sl = &cl->extra[g].slopes;
      if (!sl->n_slopes) continue;
      double tau0 = sqrt(fmax(extra_tau_sq[g], 0.0));
      uint32_t qg = 1 + sl->n_slopes, nl = mcpo_extra_n_levels_at(cl, g, max_n);
      double l[(1 + MCP_MAX_RE_SLOPES) * (1 + MCP_MAX_RE_SLOPES)], z[1 + MCP_MAX_RE_SLOPES];
      mcpo_slope_cholesky(sl, tau0, l);
      for (uint32_t c = 0; c < nl; ++c) {
        z[0] = tau0 > 0.0 ? (double)ws->extra_u[g][c * qg] / tau0 : 0.0;
        for (uint32_t j = 1; j < qg; ++j) z[j] = draw_residual(&rng, re_dist, re_df);
        for (uint32_t d = 1; d < qg; ++d) {
          double ud = 0.0;
          for (uint32_t j = 0; j <= d; ++j) ud += l[d * qg + j] * z[j];
          ws->extra_u[g][c * qg + d] = (float)ud;
        }
      }
    }
  }

  for (uint32_t i = 0; i < max_n; ++i) x[i] = 1.0f; /* column 0 */

  /* (a) continuous block: planar normals then L.z mix (data_gen.rs:316-363) */
  for (uint32_t j = 0; j < k; ++j) mcpo_fill_normal_column(rng.key, CLASS_XNORM, j, ws->z + (size_t)j * max_n, max_n);
  for (uint32_t j = 0; j < k; ++j) {
    double* acc = ws->acc;
    for (uint32_t i = 0; i < max_n; ++i) acc[i] = 0.0;
    for (uint32_t t = 0; t <= j; ++t) {
      double l_jt = L[t * k + j];
      const float* zt = ws->z + (size_t)t * max_n;
      for (uint32_t i = 0; i < max_n; ++i) acc[i] += l_jt * (double)zt[i];
    }
    float* xc = x + (size_t)(1 + j) * max_n;
    if (var_types[j] == MCP_DIST_NORMAL) {
      for (uint32_t i = 0; i < max_n; ++i) xc[i] = (float)acc[i];
    } else {
      double param = j < spec->n_var_params ? spec->var_params[j] : 0.5;
      for (uint32_t i = 0; i < max_n; ++i) acc[i] = (double)(float)acc[i];
      int rc = apply_marginal_column(acc, max_n, var_types[j], param, spec, j);
      if (rc) return rc;
      for (uint32_t i = 0; i < max_n; ++i) xc[i] = (float)acc[i];
    }
  }

  /* (c) residual column(s) (data_gen.rs:365-503) */
  float* resid = ws->resid;
  int probit = spec->outcome_kind == MCP_OUTCOME_BINARY && spec->link == MCP_LINK_PROBIT;
  if (spec->outcome_kind == MCP_OUTCOME_BINARY && !probit) {
    mcpo_fill_uniform_column(rng.key, CLASS_RESID, 0, resid, max_n);
  } else if (spec->outcome_kind != MCP_OUTCOME_CONTINUOUS || resid_dist == MCP_RESID_NORMAL) {
    mcpo_fill_normal_column(rng.key, CLASS_RESID, 0, resid, max_n);
  } else if (resid_dist == MCP_RESID_HIGH_KURTOSIS) {
    double df = fmax(resid_df, 3.0);
    int64_t dfl = (int64_t)llround(df);
    uint32_t dfi = (uint32_t)(dfl < 1 ? 1 : dfl);
    double scale = 1.0 / sqrt(df / (df - 2.0));
    mcpo_fill_normal_column(rng.key, CLASS_RESID, 0, resid, max_n);
    for (uint32_t i = 0; i < max_n; ++i) ws->acc[i] = 0.0;
    for (uint32_t s = 0; s < dfi; ++s) {
      mcpo_fill_normal_column(rng.key, CLASS_RESID, 1 + s, ws->resid_g, max_n);
      for (uint32_t i = 0; i < max_n; ++i) {
        double g = (double)ws->resid_g[i];
        ws->acc[i] += g * g;
      }
    }
    for (uint32_t i = 0; i < max_n; ++i) {
      double z = (double)resid[i], denom = sqrt(ws->acc[i] / df);
      double t = denom <= 0.0 ? z : z / denom;
      resid[i] = (float)(t * scale);
    }
  } else if (resid_dist == MCP_RESID_RIGHT_SKEWED || resid_dist == MCP_RESID_LEFT_SKEWED) {
    double df = fmax(resid_df, 3.0);
    int64_t dfl = (int64_t)llround(df);
    uint32_t dfi = (uint32_t)(dfl < 1 ? 1 : dfl);
    double scale = 1.0 / sqrt(2.0 * df);
    double sign = resid_dist == MCP_RESID_LEFT_SKEWED ? -1.0 : 1.0;
    for (uint32_t i = 0; i < max_n; ++i) ws->acc[i] = 0.0;
    for (uint32_t s = 0; s < dfi; ++s) {
      mcpo_fill_normal_column(rng.key, CLASS_RESID, s, ws->resid_g, max_n);
      for (uint32_t i = 0; i < max_n; ++i) {
        double g = (double)ws->resid_g[i];
        ws->acc[i] += g * g;
      }
    }
    for (uint32_t i = 0; i < max_n; ++i) resid[i] = (float)(sign * (ws->acc[i] - df) * scale);
  } else { /* Uniform */
    mcpo_fill_uniform_column(rng.key, CLASS_RESID, 0, resid, max_n);
    for (uint32_t i = 0; i < max_n; ++i) resid[i] = (float)((2.0 * (double)resid[i] - 1.0) * SQRT3);
  }

  /* 3. row loop: factors, broadcast, interactions (data_gen.rs:509-650) */
  uint32_t counts[MCPO_MAX_LEVELS];
  memset(counts, 0, sizeof counts);
  if (spec->n_factor_proportions > MCPO_MAX_LEVELS) return MCP_ERR_UNSUPPORTED;
  for (uint32_t i = 0; i < max_n; ++i) {
    uint32_t col_off = 1 + k, prop_off = 0;
    for (uint32_t f = 0; f < nf; ++f) {
      uint32_t nl = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
      if (nl == 0) continue;
      const double* probs = spec->factor_proportions + prop_off;
      int sampled = spec->scenario.sampled_factor_proportions;
      if (f < spec->n_factor_sampled && spec->factor_sampled[f] >= 0) sampled = spec->factor_sampled[f];
      uint32_t lvl = sampled ? mcpo_simrng_next_categorical(&rng, probs, nl) : fixed_level_next(probs, counts + prop_off, nl);
      for (uint32_t d = 0; d + 1 < nl; ++d) x[(size_t)(col_off + d) * max_n + i] = 0.0f;
      if (lvl > 0 && lvl - 1 < nl - 1) x[(size_t)(col_off + lvl - 1) * max_n + i] = 1.0f;
      prop_off += nl;
      col_off += nl - 1;
    }
    if (spec->n_bootstrap_frame_map) { /* strict mode: one source row per design row (data_gen.rs:552-604) */
      uint32_t u_rows = spec->upload_data_rows, ncf = spec->upload_data_cols;
      size_t r = (size_t)((double)mcpo_simrng_next_uniform(&rng) * (double)u_rows);
      if (r > u_rows - 1) r = u_rows - 1;
      for (uint32_t j = 0; j < k; ++j) {
        int32_t fc = spec->bootstrap_frame_map[j];
        if (fc < 0) continue;
        double v = spec->upload_data[r * ncf + (uint32_t)fc];
        if (spec->var_types[j] == MCP_DIST_UPLOADED_BINARY) { /* stored = x - p; recover 0/1 */
          double pp = j < spec->n_var_params ? spec->var_params[j] : 0.0;
          v = (v + pp >= 0.5) ? 1.0 : 0.0;
        }
        x[(size_t)(1 + j) * max_n + i] = (float)v;
      }
      uint32_t start = 1 + k;
      for (uint32_t f = 0; f < nf; ++f) {
        uint32_t nl = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
        uint32_t nd = nl ? nl - 1 : 0;
        int32_t fc = spec->bootstrap_frame_map[k + f];
        if (fc >= 0) {
          double lv = round(spec->upload_data[r * ncf + (uint32_t)fc]);
          size_t level = lv > 0.0 ? (size_t)lv : 0;
          for (uint32_t d = 0; d < nd; ++d) x[(size_t)(start + d) * max_n + i] = 0.0f;
          if (level > 0 && level - 1 < nd) x[(size_t)(start + level - 1) * max_n + i] = 1.0f;
        }
        start += nd;
      }
    }
    if (spec->n_between_var_indices && spec->cluster.present) { /* data_gen.rs:619-636 */
      uint32_t v = spec->cluster.sizing_value ? spec->cluster.sizing_value : 1;
      uint32_t rep = spec->cluster.sizing == MCP_CLUSTER_FIXED_CLUSTERS ? i % v : i - (i % v);
      if (i != rep)
        for (uint32_t b = 0; b < spec->n_between_var_indices; ++b) {
          uint32_t c = spec->between_var_indices[b];
          x[(size_t)c * max_n + i] = x[(size_t)c * max_n + rep];
        }
    }
    for (uint32_t j = 0; j < spec->n_interactions; ++j) {
      double prod = 1.0;
      for (uint32_t c = spec->interaction_offsets[j]; c < spec->interaction_offsets[j + 1]; ++c)
        prod *= (double)x[(size_t)spec->interaction_components[c] * max_n + i];
      x[(size_t)(col_off + j) * max_n + i] = (float)prod;
    }
  }

  /* 4. y assembly (data_gen.rs:652-922) */
  double het_total = fmax(spec->scenario.heterogeneity, 0.0);
  int use_het = het_total > FLOAT_NEAR_ZERO;
  double beta_eff[MCPO_MAX_P];
  if (p > MCPO_MAX_P) return MCP_ERR_UNSUPPORTED;
  if (use_het) {
    mcpo_simrng hrng;
    mcpo_simrng_new(&hrng, base_seed, sim_id ^ STREAM_TAG_HET);
    double int_sd = spec->outcome_kind == MCP_OUTCOME_CONTINUOUS ? 0.0 : het_total;
    for (uint32_t j = 0; j < p; ++j) {
      double bj = spec->effect_sizes[j];
      double sj = j == 0 ? int_sd : het_total * fabs(bj);
      double raw = bj + (double)mcpo_simrng_next_normal(&hrng) * sj;
      if (j == 0 || bj == 0.0) beta_eff[j] = raw;
      else if (bj > 0.0) beta_eff[j] = raw > 0.0 ? raw : 0.0; /* f64::max(0.0) */
      else beta_eff[j] = raw < 0.0 ? raw : 0.0;
    }
  }
  double lambda = fmax(spec->scenario.heteroskedasticity_ratio, 1.0);
  double gamma = log(lambda) / 4.0;
  double hsk_center = 0.0, hsk_std = 0.0;
  if (gamma > FLOAT_NEAR_ZERO && spec->outcome_kind == MCP_OUTCOME_CONTINUOUS) {
    double cm[MCPO_MAX_P], cs[MCPO_MAX_P], lm, ls;
    mcpo_compute_het_coeffs(spec, &lm, &ls, cm, cs);
    if (spec->heteroskedasticity_driver < 0) {
      hsk_center = lm;
      hsk_std = ls;
    } else if ((uint32_t)spec->heteroskedasticity_driver < 1 + k + nfd) {
      hsk_center = cm[spec->heteroskedasticity_driver];
      hsk_std = cs[spec->heteroskedasticity_driver];
    }
  }
  int apply_hsk = spec->outcome_kind == MCP_OUTCOME_CONTINUOUS && gamma > FLOAT_NEAR_ZERO && hsk_std > FLOAT_NEAR_ZERO;
  double hsk_inv_std = apply_hsk ? 1.0 / hsk_std : 0.0;
  double hsk_inv_norm = exp(-gamma * gamma / 2.0);

  for (uint32_t i = 0; i < max_n; ++i) {
    double lp_clean = 0.0, lp_eff = 0.0;
    for (uint32_t j = 0; j < p; ++j) {
      double xij = (double)x[(size_t)j * max_n + i];
      lp_clean += xij * spec->effect_sizes[j];
      if (use_het) lp_eff += xij * beta_eff[j];
    }
    double lp = use_het ? lp_eff : lp_clean;
    if (apply_hsk) {
      double drv = spec->heteroskedasticity_driver < 0 ? lp_clean
                                                       : (double)x[(size_t)spec->heteroskedasticity_driver * max_n + i];
      ws->acc[i] = gamma * ((drv - hsk_center) * hsk_inv_std);
    }
    double u_re = 0.0;
    if (spec->cluster.present) {
      uint32_t cid = mcpo_cluster_of_row(&spec->cluster, i);
      if (ws->cluster_ids) ws->cluster_ids[i] = cid;
      u_re = (double)ws->cluster_u[cid];
      const mcp_cluster* cl = &spec->cluster;
      for (uint32_t g = 0; g < cl->n_extra_groupings; ++g) { /* data_gen.rs:819-836 */
        const mcp_re_slopes* sl = &cl->extra[g].slopes;
        uint32_t qg = 1 + sl->n_slopes, lvl = mcpo_extra_level_of_row(cl, g, i);
        ws->extra_ids[g][i] = lvl;
        u_re += (double)ws->extra_u[g][lvl * qg];
        for (uint32_t d = 0; d < sl->n_slopes; ++d)
          u_re += (double)ws->extra_u[g][lvl * qg + 1 + d] * (double)x[(size_t)sl->design_cols[d] * max_n + i];
      }
      for (uint32_t k = 0; k < cl->slopes.n_slopes; ++k) /* :837-842 */
        u_re += (double)ws->slope_u[cid * cl->slopes.n_slopes + k] * (double)x[(size_t)cl->slopes.design_cols[k] * max_n + i];
    }
    if (spec->outcome_kind == MCP_OUTCOME_CONTINUOUS && !apply_hsk) y[i] = (float)(lp + (double)resid[i] + u_re);
    else ws->eta[i] = lp + u_re;
  }
  if (apply_hsk) { /* data_gen.rs:876-888 (libm exp for glmm exp_fill) */
    for (uint32_t i = 0; i < max_n; ++i) {
      double mult = exp(ws->acc[i]) * hsk_inv_norm;
      resid[i] = (float)((double)resid[i] * sqrt(mult));
    }
    for (uint32_t i = 0; i < max_n; ++i) y[i] = (float)(ws->eta[i] + (double)resid[i]);
  }
  if (spec->outcome_kind == MCP_OUTCOME_BINARY) { /* data_gen.rs:890-911 */
    for (uint32_t i = 0; i < max_n; ++i) {
      double th = ws->eta[i];
      if (!probit) th = 1.0 / (1.0 + exp(-th)); /* glmm sigmoid_fill: <= 2 ULP of this */
      y[i] = ((double)resid[i] < th) ? 1.0f : 0.0f;
    }
  } else if (spec->outcome_kind == MCP_OUTCOME_COUNT) { /* rng.rs:399-419 */
    for (uint32_t i = 0; i < max_n; ++i) {
      double lam = exp(ws->eta[i]), z = (double)resid[i], out;
      if (!(lam > 0.0)) out = 0.0;
      else if (lam <= 20.0) {
        double u = mcpo_phi(z), pk = exp(-lam), cdf = pk;
        uint32_t kk = 0;
        while (cdf < u && kk < 200) {
          kk += 1;
          pk *= lam / (double)kk;
          cdf += pk;
        }
        out = (double)kk;
      } else {
        out = fmax(floor(lam + z * sqrt(lam) + 0.5), 0.0);
      }
      y[i] = (float)out;
    }
  }
  return MCP_OK;
}

/* ---- workspace -------------------------------------------------------------- */
mcpo_ws* mcpo_ws_new(uint32_t max_n, uint32_t p, uint32_t k, uint32_t cap_clusters) {
  mcpo_ws* ws = (mcpo_ws*)calloc(1, sizeof(mcpo_ws));
  if (!ws) return NULL;
  ws->x = (float*)malloc(sizeof(float) * (size_t)max_n * p);
  ws->y = (float*)malloc(sizeof(float) * max_n);
  ws->z = (float*)malloc(sizeof(float) * (size_t)max_n * (k ? k : 1));
  ws->resid = (float*)malloc(sizeof(float) * max_n);
  ws->resid_g = (float*)malloc(sizeof(float) * max_n);
  ws->acc = (double*)malloc(sizeof(double) * max_n);
  ws->eta = (double*)malloc(sizeof(double) * max_n);
  ws->cluster_u = (float*)malloc(sizeof(float) * (cap_clusters ? cap_clusters : 1));
  ws->cluster_ids = (uint32_t*)malloc(sizeof(uint32_t) * max_n);
  ws->cap_clusters = cap_clusters;
  ws->slope_u = (float*)calloc((size_t)(cap_clusters ? cap_clusters : 1) * MCP_MAX_RE_SLOPES, sizeof(float));
  return ws;
}

/* buffers of the extra groupings (levels at max_n x q_g draws, one level id per row) */
int mcpo_ws_size_extras(mcpo_ws* ws, const mcp_spec* spec, uint32_t max_n) {
  for (uint32_t g = 0; spec->cluster.present && g < spec->cluster.n_extra_groupings && g < MCP_MAX_EXTRA_GROUPINGS; ++g) {
    size_t need = (size_t)mcpo_extra_n_levels_at(&spec->cluster, g, max_n) * (1 + MCP_MAX_RE_SLOPES);
    if (need > ws->cap_extra[g]) {
      free(ws->extra_u[g]);
      ws->extra_u[g] = (float*)calloc(need ? need : 1, sizeof(float));
      ws->cap_extra[g] = need;
    }
    if (!ws->extra_ids[g]) ws->extra_ids[g] = (uint32_t*)calloc(max_n ? max_n : 1, sizeof(uint32_t));
    if (!ws->extra_u[g] || !ws->extra_ids[g]) return MCP_ERR_DEVICE;
  }
  return MCP_OK;
}

void mcpo_ws_free(mcpo_ws* ws) {
  if (!ws) return;
  free(ws->x); free(ws->y); free(ws->z); free(ws->resid); free(ws->resid_g);
  free(ws->acc); free(ws->eta); free(ws->cluster_u); free(ws->cluster_ids); free(ws->slope_u);
  for (uint32_t g = 0; g < MCP_MAX_EXTRA_GROUPINGS; ++g) { free(ws->extra_u[g]); free(ws->extra_ids[g]); }
  free(ws);
}

int mcpo_generate_sim_data(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n, float* x_out,
                           float* y_out, uint32_t* cluster_ids_out) {
  return mcpo_generate_sim_data_ids(spec, base_seed, sim_id, max_n, x_out, y_out, cluster_ids_out, NULL);
}

int mcpo_generate_sim_data_ids(const mcp_spec* spec, uint64_t base_seed, uint64_t sim_id, uint32_t max_n, float* x_out,
                               float* y_out, uint32_t* cluster_ids_out, uint32_t* extra_ids_out) {
  uint32_t k = spec->n_non_factor;
  uint32_t p = 1 + k + spec->n_factor_dummies + spec->n_interactions;
  uint32_t ncl = spec->cluster.present ? mcpo_n_clusters_at(&spec->cluster, max_n) : 0;
  mcpo_ws* ws = mcpo_ws_new(max_n, p, k, ncl);
  if (!ws) return MCP_ERR_DEVICE;
  int rc = mcpo_ws_size_extras(ws, spec, max_n);
  if (rc == MCP_OK) rc = mcpo_generate_ws(spec, base_seed, sim_id, max_n, ws);
  if (rc == MCP_OK) {
    memcpy(x_out, ws->x, sizeof(float) * (size_t)max_n * p);
    memcpy(y_out, ws->y, sizeof(float) * max_n);
    if (cluster_ids_out && spec->cluster.present) memcpy(cluster_ids_out, ws->cluster_ids, sizeof(uint32_t) * max_n);
    for (uint32_t g = 0; extra_ids_out && spec->cluster.present && g < spec->cluster.n_extra_groupings; ++g)
      memcpy(extra_ids_out + (size_t)g * max_n, ws->extra_ids[g], sizeof(uint32_t) * max_n);
  }
  mcpo_ws_free(ws);
  return rc;
}
