/*
 * oracle/fit_glm.c -- CPU ORACLE (test infrastructure): logistic GLM by IRLS and the GLM arm of
 * run_one_sim.
 *
 * The reference delegates the fit to the third-party crate glmm 0.1.2
 * (`loop_advanced::glm_irls_fit`, Cargo.lock:2060-2063), which is NOT in the tree; only its call
 * site and contract are readable:
 *   crates/engine-core/src/batch.rs:1562-2134 (GLM arm: refit per grid N from the truth start
 *     beta0 = spec.effect_sizes :1945-1951, no step halving, BETA_CAP |beta| > 30 -> not converged
 *     :1943, :4550; outputs converged / t_sq at target rank / deviance / deviance_null;
 *     Wald z^2 vs norm_ppf^2; LRT overall = deviance_null - deviance vs chi2(P-1) :2116-2124),
 *   crates/engine-core/src/glm_dgp_tests.rs:117-180 (converges within MAX_IRLS_ITERS on
 *     well-behaved data, finite beta / z^2).
 * The algorithm restated here is the published one (McCullagh & Nelder; R `glm.fit`): Fisher
 * scoring with working response z = eta + (y - mu)/mu', w = mu'^2 / V(mu); convergence on
 * |dev - dev_old| / (|dev| + 0.1) < 1e-8, at most 25 passes (R's glm.control defaults).
 * Families (mixed_workspace::glmm_family, batch.rs:1958-1962): Binary/logit, Binary/probit
 * (mu = Phi(eta), mu' = phi(eta), eta clamped to +-qnorm(eps) like R's probit link), Count/log
 * (Poisson: mu = mu' = V = exp(eta)).
 *
 * PARITY STATUS: beta, se, z pinned to R `glm` on the reference's own bytes
 * (validation/data/glm_*.golden.rds) at 1e-4 relative (validation/tolerances.R:29-31).  The
 * iteration cap, tolerance, weight clamp and hence per-fit `converged` flags / separation
 * behaviour of glmm's kernel are NOT readable: **parity unpinned** beyond those bands
 * (SURVEY.md 8c, row "GLM fit").
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define MAX_IRLS_ITERS 25
#define IRLS_EPS 1e-8
#define BETA_CAP 30.0      /* batch.rs:4550 */
#define W_FLOOR 1e-10
#define FLOAT_NEAR_ZERO 1e-30

typedef struct glm_fit {
  int converged;
  uint32_t n_iter;
  double beta[MCPO_MAX_P];
  double L[MCPO_MAX_P * MCPO_MAX_P]; /* lower Cholesky factor of X'WX at the final beta */
  double deviance, deviance_null;
} glm_fit;

static double sigmoid_stable(double eta) {
  if (eta >= 0.0) {
    double e = exp(-eta);
    return 1.0 / (1.0 + e);
  }
  double e = exp(eta);
  return e / (1.0 + e);
}

enum { FAM_LOGIT = 0, FAM_PROBIT = 1, FAM_POISSON = 2 };

static int glm_family_of(const mcp_spec* spec) {
  if (spec->outcome_kind == MCP_OUTCOME_COUNT) return FAM_POISSON;
  return spec->link == MCP_LINK_PROBIT ? FAM_PROBIT : FAM_LOGIT;
}

/* mean, IRLS weight, working response and deviance contribution of one observation */
void mcpo_fam_eval(int fam, double eta, double yi, double* w_out, double* z_out, double* dev_out) {
  if (fam == FAM_LOGIT) {
    double pr = sigmoid_stable(eta);
    double w = pr * (1.0 - pr);
    if (w < W_FLOOR) w = W_FLOOR;
    /* binomial deviance contribution, y in {0,1}: -2 log(p) or -2 log(1-p), via log1p for stability */
    double ll = yi > 0.5 ? (eta >= 0.0 ? -log1p(exp(-eta)) : eta - log1p(exp(eta)))
                         : (eta >= 0.0 ? -eta - log1p(exp(-eta)) : -log1p(exp(eta)));
    *w_out = w;
    *z_out = eta + (yi - pr) / w;
    *dev_out = -2.0 * ll;
  } else if (fam == FAM_PROBIT) {
    const double thresh = 8.125890664701906; /* -qnorm(.Machine$double.eps) */
    double e = eta < -thresh ? -thresh : (eta > thresh ? thresh : eta);
    double mu = 0.5 * erfc(-e * M_SQRT1_2), omu = 0.5 * erfc(e * M_SQRT1_2);
    double dmu = exp(-0.5 * e * e) / sqrt(2.0 * M_PI);
    if (dmu < 2.220446049250313e-16) dmu = 2.220446049250313e-16;
    double w = dmu * dmu / (mu * omu);
    if (w < W_FLOOR) w = W_FLOOR;
    *w_out = w;
    *z_out = eta + (yi - mu) / dmu;
    *dev_out = -2.0 * (yi > 0.5 ? log(mu) : log(omu));
  } else { /* Poisson, log link */
    double mu = exp(eta);
    double w = mu < W_FLOOR ? W_FLOOR : mu;
    *w_out = w;
    *z_out = eta + (yi - mu) / w;
    *dev_out = 2.0 * ((yi > 0.0 ? yi * log(yi / mu) : 0.0) - (yi - mu));
  }
}

/* one pass at `beta`: X'WX (lower, row-major), X'Wz, deviance */
static void irls_pass(int fam, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, const double* beta,
                      double* xtwx, double* xtwz, double* dev) {
  memset(xtwx, 0, sizeof(double) * p * p);
  memset(xtwz, 0, sizeof(double) * p);
  double d = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    double eta = 0.0;
    for (uint32_t c = 0; c < p; ++c) eta += (double)x[(size_t)c * ld + i] * beta[c];
    double w, z, di;
    mcpo_fam_eval(fam, eta, (double)y[i], &w, &z, &di);
    d += di;
    for (uint32_t a = 0; a < p; ++a) {
      double wxa = w * (double)x[(size_t)a * ld + i];
      for (uint32_t b = 0; b <= a; ++b) xtwx[a * p + b] += wxa * (double)x[(size_t)b * ld + i];
      xtwz[a] += wxa * z;
    }
  }
  *dev = d;
}

static int chol_solve(const double* a, const double* rhs, uint32_t p, double* L, double* sol) {
  memset(L, 0, sizeof(double) * p * p);
  for (uint32_t j = 0; j < p; ++j) {
    double d = a[j * p + j];
    for (uint32_t t = 0; t < j; ++t) d -= L[j * p + t] * L[j * p + t];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    double ljj = sqrt(d);
    L[j * p + j] = ljj;
    for (uint32_t i = j + 1; i < p; ++i) {
      double v = a[i * p + j];
      for (uint32_t t = 0; t < j; ++t) v -= L[i * p + t] * L[j * p + t];
      L[i * p + j] = v / ljj;
    }
  }
  double z[MCPO_MAX_P];
  for (uint32_t i = 0; i < p; ++i) {
    double v = rhs[i];
    for (uint32_t t = 0; t < i; ++t) v -= L[i * p + t] * z[t];
    z[i] = v / L[i * p + i];
  }
  for (int64_t i = (int64_t)p - 1; i >= 0; --i) {
    double v = z[i];
    for (uint32_t t = (uint32_t)i + 1; t < p; ++t) v -= L[t * p + i] * sol[t];
    sol[i] = v / L[i * p + i];
  }
  return 0;
}

static void glm_irls(int fam, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, const double* beta0,
                     glm_fit* f) {
  double xtwx[MCPO_MAX_P * MCPO_MAX_P], xtwz[MCPO_MAX_P], beta[MCPO_MAX_P], nb[MCPO_MAX_P];
  memset(f, 0, sizeof *f);
  for (uint32_t c = 0; c < p; ++c) beta[c] = beta0 ? beta0[c] : 0.0;
  double sum_y = 0.0;
  for (uint32_t i = 0; i < n; ++i) sum_y += (double)y[i];
  double pbar = sum_y / (double)n;
  if (fam == FAM_POISSON) { /* intercept-only fit: mu = mean(y) */
    double dn = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
      double yi = (double)y[i];
      dn += 2.0 * ((yi > 0.0 && pbar > 0.0 ? yi * log(yi / pbar) : 0.0) - (yi - pbar));
    }
    f->deviance_null = dn;
  } else {
    f->deviance_null = (pbar > 0.0 && pbar < 1.0)
                           ? -2.0 * (sum_y * log(pbar) + ((double)n - sum_y) * log(1.0 - pbar))
                           : 0.0;
  }
  double dev_prev = INFINITY, dev = NAN;
  int ok = 0;
  uint32_t it = 0;
  for (it = 0; it < MAX_IRLS_ITERS; ++it) {
    irls_pass(fam, x, y, n, ld, p, beta, xtwx, xtwz, &dev);
    if (!isfinite(dev)) break;
    if (it > 0 && fabs(dev - dev_prev) / (fabs(dev) + 0.1) < IRLS_EPS) {
      /* converged at `beta`; factor X'WX at the final beta for the Wald variances */
      if (chol_solve(xtwx, xtwz, p, f->L, nb) == 0) ok = 1;
      break;
    }
    if (chol_solve(xtwx, xtwz, p, f->L, nb) != 0) break;
    int capped = 0;
    for (uint32_t c = 0; c < p; ++c)
      if (!isfinite(nb[c]) || fabs(nb[c]) > BETA_CAP) capped = 1;
    if (capped) break;
    memcpy(beta, nb, sizeof(double) * p);
    dev_prev = dev;
  }
  f->n_iter = it + 1;
  f->converged = ok;
  f->deviance = dev;
  memcpy(f->beta, beta, sizeof(double) * p);
}

/* variance of c'beta with cov = (X'WX)^-1 = L^-T L^-1 (dispersion 1) */
static double glm_var(const glm_fit* f, uint32_t p, const double* c) {
  double u[MCPO_MAX_P], nq = 0.0;
  for (uint32_t i = 0; i < p; ++i) {
    double acc = c[i];
    for (uint32_t t = 0; t < i; ++t) acc -= f->L[i * p + t] * u[t];
    u[i] = acc / f->L[i * p + i];
    nq += u[i] * u[i];
  }
  return nq;
}

/* Wald z^2 of c'beta */
static double glm_wald_sq(const glm_fit* f, uint32_t p, const double* c) {
  double est = 0.0;
  for (uint32_t j = 0; j < p; ++j) est += c[j] * f->beta[j];
  double nq = glm_var(f, p, c);
  if (nq > FLOAT_NEAR_ZERO && isfinite(nq)) return est * est / nq;
  return NAN;
}

/* fit provided bytes (introspect::fit_provided_data twin for the logistic arm) */
int mcpo_glm_fit_family(int family, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                        const double* beta_start, double* beta, double* se, double* z_sq, double* deviance,
                        double* deviance_null, uint8_t* converged, uint32_t* n_iter) {
  if (p > MCPO_MAX_P || family < FAM_LOGIT || family > FAM_POISSON) return MCP_ERR_UNSUPPORTED;
  glm_fit f;
  glm_irls(family, x, y, n, ld, p, beta_start, &f);
  *converged = (uint8_t)f.converged;
  *deviance = f.deviance;
  *deviance_null = f.deviance_null;
  *n_iter = f.n_iter;
  for (uint32_t j = 0; j < p; ++j) {
    double c[MCPO_MAX_P] = {0};
    c[j] = 1.0;
    beta[j] = f.converged ? f.beta[j] : NAN;
    z_sq[j] = f.converged ? glm_wald_sq(&f, p, c) : NAN;
    se[j] = f.converged ? sqrt(glm_var(&f, p, c)) : NAN;
  }
  return MCP_OK;
}

int mcpo_glm_fit(const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, const double* beta_start,
                 double* beta, double* se, double* z_sq, double* deviance, double* deviance_null, uint8_t* converged,
                 uint32_t* n_iter) {
  return mcpo_glm_fit_family(FAM_LOGIT, x, y, n, ld, p, beta_start, beta, se, z_sq, deviance, deviance_null, converged,
                             n_iter);
}

/* GLM arm of run_one_sim (batch.rs:1562-2134), unclustered logit / probit / Poisson.  Sparse-factor exclusion recounts the
 * levels of rows [0, N) at every grid point (:1836-1853); a reduced round gathers the kept design columns and the
 * kept columns' true effects as the start (:1893-1925); when attempt 0 does not converge and the feature is on,
 * the included factor with the smallest minimum level count is dropped (code 2) and the fit retried once from a
 * cold start (:1976-2030). */
int mcpo_glm_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, mcpo_red_cache* cache, uint8_t* unc, uint8_t* cor,
                 uint8_t* conv, uint8_t* overall, uint8_t* fx, double* stat_row) {
  if (spec->cluster.present) return MCP_ERR_UNSUPPORTED;            /* Laplace GLMM */
  if (spec->outcome_kind == MCP_OUTCOME_CONTINUOUS) return MCP_ERR_UNSUPPORTED;
  if (spec->outcome_kind == MCP_OUTCOME_COUNT && spec->link != MCP_LINK_CANONICAL) return MCP_ERR_UNSUPPORTED;
  if (spec->n_contrast_pairs) return MCP_ERR_UNSUPPORTED;           /* GLM has no contrasts (batch.rs:2069) */
  const int fam = glm_family_of(spec);
  const uint32_t p = 1 + spec->n_non_factor + spec->n_factor_dummies + spec->n_interactions;
  const uint32_t nt = spec->n_target_indices, nf = spec->n_factors;
  const int test_reduces = spec->n_fit_columns && spec->n_fit_columns < p;
  uint32_t prefix_counts[MCPO_MAX_LEVELS], kept[MCPO_MAX_P];
  int32_t remap[MCPO_MAX_P];
  uint8_t flags[16] = {0};
  float* xr = NULL; /* gathered kept columns, column-major with ld = max_n */
  int rc = MCP_OK;
  for (uint32_t g = 0; g < n_ns && rc == MCP_OK; ++g) {
    const uint32_t n = ns[g];
    if (nf && spec->factor_min_level_count > 0) {
      memset(prefix_counts, 0, sizeof prefix_counts);
      mcpo_update_factor_exclusions(ws->x, max_n, spec, 0, n, prefix_counts, flags);
    }
    glm_fit f;
    uint32_t p_red = p;
    int reduced = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      int any_excl = 0;
      for (uint32_t q = 0; q < nf; ++q) any_excl |= flags[q] != 0;
      reduced = any_excl || test_reduces;
      if (!reduced) {
        p_red = p;
        for (uint32_t c = 0; c < p; ++c) remap[c] = (int32_t)c;
        glm_irls(fam, ws->x, ws->y, n, max_n, p, attempt > 0 ? NULL : spec->effect_sizes, &f); /* truth start :1945-1951 */
      } else {
        double start[MCPO_MAX_P];
        p_red = mcpo_build_exclusion_remap(spec, flags, kept, remap);
        if (!xr) xr = (float*)malloc(sizeof(float) * (size_t)max_n * p);
        for (uint32_t rj = 0; rj < p_red; ++rj) {
          memcpy(xr + (size_t)rj * max_n, ws->x + (size_t)kept[rj] * max_n, sizeof(float) * n);
          start[rj] = spec->effect_sizes[kept[rj]];
        }
        glm_irls(fam, xr, ws->y, n, max_n, p_red, attempt > 0 ? NULL : start, &f);
      }
      if (f.converged || attempt == 1 || spec->factor_min_level_count == 0) break;
      int drop = mcpo_smallest_min_count_factor(spec, prefix_counts, flags);
      if (drop < 0) break;
      flags[drop] = 2;
    }
    const double* crow = crit + (size_t)g * crit_stride;
    double tsq[MCPO_MAX_P * 2];
    conv[g] = (uint8_t)f.converged;
    memset(unc + (size_t)g * nt, 0, nt);
    for (uint32_t t = 0; t < nt; ++t) {
      tsq[t] = NAN;
      int32_t r = remap[spec->target_indices[t]];
      if (f.converged && r >= 0) {
        double c[MCPO_MAX_P] = {0};
        c[r] = 1.0;
        tsq[t] = glm_wald_sq(&f, p_red, c);
        if (!isnan(tsq[t]) && tsq[t] > crow[0]) unc[(size_t)g * nt + t] = 1;
      }
    }
    if (g == 0 && stat_row) memcpy(stat_row, tsq, sizeof(double) * nt);
    if (f.converged) mcpo_apply_correction(spec->correction_method, tsq, nt, crow[0], crow + 1, cor + (size_t)g * nt);
    else memset(cor + (size_t)g * nt, 0, nt);
    if (overall) { /* LRT df follows the fitted column count (batch.rs:2096-2113) */
      double crit_overall = crow[3 + nt];
      if (reduced) {
        const mcpo_red_crit* e = mcpo_reduced_crit(cache, spec, n, p_red);
        if (!e) { rc = MCP_ERR_INVALID_SPEC; break; }
        crit_overall = e->row[3 + nt];
      }
      uint8_t sig = 0;
      if (f.converged && isfinite(f.deviance) && isfinite(f.deviance_null)) {
        double lrt = f.deviance_null - f.deviance;
        if (isfinite(lrt) && lrt > crit_overall) sig = 1;
      }
      overall[g] = sig;
    }
    if (nf && fx) memcpy(fx + (size_t)g * nf, flags, nf);
  }
  free(xr);
  return rc;
}
