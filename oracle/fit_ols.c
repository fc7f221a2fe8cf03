/*
 * oracle/fit_ols.c -- CPU ORACLE (test infrastructure): OLS on sufficient
 * statistics.
 *
 * The reference delegates this arithmetic to the third-party crate glmm 0.1.2
 * (`loop_advanced::{OlsSuffStats::add_rows, fit_suff_stats_t_sq,
 * ols_contrast_t_sq}`, Cargo.lock:2060-2063), which is NOT in the tree.  The
 * semantics below are restated from the published normal-equations algorithm
 * and from the reference's own call sites and consumers:
 *   crates/engine-core/src/batch.rs:884-898 (add_rows on row segments, grid
 *     reuse), :958-968 (fit call, EPS_RANK = 1e-12 at :41), :1015-1142
 *     (converged, t_sq at target rank, df_resid, rss, sst, overall F),
 *   crates/engine-core/src/posthoc.rs:62-99,127-149 (L = lower Cholesky of
 *     X'X kept in the fit, sigma_sq, se^2 = sigma^2 * ||L^-1 c||^2),
 *   crates/engine-core/src/introspect.rs:280-284.
 * Pinned by validation/data/ols_*.golden.rds (R lm on the same bytes):
 * beta / se / t within 1e-11 relative (validation/tolerances.R:24-30).
 */
#include <math.h>
#include <string.h>
#include "oracle_internal.h"

#define EPS_RANK 1e-12        /* batch.rs:41 */
#define FLOAT_NEAR_ZERO 1e-30 /* lib.rs:46 */

void mcpo_suff_reset(mcpo_suff* s, uint32_t p) {
  memset(s, 0, sizeof *s);
  s->p = p;
}

void mcpo_suff_add_rows(mcpo_suff* s, const float* x, const float* y, uint32_t ld, uint32_t from, uint32_t to) {
  const uint32_t p = s->p;
  for (uint32_t i = from; i < to; ++i) {
    double yi = (double)y[i];
    for (uint32_t a = 0; a < p; ++a) {
      double xa = (double)x[(size_t)a * ld + i];
      for (uint32_t b = 0; b <= a; ++b) s->xtx[a * p + b] += xa * (double)x[(size_t)b * ld + i];
      s->xty[a] += xa * yi;
    }
    s->yty += yi * yi;
    s->sum_y += yi;
  }
  s->n_rows += to - from;
}

void mcpo_fit_suff(const mcpo_suff* s, mcpo_olsfit_t* fit) {
  const uint32_t p = s->p;
  fit->p = p;
  fit->converged = 0;
  double* L = fit->L;
  memset(L, 0, sizeof(double) * p * p);
  double dmin = INFINITY, dmax = 0.0;
  int ok = 1;
  for (uint32_t j = 0; j < p && ok; ++j) {
    double d = s->xtx[j * p + j];
    for (uint32_t t = 0; t < j; ++t) d -= L[j * p + t] * L[j * p + t];
    if (!(d > 0.0) || !isfinite(d)) { ok = 0; break; }
    double ljj = sqrt(d);
    L[j * p + j] = ljj;
    if (ljj < dmin) dmin = ljj;
    if (ljj > dmax) dmax = ljj;
    for (uint32_t i = j + 1; i < p; ++i) {
      double v = s->xtx[i * p + j];
      for (uint32_t t = 0; t < j; ++t) v -= L[i * p + t] * L[j * p + t];
      L[i * p + j] = v / ljj;
    }
  }
  double n = (double)s->n_rows;
  fit->df_resid = n - (double)p;
  fit->sst = s->yty - s->sum_y * s->sum_y / n;
  fit->rss = NAN;
  fit->sigma_sq = NAN;
  if (!ok || !(dmin / dmax >= EPS_RANK)) return;
  double z[MCPO_MAX_P];
  for (uint32_t i = 0; i < p; ++i) {
    double v = s->xty[i];
    for (uint32_t t = 0; t < i; ++t) v -= L[i * p + t] * z[t];
    z[i] = v / L[i * p + i];
  }
  for (int64_t i = (int64_t)p - 1; i >= 0; --i) {
    double v = z[i];
    for (uint32_t t = (uint32_t)i + 1; t < p; ++t) v -= L[t * p + i] * fit->beta[t];
    fit->beta[i] = v / L[i * p + i];
  }
  double bxty = 0.0;
  for (uint32_t i = 0; i < p; ++i) bxty += fit->beta[i] * s->xty[i];
  fit->rss = s->yty - bxty;
  fit->sigma_sq = fit->rss / fit->df_resid;
  fit->converged = 1;
}

/* se^2 of c'beta = sigma^2 * ||L^-1 c||^2 (posthoc.rs:86-99,127-149); NaN when not well formed */
static double contrast_se_sq(const mcpo_olsfit_t* fit, const double* c) {
  const uint32_t p = fit->p;
  double u[MCPO_MAX_P], nq = 0.0;
  for (uint32_t i = 0; i < p; ++i) {
    double acc = c[i];
    for (uint32_t t = 0; t < i; ++t) acc -= fit->L[i * p + t] * u[t];
    double lii = fit->L[i * p + i];
    if (fabs(lii) < FLOAT_NEAR_ZERO) return NAN;
    u[i] = acc / lii;
  }
  for (uint32_t i = 0; i < p; ++i) nq += u[i] * u[i];
  if (!isfinite(nq)) return NAN;
  return fit->sigma_sq * nq;
}

double mcpo_contrast_t_sq(const mcpo_olsfit_t* fit, const double* c) {
  double est = 0.0;
  for (uint32_t j = 0; j < fit->p; ++j) est += c[j] * fit->beta[j];
  double se_sq = contrast_se_sq(fit, c);
  if (se_sq > FLOAT_NEAR_ZERO && isfinite(se_sq)) return est * est / se_sq;
  return NAN;
}

int mcpo_ols_fit(const float* x, const float* y, uint32_t n_rows, uint32_t ld, uint32_t p, double* beta, double* se,
                 double* t_sq, double* rss, double* sst, uint8_t* converged) {
  if (p > MCPO_MAX_P) return MCP_ERR_UNSUPPORTED;
  mcpo_suff s;
  mcpo_olsfit_t f;
  mcpo_suff_reset(&s, p);
  mcpo_suff_add_rows(&s, x, y, ld, 0, n_rows);
  mcpo_fit_suff(&s, &f);
  *converged = (uint8_t)f.converged;
  *rss = f.rss;
  *sst = f.sst;
  for (uint32_t j = 0; j < p; ++j) {
    double c[MCPO_MAX_P] = {0};
    c[j] = 1.0;
    beta[j] = f.converged ? f.beta[j] : NAN;
    t_sq[j] = f.converged ? mcpo_contrast_t_sq(&f, c) : NAN;
    se[j] = f.converged ? sqrt(contrast_se_sq(&f, c)) : NAN;
  }
  return MCP_OK;
}
