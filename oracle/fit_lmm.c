/*
 * oracle/fit_lmm.c -- CPU ORACLE (test infrastructure): random-intercept LMM by profiled REML
 * and the Mle arm of run_one_sim.
 *
 * The reference delegates the fit to the third-party crates glmm 0.1.2 (`loop_advanced::fit_on`)
 * and bobyqa 0.1.1 (Cargo.lock:2060-2063, :379-382), which are NOT in the tree.  Readable:
 *   scripts/lme_handcompute_theta1.py:55-93 -- the closed-form profiled REML deviance of a random
 *     intercept in theta = tau/sigma:  V_c^-1 = I - s_c 11', s_c = theta^2 / (1 + theta^2 n_c);
 *     dev(theta) = sum_c log(1 + theta^2 n_c) + log|X'V^-1X| + (N - P) log(sigma_hat^2);
 *   crates/engine-core/src/batch.rs:1214-1560 -- the Mle arm: refit per grid N; start theta =
 *     truth (tau, since the DGP has sigma = 1: mixed_workspace.rs:125) when scenario.truth_start
 *     or the carried theta-hat of the previous converged grid point (:1323-1332, :1423-1430), else
 *     a blind start; outputs predictor-indexed Wald z^2 (:1413-1415), joint Wald chi2 over the
 *     targets vs chi2_ppf(1-alpha, k) (:1453-1477, joint_cor == joint_unc), boundary_hit
 *     (spec.rs:562-565: 1 = tau-hat ~ 0, 2 = optimiser stopped at the upper theta bound),
 *     pinned_components bit 0, overall always 0 (:1449-1451);
 *   engine-contract/src/generation.rs:125-158 -- row -> cluster maps.
 * The 1-D minimiser here is Brent's method (golden section + parabolic steps) on [0, THETA_MAX];
 * the reference minimises the same objective with BOBYQA (rho_end 1e-6, validation/tolerances.R:38-41).
 *
 * PARITY STATUS: beta, se, z pinned to lme4 1.1.38 REML on the reference's own bytes
 * (validation/data/lme_*.golden.rds) at 1e-4 relative.  The optimiser trajectory, the blind start
 * THETA0, the boundary-hit classification and theta-hat warm-carry effects are NOT readable:
 * **parity unpinned** beyond those bands (SURVEY.md 8c, row "LMM fit").
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define THETA_MAX 32.0        /* tau/sigma = 32 <=> ICC 0.999 */
#define THETA_BLIND 1.0       /* blind start (lme4's default theta start) */
#define THETA_TOL 1e-9
#define THETA_ZERO 1e-6       /* theta-hat below this is reported as the tau-hat ~ 0 boundary */
#define BRENT_MAX_ITERS 100
#define LMM_NEAR_ZERO 1e-30

typedef struct lmm_data {
  uint32_t p, n, n_clusters;
  double xtx[MCPO_MAX_P * MCPO_MAX_P]; /* row-major, lower valid */
  double xty[MCPO_MAX_P];
  double yty;
  const double* csum; /* per cluster: [n_c, Sx_1..Sx_{p-1}, Sy]  (Sx_0 = n_c) */
} lmm_data;

typedef struct lmm_eval {
  int ok;
  double dev, sigma_sq;
  double beta[MCPO_MAX_P];
  double L[MCPO_MAX_P * MCPO_MAX_P]; /* lower Cholesky factor of X'V^-1X */
} lmm_eval;

static void lmm_evaluate(const lmm_data* d, double theta, lmm_eval* e) {
  const uint32_t p = d->p;
  double A[MCPO_MAX_P * MCPO_MAX_P], b[MCPO_MAX_P], yy = d->yty, logdetv = 0.0;
  memcpy(A, d->xtx, sizeof(double) * p * p);
  memcpy(b, d->xty, sizeof(double) * p);
  const double th2 = theta * theta;
  for (uint32_t c = 0; c < d->n_clusters; ++c) {
    const double* cs = d->csum + (size_t)c * (p + 1);
    double nc = cs[0];
    if (nc <= 0.0) continue;
    double s = th2 / (1.0 + th2 * nc);
    double sx[MCPO_MAX_P];
    sx[0] = nc;
    for (uint32_t a = 1; a < p; ++a) sx[a] = cs[a];
    double sy = cs[p];
    for (uint32_t a = 0; a < p; ++a) {
      for (uint32_t q = 0; q <= a; ++q) A[a * p + q] -= s * sx[a] * sx[q];
      b[a] -= s * sx[a] * sy;
    }
    yy -= s * sy * sy;
    logdetv += log(1.0 + th2 * nc);
  }
  e->ok = 0;
  e->dev = INFINITY;
  double* L = e->L;
  memset(L, 0, sizeof(double) * p * p);
  double logdetA = 0.0;
  for (uint32_t j = 0; j < p; ++j) {
    double dd = A[j * p + j];
    for (uint32_t t = 0; t < j; ++t) dd -= L[j * p + t] * L[j * p + t];
    if (!(dd > 0.0) || !isfinite(dd)) return;
    double ljj = sqrt(dd);
    L[j * p + j] = ljj;
    logdetA += 2.0 * log(ljj);
    for (uint32_t i = j + 1; i < p; ++i) {
      double v = A[i * p + j];
      for (uint32_t t = 0; t < j; ++t) v -= L[i * p + t] * L[j * p + t];
      L[i * p + j] = v / ljj;
    }
  }
  double z[MCPO_MAX_P];
  for (uint32_t i = 0; i < p; ++i) {
    double v = b[i];
    for (uint32_t t = 0; t < i; ++t) v -= L[i * p + t] * z[t];
    z[i] = v / L[i * p + i];
  }
  for (int64_t i = (int64_t)p - 1; i >= 0; --i) {
    double v = z[i];
    for (uint32_t t = (uint32_t)i + 1; t < p; ++t) v -= L[t * p + i] * e->beta[t];
    e->beta[i] = v / L[i * p + i];
  }
  double bxy = 0.0;
  for (uint32_t i = 0; i < p; ++i) bxy += e->beta[i] * b[i];
  double df = (double)d->n - (double)p;
  e->sigma_sq = (yy - bxy) / df;
  if (!(e->sigma_sq > 0.0) || !isfinite(e->sigma_sq)) return;
  e->dev = logdetv + logdetA + df * log(e->sigma_sq);
  e->ok = isfinite(e->dev);
}

/* Brent's minimiser of dev(theta) on [0, THETA_MAX], first probe at `start` (fmin, Brent 1973 ch. 5) */
static double lmm_minimise(const lmm_data* d, double start, lmm_eval* best, uint32_t* n_eval) {
  const double CGOLD = 0.3819660112501051;
  double a = 0.0, b = THETA_MAX;
  double x = start;
  if (!(x > a + 1e-3)) x = a + 1e-3;
  if (!(x < b - 1e-3)) x = b - 1e-3;
  double w = x, v = x, e = 0.0, dstep = 0.0;
  lmm_eval ev;
  lmm_evaluate(d, x, &ev);
  *n_eval = 1;
  double fx = ev.dev, fw = fx, fv = fx;
  *best = ev;
  for (int it = 0; it < BRENT_MAX_ITERS; ++it) {
    double xm = 0.5 * (a + b);
    double tol1 = THETA_TOL * fabs(x) + 1e-10, tol2 = 2.0 * tol1;
    if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
    int golden = 1;
    if (fabs(e) > tol1) {
      double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), pp = (x - v) * q - (x - w) * r;
      q = 2.0 * (q - r);
      if (q > 0.0) pp = -pp;
      q = fabs(q);
      double etemp = e;
      e = dstep;
      if (!(fabs(pp) >= fabs(0.5 * q * etemp) || pp <= q * (a - x) || pp >= q * (b - x))) {
        dstep = pp / q;
        double u = x + dstep;
        if (u - a < tol2 || b - u < tol2) dstep = (xm - x >= 0.0) ? tol1 : -tol1;
        golden = 0;
      }
    }
    if (golden) {
      e = (x >= xm) ? a - x : b - x;
      dstep = CGOLD * e;
    }
    double u = fabs(dstep) >= tol1 ? x + dstep : x + (dstep >= 0.0 ? tol1 : -tol1);
    lmm_evaluate(d, u, &ev);
    *n_eval += 1;
    double fu = ev.dev;
    if (fu <= fx) {
      if (u >= x) a = x; else b = x;
      v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
      *best = ev;
    } else {
      if (u < x) a = u; else b = u;
      if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    }
  }
  return x;
}

typedef struct lmm_fit {
  int converged;
  uint8_t boundary_hit;
  double theta, sigma_sq, tau_sq;
  uint32_t n_eval;
  lmm_eval ev;
} lmm_fit;

static void lmm_fit_data(const lmm_data* d, double theta_start, lmm_fit* f) {
  memset(f, 0, sizeof *f);
  lmm_eval at0;
  double th = lmm_minimise(d, theta_start, &f->ev, &f->n_eval);
  /* the interior minimiser never evaluates theta = 0 exactly: compare with the boundary value */
  lmm_evaluate(d, 0.0, &at0);
  f->n_eval += 1;
  if (at0.ok && (!f->ev.ok || at0.dev <= f->ev.dev || th < THETA_ZERO)) {
    f->ev = at0;
    th = 0.0;
  }
  f->theta = th;
  f->converged = f->ev.ok;
  f->boundary_hit = !f->ev.ok ? 0 : (th < THETA_ZERO ? 1 : (th > THETA_MAX - 1e-3 ? 2 : 0));
  if (f->boundary_hit == 2) f->converged = 0;
  f->sigma_sq = f->ev.sigma_sq;
  f->tau_sq = th * th * f->ev.sigma_sq;
}

/* Wald z^2 of c'beta: cov(beta) = sigma^2 (X'V^-1X)^-1 */
static double lmm_var(const lmm_fit* f, uint32_t p, const double* c) {
  double u[MCPO_MAX_P], nq = 0.0;
  for (uint32_t i = 0; i < p; ++i) {
    double acc = c[i];
    for (uint32_t t = 0; t < i; ++t) acc -= f->ev.L[i * p + t] * u[t];
    u[i] = acc / f->ev.L[i * p + i];
    nq += u[i] * u[i];
  }
  return f->sigma_sq * nq;
}

/* joint Wald chi2 of the target coefficients: b_T' [Cov_TT]^-1 b_T */
static double lmm_joint(const lmm_fit* f, uint32_t p, const uint32_t* targets, uint32_t k) {
  if (k == 0) return NAN;
  double cov[MCPO_MAX_P * MCPO_MAX_P], ci[MCPO_MAX_P], cj[MCPO_MAX_P];
  /* Cov_ij = sigma^2 (L^-1 e_i).(L^-1 e_j) */
  double U[MCPO_MAX_P][MCPO_MAX_P];
  for (uint32_t a = 0; a < k; ++a) {
    memset(ci, 0, sizeof ci);
    ci[targets[a]] = 1.0;
    for (uint32_t i = 0; i < p; ++i) {
      double acc = ci[i];
      for (uint32_t t = 0; t < i; ++t) acc -= f->ev.L[i * p + t] * U[a][t];
      U[a][i] = acc / f->ev.L[i * p + i];
    }
  }
  for (uint32_t a = 0; a < k; ++a)
    for (uint32_t b = 0; b <= a; ++b) {
      double s = 0.0;
      for (uint32_t i = 0; i < p; ++i) s += U[a][i] * U[b][i];
      cov[a * k + b] = f->sigma_sq * s;
    }
  /* Cholesky of Cov_TT, solve */
  double Lc[MCPO_MAX_P * MCPO_MAX_P];
  memset(Lc, 0, sizeof Lc);
  for (uint32_t j = 0; j < k; ++j) {
    double dd = cov[j * k + j];
    for (uint32_t t = 0; t < j; ++t) dd -= Lc[j * k + t] * Lc[j * k + t];
    if (!(dd > 0.0)) return NAN;
    Lc[j * k + j] = sqrt(dd);
    for (uint32_t i = j + 1; i < k; ++i) {
      double v = cov[i * k + j];
      for (uint32_t t = 0; t < j; ++t) v -= Lc[i * k + t] * Lc[j * k + t];
      Lc[i * k + j] = v / Lc[j * k + j];
    }
  }
  double q = 0.0;
  for (uint32_t i = 0; i < k; ++i) {
    double v = f->ev.beta[targets[i]];
    for (uint32_t t = 0; t < i; ++t) v -= Lc[i * k + t] * cj[t];
    cj[i] = v / Lc[i * k + i];
    q += cj[i] * cj[i];
  }
  return q;
}

static int build_lmm_data(const float* x, const float* y, const uint32_t* ids, uint32_t n, uint32_t ld, uint32_t p,
                          uint32_t n_clusters, lmm_data* d, double* csum) {
  memset(d, 0, sizeof *d);
  d->p = p; d->n = n; d->n_clusters = n_clusters; d->csum = csum;
  memset(csum, 0, sizeof(double) * (size_t)n_clusters * (p + 1));
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t c = ids[i];
    if (c >= n_clusters) return MCP_ERR_INVALID_SPEC;
    double yi = (double)y[i];
    double* cs = csum + (size_t)c * (p + 1);
    cs[0] += 1.0;
    cs[p] += yi;
    for (uint32_t a = 0; a < p; ++a) {
      double xa = a == 0 ? 1.0 : (double)x[(size_t)a * ld + i];
      if (a > 0) cs[a] += xa;
      for (uint32_t q = 0; q <= a; ++q) d->xtx[a * p + q] += xa * (q == 0 ? 1.0 : (double)x[(size_t)q * ld + i]);
      d->xty[a] += xa * yi;
    }
    d->yty += yi * yi;
  }
  return MCP_OK;
}

/* fit provided bytes (introspect::fit_provided_data twin for the random-intercept arm) */
int mcpo_lmm_fit(const float* x, const float* y, const uint32_t* cluster_ids, uint32_t n, uint32_t ld, uint32_t p,
                 uint32_t n_clusters, double theta_start, double* beta, double* se, double* z_sq, double* theta_hat,
                 double* sigma_sq, double* tau_sq, uint8_t* converged, uint8_t* boundary_hit, uint32_t* n_eval) {
  if (p > MCPO_MAX_P) return MCP_ERR_UNSUPPORTED;
  double* csum = (double*)malloc(sizeof(double) * (size_t)n_clusters * (p + 1));
  lmm_data d;
  int rc = build_lmm_data(x, y, cluster_ids, n, ld, p, n_clusters, &d, csum);
  if (rc) { free(csum); return rc; }
  lmm_fit f;
  lmm_fit_data(&d, theta_start > 0.0 ? theta_start : THETA_BLIND, &f);
  *converged = (uint8_t)f.converged; *boundary_hit = f.boundary_hit; *theta_hat = f.theta;
  *sigma_sq = f.sigma_sq; *tau_sq = f.tau_sq; *n_eval = f.n_eval;
  for (uint32_t j = 0; j < p; ++j) {
    double c[MCPO_MAX_P] = {0};
    c[j] = 1.0;
    double var = lmm_var(&f, p, c);
    beta[j] = f.converged ? f.ev.beta[j] : NAN;
    se[j] = f.converged ? sqrt(var) : NAN;
    z_sq[j] = (f.converged && var > LMM_NEAR_ZERO) ? f.ev.beta[j] * f.ev.beta[j] / var : NAN;
  }
  free(csum);
  return MCP_OK;
}

/* Mle arm of run_one_sim (batch.rs:1214-1560), random intercept only.  Exclusion flags are recounted from rows
 * [0, N) before every fit (:1236-1250); a reduced round fits the gathered kept columns, never updates the
 * grid-sequential theta carry (:1414-1428), and runs the joint Wald over the kept targets against chi2(k_red)
 * (:1456-1468); marginal / correction thresholds are Wald-z, df-free, and stay on the full table. */
int mcpo_mle_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, uint8_t* unc, uint8_t* cor, uint8_t* conv, uint8_t* bh,
                 uint8_t* joint_unc, uint8_t* joint_cor, uint8_t* fx, double* stat_row, uint64_t* pinned) {
  if (!spec->cluster.present) return MCP_ERR_UNSUPPORTED;
  const mcp_cluster* cl = &spec->cluster;
  /* random slopes or extra groupings: the general solver (fit_lmm_general.c); a random intercept alone keeps Brent */
  int general = cl->slopes.n_slopes || cl->n_extra_groupings;
  for (uint32_t e = 0; e < cl->n_extra_groupings; ++e) general |= cl->extra[e].slopes.n_slopes != 0;
  if (spec->outcome_kind != MCP_OUTCOME_CONTINUOUS || spec->n_contrast_pairs) return MCP_ERR_UNSUPPORTED;
  const uint32_t p = 1 + spec->n_non_factor + spec->n_factor_dummies + spec->n_interactions;
  const uint32_t nt = spec->n_target_indices, nf = spec->n_factors;
  const uint32_t cv = spec->cluster.sizing_value ? spec->cluster.sizing_value : 1;
  const int fixed_clusters = spec->cluster.sizing == MCP_CLUSTER_FIXED_CLUSTERS;
  const int test_reduces = spec->n_fit_columns && spec->n_fit_columns < p;
  uint32_t ncl_max = fixed_clusters ? cv : (max_n + cv - 1) / cv;
  double* csum = (double*)malloc(sizeof(double) * (size_t)(ncl_max ? ncl_max : 1) * (p + 1));
  uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * max_n);
  float* xr = NULL;
  for (uint32_t i = 0; i < max_n; ++i) ids[i] = mcpo_cluster_of_row(&spec->cluster, i);
  int has_prev = 0;
  double theta_carry = 0.0;
  double gen_carry[12];
  uint32_t* extra_ids[MCP_MAX_EXTRA_GROUPINGS] = {NULL, NULL};
  for (uint32_t e = 0; general && e < cl->n_extra_groupings; ++e) {
    extra_ids[e] = (uint32_t*)malloc(sizeof(uint32_t) * max_n);
    for (uint32_t i = 0; i < max_n; ++i) extra_ids[e][i] = mcpo_extra_level_of_row(cl, e, i);
  }
  uint32_t prefix_counts[MCPO_MAX_LEVELS], kept[MCPO_MAX_P];
  int32_t remap[MCPO_MAX_P];
  uint8_t flags[16] = {0};
  int rc = MCP_OK;
  for (uint32_t g = 0; g < n_ns && rc == MCP_OK; ++g) {
    uint32_t n = ns[g];
    if (nf && spec->factor_min_level_count > 0) {
      memset(prefix_counts, 0, sizeof prefix_counts);
      mcpo_update_factor_exclusions(ws->x, max_n, spec, 0, n, prefix_counts, flags);
    }
    int any_excl = 0;
    for (uint32_t q = 0; q < nf; ++q) any_excl |= flags[q] != 0;
    const int reduced = any_excl || test_reduces;
    uint32_t p_red = p;
    const float* xfit = ws->x;
    if (!reduced) {
      for (uint32_t c = 0; c < p; ++c) remap[c] = (int32_t)c;
    } else {
      p_red = mcpo_build_exclusion_remap(spec, flags, kept, remap);
      if (p_red == 0 || kept[0] != 0) { rc = MCP_ERR_UNSUPPORTED; break; } /* intercept-free mixed fit */
      if (!xr) xr = (float*)malloc(sizeof(float) * (size_t)max_n * p);
      for (uint32_t rj = 0; rj < p_red; ++rj)
        memcpy(xr + (size_t)rj * max_n, ws->x + (size_t)kept[rj] * max_n, sizeof(float) * n);
      xfit = xr;
    }
    /* clusters that own at least one of the first n rows (generation.rs:152-157); empty ones are skipped */
    uint32_t ncl = fixed_clusters ? cv : (n + cv - 1) / cv;
    lmm_fit f;
    uint64_t pin = 0;
    if (!general) {
      lmm_data d;
      rc = build_lmm_data(xfit, ws->y, ids, n, max_n, p_red, ncl, &d, csum);
      if (rc) break;
      double start = has_prev ? theta_carry
                              : (spec->scenario.truth_start ? sqrt(fmax(spec->cluster.tau_squared, 0.0)) : THETA_BLIND);
      if (start < 0.01) start = 0.01; /* THETA_TRUTH_FLOOR, batch.rs:1316-1321 */
      lmm_fit_data(&d, start, &f);
      if (!reduced && f.converged) { theta_carry = f.theta; has_prev = 1; }
      pin = f.boundary_hit == 1 ? 1u : 0u;
    } else {
      /* one term per grouping; a reduced round strips every random slope (batch.rs:1359-1376) and never
       * feeds or reads the theta carry (its theta has another shape, :1414-1428) */
      mcpo_re_term terms[MCPO_GEN_MAX_TERMS];
      memset(terms, 0, sizeof terms);
      uint32_t n_terms = 1 + cl->n_extra_groupings;
      for (uint32_t t = 0; t < n_terms; ++t) {
        const mcp_re_slopes* sl = t == 0 ? &cl->slopes : &cl->extra[t - 1].slopes;
        terms[t].q = reduced ? 1u : 1u + sl->n_slopes;
        terms[t].n_levels = t == 0 ? ncl : mcpo_extra_n_levels_at(cl, t - 1, n);
        terms[t].ids = t == 0 ? ids : extra_ids[t - 1];
        for (uint32_t d = 0; d < sl->n_slopes; ++d) terms[t].cols[d] = sl->design_cols[d];
      }
      double th0[12];
      const double* start = NULL;
      if (!reduced && has_prev) start = gen_carry;
      else if (spec->scenario.truth_start) { mcpo_truth_theta(cl, !reduced, th0); start = th0; }
      mcpo_gen_fit gf;
      rc = mcpo_lmm_general_fit_terms(xfit, ws->y, n, max_n, p_red, terms, n_terms, mcpo_elim_term(cl, max_n, 1), start, &gf);
      if (rc) break;
      memset(&f, 0, sizeof f);
      f.converged = gf.converged; f.boundary_hit = gf.boundary_hit; f.sigma_sq = gf.sigma_sq;
      f.ev.ok = gf.converged; f.ev.sigma_sq = gf.sigma_sq;
      memcpy(f.ev.beta, gf.beta, sizeof gf.beta);
      memcpy(f.ev.L, gf.L, sizeof gf.L);
      pin = gf.pinned;
      if (!reduced && gf.converged) { memcpy(gen_carry, gf.theta, sizeof(double) * gf.n_theta); has_prev = 1; }
    }
    if (pinned) pinned[g] = f.converged ? pin : 0;
    const double* crow = crit + (size_t)g * crit_stride;
    conv[g] = (uint8_t)f.converged;
    bh[g] = f.boundary_hit;
    double tsq[MCPO_MAX_P * 2];
    uint32_t kept_targets[MCPO_MAX_P * 2], k_red = 0;
    memset(unc + (size_t)g * nt, 0, nt);
    for (uint32_t t = 0; t < nt; ++t) {
      tsq[t] = NAN;
      int32_t r = remap[spec->target_indices[t]];
      if (r >= 0) kept_targets[k_red++] = (uint32_t)r;
      if (f.converged && r >= 0) {
        double c[MCPO_MAX_P] = {0};
        c[r] = 1.0;
        double var = lmm_var(&f, p_red, c), b = f.ev.beta[r];
        if (var > LMM_NEAR_ZERO && isfinite(var)) tsq[t] = b * b / var;
        if (!isnan(tsq[t]) && tsq[t] > crow[0]) unc[(size_t)g * nt + t] = 1;
      }
    }
    if (g == 0 && stat_row) memcpy(stat_row, tsq, sizeof(double) * nt);
    if (f.converged) mcpo_apply_correction(spec->correction_method, tsq, nt, crow[0], crow + 1, cor + (size_t)g * nt);
    else memset(cor + (size_t)g * nt, 0, nt);
    double joint_crit = !reduced ? crow[2 + nt] : (k_red == 0 ? INFINITY : mcpo_chi2_ppf(1.0 - spec->alpha, (double)k_red));
    double jq = (f.converged && k_red) ? lmm_joint(&f, p_red, kept_targets, k_red) : NAN;
    uint8_t js = (f.converged && isfinite(jq) && jq > joint_crit) ? 1 : 0;
    joint_unc[g] = js;
    joint_cor[g] = js;
    if (nf && fx) memcpy(fx + (size_t)g * nf, flags, nf);
  }
  free(csum);
  free(ids);
  free(xr);
  for (uint32_t e = 0; e < MCP_MAX_EXTRA_GROUPINGS; ++e) free(extra_ids[e]);
  return rc;
}
