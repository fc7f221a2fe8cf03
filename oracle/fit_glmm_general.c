/*
 * oracle/fit_glmm_general.c -- CPU ORACLE (test infrastructure): clustered GLM with ANY random-effect structure the
 * shared POD can carry (random slopes on the primary grouping, up to two extra groupings crossed with or nested in it,
 * slopes on those), Laplace approximation.  Extra groupings couple the clusters, so the conditional modes are found
 * jointly: with b = Lambda_theta u and A = Z Lambda (one sparse row per observation),
 *   d(theta, beta) = sum_i devres_i + |u~|^2 + log|I + A'WA|,   u <- (I + A'WA)^-1 A'W (z - eta0)   (Fisher scoring),
 * by dense Cholesky over all R = sum_t q_t L_t random effects (R <= 128).  (theta, beta) jointly by the bounded
 * Nelder-Mead simplex of fit_glmm_slopes.c; Wald SE mode "rx": RX'RX = X'WX - (A'WX)'(I + A'WA)^-1 (A'WX).
 * On a single grouping this is fit_glmm_slopes.c's model with the scoring loop run globally instead of per cluster.
 *
 * PARITY STATUS: data bit-exact on validation/data/glmm_slope_crossed.rds, glmm_slope_crossed_extra_slopes.rds,
 * glmm_slope_nested_extra_slopes.rds; beta-hat against lme4 1.1.38 glmer on those bytes inside GLMM_TOL
 * (tolerances.R:186-200).  Optimiser, boundary rules: **parity unpinned**.  Checker of glmm_general_kernels.hip.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define GG_MAX_TERMS 3
#define GG_MAX_Q 3
#define GG_MAX_R 160
#define GG_MAX_DIM (MCPO_MAX_P + 12)
#define GG_MAX_EVAL 30000
#define GG_THETA_MAX 32.0
#define GG_THETA_ZERO 1e-6
#define GG_PIRLS_ITERS 60

/* test hook: tests lower the Fisher-step cap (here and, through MCP_DEBUG_GLMM_PIRLS_CAP, in the kernel) to drive every
 * evaluation into the cap-exit branch: the loop then stops on an accepted point that is not converged */
static int gg_pirls_cap = GG_PIRLS_ITERS;
void mcpo_set_glmm_general_pirls_cap(int cap) { gg_pirls_cap = (cap >= 1 && cap <= GG_PIRLS_ITERS) ? cap : GG_PIRLS_ITERS; }

typedef struct gg_model {
  int fam;
  uint32_t n, ld, p, nt, nth, R;
  mcpo_re_term terms[GG_MAX_TERMS];
  uint32_t roff[GG_MAX_TERMS], toff[GG_MAX_TERMS];
  uint8_t is_diag[12];
  const float *x, *y;
  double *eta0, *u, *M, *L, *r;  /* M, L: R x R row-major */
} gg_model;

static void gg_free(gg_model* m) { free(m->eta0); free(m->u); free(m->M); free(m->L); free(m->r); }

static void gg_templates(const gg_model* m, const double* th, double T[GG_MAX_TERMS][GG_MAX_Q][GG_MAX_Q]) {
  memset(T, 0, sizeof(double) * GG_MAX_TERMS * GG_MAX_Q * GG_MAX_Q);
  for (uint32_t t = 0; t < m->nt; ++t) {
    uint32_t k = m->toff[t];
    for (uint32_t c = 0; c < m->terms[t].q; ++c)
      for (uint32_t r = c; r < m->terms[t].q; ++r) T[t][r][c] = th[k++];
  }
}

/* sparse row of A = Z Lambda: up to sum_t q_t entries (index, value) */
static uint32_t gg_row(const gg_model* m, double T[GG_MAX_TERMS][GG_MAX_Q][GG_MAX_Q], uint32_t i, uint32_t* idx, double* val) {
  uint32_t k = 0;
  for (uint32_t t = 0; t < m->nt; ++t) {
    const mcpo_re_term* tm = &m->terms[t];
    double z[GG_MAX_Q];
    z[0] = 1.0;
    for (uint32_t d = 1; d < tm->q; ++d) z[d] = (double)m->x[(size_t)tm->cols[d - 1] * m->ld + i];
    const uint32_t base = m->roff[t] + tm->ids[i] * tm->q;
    for (uint32_t b = 0; b < tm->q; ++b) {
      double s = 0.0;
      for (uint32_t d = b; d < tm->q; ++d) s += T[t][d][b] * z[d];
      idx[k] = base + b; val[k] = s; ++k;
    }
  }
  return k;
}

static int gg_chol(const double* a, uint32_t R, double* l) {
  memset(l, 0, sizeof(double) * R * R);
  for (uint32_t j = 0; j < R; ++j) {
    double d = a[(size_t)j * R + j];
    for (uint32_t k = 0; k < j; ++k) d -= l[(size_t)j * R + k] * l[(size_t)j * R + k];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    const double ljj = sqrt(d);
    l[(size_t)j * R + j] = ljj;
    for (uint32_t i = j + 1; i < R; ++i) {
      double v = a[(size_t)i * R + j];
      for (uint32_t k = 0; k < j; ++k) v -= l[(size_t)i * R + k] * l[(size_t)j * R + k];
      l[(size_t)i * R + j] = v / ljj;
    }
  }
  return 0;
}

static double gg_deviance(gg_model* m, const double* par) {
  const uint32_t p = m->p, R = m->R;
  const double* beta = par + m->nth;
  double T[GG_MAX_TERMS][GG_MAX_Q][GG_MAX_Q];
  gg_templates(m, par, T);
  for (uint32_t i = 0; i < m->n; ++i) {
    double e = 0.0;
    for (uint32_t c = 0; c < p; ++c) e += (c == 0 ? 1.0 : (double)m->x[(size_t)c * m->ld + i]) * beta[c];
    m->eta0[i] = e;
  }
  /* joint conditional modes by Fisher scoring with step halving (see fit_glmm.c): m->u, M = L L', dev, uu, logdet
   * belong to the ACCEPTED point, ut is the trial; a trial whose penalised deviance is not below the accepted one's is
   * pulled half-way back to it (the undamped step 2-cycles on levels far from their prior) */
  double* u = m->u;
  memset(u, 0, sizeof(double) * R);
  double pd_acc = INFINITY, dev = 0.0, logdet = 0.0, uu = 0.0;
  double ut[GG_MAX_R], un[GG_MAX_R], tv[GG_MAX_R];
  memset(ut, 0, sizeof(double) * R);
  for (int it = 0; it < gg_pirls_cap; ++it) {
    memset(m->M, 0, sizeof(double) * R * R);
    memset(m->r, 0, sizeof(double) * R);
    double devt = 0.0;
    for (uint32_t i = 0; i < m->n; ++i) {
      uint32_t idx[GG_MAX_TERMS * GG_MAX_Q];
      double val[GG_MAX_TERMS * GG_MAX_Q], au = 0.0;
      const uint32_t k = gg_row(m, T, i, idx, val);
      for (uint32_t a = 0; a < k; ++a) au += val[a] * ut[idx[a]];
      double w, zw, dv;
      mcpo_fam_eval(m->fam, m->eta0[i] + au, (double)m->y[i], &w, &zw, &dv);
      devt += dv;
      const double res = w * (zw - m->eta0[i]);
      for (uint32_t a = 0; a < k; ++a) {
        m->r[idx[a]] += res * val[a];
        for (uint32_t b = 0; b < k; ++b) m->M[(size_t)idx[a] * R + idx[b]] += w * val[a] * val[b];
      }
    }
    double uut = 0.0;
    for (uint32_t a = 0; a < R; ++a) { m->M[(size_t)a * R + a] += 1.0; uut += ut[a] * ut[a]; }
    const double pd = devt + uut, tol = 1e-13 * (fabs(pd) + 1.0);
    if (!(pd <= pd_acc + tol)) {                     /* worse than the accepted point (or not finite): halve the step */
      int stuck_t = 1, stuck_a = 1;
      for (uint32_t a = 0; a < R; ++a) {
        const double half = 0.5 * (ut[a] + u[a]);
        if (half != ut[a]) stuck_t = 0;
        if (half != u[a]) stuck_a = 0;
        ut[a] = half;
      }
      if (stuck_t || stuck_a) break;
      continue;                                      /* m->L still holds the accepted point's factor */
    }
    if (gg_chol(m->M, R, m->L)) return INFINITY;
    logdet = 0.0;
    for (uint32_t a = 0; a < R; ++a) logdet += 2.0 * log(m->L[(size_t)a * R + a]);
    const int conv = fabs(pd_acc - pd) <= tol;       /* pd_acc = inf on the first pass: never */
    memcpy(u, ut, sizeof(double) * R);
    dev = devt; uu = uut; pd_acc = pd;
    if (conv) break;
    for (uint32_t a = 0; a < R; ++a) {
      double v = m->r[a];
      for (uint32_t k = 0; k < a; ++k) v -= m->L[(size_t)a * R + k] * tv[k];
      tv[a] = v / m->L[(size_t)a * R + a];
    }
    for (int a = (int)R - 1; a >= 0; --a) {
      double v = tv[a];
      for (uint32_t k = (uint32_t)a + 1; k < R; ++k) v -= m->L[(size_t)k * R + (uint32_t)a] * un[k];
      un[a] = v / m->L[(size_t)a * R + (uint32_t)a];
    }
    int same = 1;
    for (uint32_t a = 0; a < R; ++a) if (un[a] != u[a]) same = 0;
    if (same) break;
    memcpy(ut, un, sizeof(double) * R);
  }
  if (!(pd_acc < INFINITY)) return INFINITY;         /* not even the prior modes evaluate */
  const double total = dev + uu + logdet;
  return isfinite(total) ? total : INFINITY;
}

static double gg_f(gg_model* m, double* par, uint32_t* n_eval) {
  for (uint32_t k = 0; k < m->nth; ++k) {
    if (m->is_diag[k]) par[k] = par[k] < 0.0 ? 0.0 : (par[k] > GG_THETA_MAX ? GG_THETA_MAX : par[k]);
    else par[k] = par[k] < -GG_THETA_MAX ? -GG_THETA_MAX : (par[k] > GG_THETA_MAX ? GG_THETA_MAX : par[k]);
  }
  for (uint32_t k = m->nth; k < m->nth + m->p; ++k) par[k] = par[k] < -50.0 ? -50.0 : (par[k] > 50.0 ? 50.0 : par[k]);
  ++*n_eval;
  return gg_deviance(m, par);
}

/* no-tree-vectorize: gcc 11.4 at -O3 -march=native (the cpu_baseline build, AVX-512 hosts) emits an ALIGNED 32-byte load
 * for the shrink loop below at an address that is only 16-byte aligned -- a #GP on the first shrink step (seen on the
 * (1 + x1 | g) + (1 | item) bench shape).  The simplex arithmetic is a vanishing share of a fit; scalar code is exact too. */
__attribute__((optimize("no-tree-vectorize")))
static double gg_nelder_mead(gg_model* m, double* x0, double step, double ftol, double xtol, uint32_t* n_eval) {
  const uint32_t d = m->nth + m->p;
  double sx[GG_MAX_DIM + 1][GG_MAX_DIM], fv[GG_MAX_DIM + 1];
  memcpy(sx[0], x0, sizeof(double) * d);
  fv[0] = gg_f(m, sx[0], n_eval);
  for (uint32_t i = 0; i < d; ++i) {
    memcpy(sx[i + 1], x0, sizeof(double) * d);
    sx[i + 1][i] += step;
    fv[i + 1] = gg_f(m, sx[i + 1], n_eval);
  }
  while (*n_eval < GG_MAX_EVAL) {
    uint32_t lo = 0, hi = 0, nh;
    for (uint32_t i = 1; i <= d; ++i) { if (fv[i] < fv[lo]) lo = i; if (fv[i] > fv[hi]) hi = i; }
    nh = lo;
    for (uint32_t i = 0; i <= d; ++i) if (i != hi && fv[i] > fv[nh]) nh = i;
    double size = 0.0;
    for (uint32_t i = 0; i <= d; ++i)
      for (uint32_t k = 0; k < d; ++k) size = fmax(size, fabs(sx[i][k] - sx[lo][k]));
    if (isfinite(fv[hi]) && fv[hi] - fv[lo] <= ftol && size <= xtol) break;
    double cen[GG_MAX_DIM], xr[GG_MAX_DIM], xe[GG_MAX_DIM];
    for (uint32_t k = 0; k < d; ++k) {
      double s = 0.0;
      for (uint32_t i = 0; i <= d; ++i) if (i != hi) s += sx[i][k];
      cen[k] = s / (double)d;
      xr[k] = cen[k] + (cen[k] - sx[hi][k]);
    }
    double fr = gg_f(m, xr, n_eval);
    if (fr < fv[lo]) {
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 2.0 * (xr[k] - cen[k]);
      double fe = gg_f(m, xe, n_eval);
      if (fe < fr) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fe; }
      else { memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr; }
    } else if (fr < fv[nh]) {
      memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr;
    } else {
      int outside = fr < fv[hi];
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 0.5 * ((outside ? xr[k] : sx[hi][k]) - cen[k]);
      double fc = gg_f(m, xe, n_eval);
      if (fc < (outside ? fr : fv[hi])) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fc; }
      else {
        for (uint32_t i = 0; i <= d; ++i) {
          if (i == lo) continue;
          for (uint32_t k = 0; k < d; ++k) sx[i][k] = sx[lo][k] + 0.5 * (sx[i][k] - sx[lo][k]);
          fv[i] = gg_f(m, sx[i], n_eval);
        }
      }
    }
  }
  uint32_t lo = 0;
  for (uint32_t i = 1; i <= d; ++i) if (fv[i] < fv[lo]) lo = i;
  memcpy(x0, sx[lo], sizeof(double) * d);
  return fv[lo];
}

static double gg_dev_cb(void* ctx, double* par) {
  uint32_t scratch = 0;
  return gg_f((gg_model*)ctx, par, &scratch);
}

typedef struct gg_count_cb { gg_model* m; uint32_t* n_eval; } gg_count_cb;
static double gg_probe_cb(void* ctx, double* par) {  /* a probe of the stationarity check: counted like any evaluation */
  gg_count_cb* c = (gg_count_cb*)ctx;
  return gg_f(c->m, par, c->n_eval);
}

/* mode: MCPO_MODE_HESSIAN selects the finite-difference Hessian Wald covariance (fit_glmm.c) over the free parameters */
int mcpo_glmm_general_fit_terms(int fam, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                                const mcpo_re_term* terms, uint32_t n_terms, uint32_t elim, const double* theta0, const double* beta0,
                                uint32_t mode, mcpo_gen_fit* f) {
  if (p > MCPO_MAX_P || n_terms == 0 || n_terms > GG_MAX_TERMS) return MCP_ERR_UNSUPPORTED;
  gg_model m;
  memset(&m, 0, sizeof m);
  m.fam = fam; m.n = n; m.ld = ld; m.p = p; m.nt = n_terms; m.x = x; m.y = y;
  if (elim >= n_terms) return MCP_ERR_UNSUPPORTED;
  /* index layout of the random effects: the term `elim` leads (mcpo_elim_term: the one with the most columns -- its levels
   * are mutually exclusive, so the leading part of I + A'WA is block diagonal and the device kernel factors it one block
   * per lane), the others follow in declaration order; theta stays in declaration order */
  m.roff[elim] = 0; m.R = terms[elim].q * terms[elim].n_levels;
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (terms[t].q == 0 || terms[t].q > GG_MAX_Q) return MCP_ERR_UNSUPPORTED;
    m.terms[t] = terms[t];
    if (t != elim) { m.roff[t] = m.R; m.R += terms[t].q * terms[t].n_levels; }
    m.toff[t] = m.nth;
    for (uint32_t c = 0; c < terms[t].q; ++c)
      for (uint32_t r = c; r < terms[t].q; ++r) m.is_diag[m.nth++] = r == c;
    for (uint32_t i = 0; i < n; ++i) if (terms[t].ids[i] >= terms[t].n_levels) return MCP_ERR_INVALID_SPEC;
  }
  if (m.R > GG_MAX_R || m.nth > 12) return MCP_ERR_UNSUPPORTED;
  m.eta0 = (double*)malloc(sizeof(double) * (n ? n : 1));
  m.u = (double*)calloc(m.R, sizeof(double));
  m.r = (double*)calloc(m.R, sizeof(double));
  m.M = (double*)malloc(sizeof(double) * m.R * m.R);
  m.L = (double*)malloc(sizeof(double) * m.R * m.R);
  if (!m.eta0 || !m.u || !m.r || !m.M || !m.L) { gg_free(&m); return MCP_ERR_DEVICE; }
  const uint32_t R = m.R;
  double par[GG_MAX_DIM];
  for (uint32_t k = 0; k < m.nth; ++k) {
    par[k] = theta0 ? theta0[k] : (m.is_diag[k] ? 1.0 : 0.0);
    if (theta0 && par[k] < 0.01) par[k] = 0.01;
  }
  for (uint32_t c = 0; c < p; ++c) par[m.nth + c] = beta0 ? beta0[c] : 0.0;
  memset(f, 0, sizeof *f);
  uint32_t n_eval = 0;
  double best = gg_nelder_mead(&m, par, 0.25, 1e-10, 1e-7, &n_eval);
  /* restarts only for a point the stationarity check rejects (oracle_internal.h) */
  uint8_t lower0[GG_MAX_DIM] = {0};
  for (uint32_t k = 0; k < m.nth; ++k) lower0[k] = m.is_diag[k];
  gg_count_cb probe = {&m, &n_eval};
  const int stationary = mcpo_stationary(gg_probe_cb, &probe, par, m.nth + p, lower0, best);
  for (int restart = 0; !stationary && restart < 4 && n_eval < GG_MAX_EVAL; ++restart) {
    double prev = best;
    best = gg_nelder_mead(&m, par, restart == 0 ? 0.05 : 0.01, 1e-10, 1e-7, &n_eval);
    if (!(prev - best > 1e-8)) break;
  }
  double cur = gg_f(&m, par, &n_eval);
  for (uint32_t k = 0; k < m.nth; ++k) {
    if (!m.is_diag[k] || par[k] == 0.0 || par[k] > 1e-4) continue;
    double keep = par[k];
    par[k] = 0.0;
    double f0 = gg_f(&m, par, &n_eval);
    if (f0 <= cur + 1e-8) cur = f0; else par[k] = keep;
  }
  cur = gg_f(&m, par, &n_eval);
  f->n_eval = n_eval; f->n_theta = m.nth; f->p = p; f->dev = cur; f->sigma_sq = 1.0;
  memcpy(f->theta, par, sizeof(double) * m.nth);
  memcpy(f->beta, par + m.nth, sizeof(double) * p);
  int ok = isfinite(cur) && n_eval < GG_MAX_EVAL;
  /* RX'RX = X'WX - C' M^-1 C, C = A'WX (R x p); M = L L' of the last evaluation */
  double A[MCPO_MAX_P * MCPO_MAX_P];
  memset(A, 0, sizeof A);
  if (ok) {
    double T[GG_MAX_TERMS][GG_MAX_Q][GG_MAX_Q];
    gg_templates(&m, par, T);
    double* Cm = (double*)calloc((size_t)R * p, sizeof(double));
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t idx[GG_MAX_TERMS * GG_MAX_Q];
      double val[GG_MAX_TERMS * GG_MAX_Q], au = 0.0;
      const uint32_t k = gg_row(&m, T, i, idx, val);
      for (uint32_t a = 0; a < k; ++a) au += val[a] * m.u[idx[a]];
      double w, zw, dv;
      mcpo_fam_eval(fam, m.eta0[i] + au, (double)y[i], &w, &zw, &dv);
      for (uint32_t r = 0; r < p; ++r) {
        double xr = r == 0 ? 1.0 : (double)x[(size_t)r * ld + i];
        for (uint32_t a = 0; a < k; ++a) Cm[(size_t)idx[a] * p + r] += w * val[a] * xr;
        for (uint32_t s = 0; s <= r; ++s) A[r * p + s] += w * xr * (s == 0 ? 1.0 : (double)x[(size_t)s * ld + i]);
      }
    }
    for (uint32_t r = 0; r < p; ++r) /* D = L^-1 C, column r */
      for (uint32_t a = 0; a < R; ++a) {
        double v = Cm[(size_t)a * p + r];
        for (uint32_t k = 0; k < a; ++k) v -= m.L[(size_t)a * R + k] * Cm[(size_t)k * p + r];
        Cm[(size_t)a * p + r] = v / m.L[(size_t)a * R + a];
      }
    for (uint32_t r = 0; r < p; ++r)
      for (uint32_t s = 0; s <= r; ++s) {
        double v = 0.0;
        for (uint32_t a = 0; a < R; ++a) v += Cm[(size_t)a * p + r] * Cm[(size_t)a * p + s];
        A[r * p + s] -= v;
      }
    free(Cm);
  }
  memset(f->L, 0, sizeof f->L);
  for (uint32_t j = 0; j < p && ok; ++j) {
    double dd = A[j * p + j];
    for (uint32_t t = 0; t < j; ++t) dd -= f->L[j * p + t] * f->L[j * p + t];
    if (!(dd > 0.0) || !isfinite(dd)) { ok = 0; break; }
    f->L[j * p + j] = sqrt(dd);
    for (uint32_t i = j + 1; i < p; ++i) {
      double v = A[i * p + j];
      for (uint32_t t = 0; t < j; ++t) v -= f->L[i * p + t] * f->L[j * p + t];
      f->L[i * p + j] = v / f->L[j * p + j];
    }
  }
  int singular = 0;
  for (uint32_t k = 0; k < m.nth; ++k) if (m.is_diag[k] && par[k] < GG_THETA_ZERO) singular = 1;
  (void)singular;
  if (ok && (mode & MCPO_MODE_HESSIAN)) { /* over the free parameters (fit_glmm_slopes.c): pinned diagonals are even points */
    uint8_t th_free[12];
    for (uint32_t k = 0; k < m.nth; ++k) th_free[k] = !(m.is_diag[k] && par[k] < GG_THETA_ZERO);
    double lh[MCPO_MAX_P * MCPO_MAX_P];
    if (mcpo_fd_hessian_factor(gg_dev_cb, &m, par, m.nth, p, th_free, cur, lh) == 0) memcpy(f->L, lh, sizeof(double) * p * p);
  }
  f->converged = ok;
  uint32_t comp = 0;
  for (uint32_t k = 0; k < m.nth; ++k) {
    if (!m.is_diag[k]) continue;
    if (ok && par[k] < GG_THETA_ZERO) f->pinned |= 1ull << comp;
    if (par[k] > GG_THETA_MAX - 1e-3) f->at_upper = 1;
    ++comp;
  }
  f->boundary_hit = !ok ? 0 : (f->at_upper ? 2 : (f->pinned ? 1 : 0));
  if (f->at_upper) f->converged = 0;
  gg_free(&m);
  return MCP_OK;
}

/* provided-bytes entry, mirroring mcpo_lmm_general_fit: vars_out = Lambda_t Lambda_t' per term (3 x 3 row-major each) */
int mcpo_glmm_general_fit(int family, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, uint32_t n_terms,
                          const uint32_t* term_q, const uint32_t* term_levels, const uint32_t* ids, const uint32_t* slope_cols,
                          const double* theta_start, const double* beta_start, double* beta, double* se, double* theta_hat,
                          double* vars_out, uint8_t* converged, uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval,
                          uint32_t mode) {
  if (n_terms == 0 || n_terms > GG_MAX_TERMS) return MCP_ERR_UNSUPPORTED;
  mcpo_re_term terms[GG_MAX_TERMS];
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (term_q[t] == 0 || term_q[t] > 1 + MCP_MAX_RE_SLOPES) return MCP_ERR_UNSUPPORTED;
    terms[t].q = term_q[t]; terms[t].n_levels = term_levels[t]; terms[t].ids = ids + (size_t)t * ld;
    for (uint32_t d = 0; d < MCP_MAX_RE_SLOPES; ++d) terms[t].cols[d] = slope_cols[t * MCP_MAX_RE_SLOPES + d];
  }
  mcpo_gen_fit f;
  /* mode bits 16..17: the leading term of the index layout (mcpo_elim_term); the result does not depend on it beyond rounding */
  int rc = mcpo_glmm_general_fit_terms(family, x, y, n, ld, p, terms, n_terms, (mode >> 16) & 3u, theta_start, beta_start, mode & 0xffffu, &f);
  if (rc) return rc;
  *converged = (uint8_t)f.converged; *boundary_hit = f.boundary_hit; *pinned = f.pinned; *n_eval = f.n_eval;
  memcpy(theta_hat, f.theta, sizeof(double) * f.n_theta);
  for (uint32_t j = 0; j < p; ++j) {
    double u[MCPO_MAX_P], nq = 0.0;
    for (uint32_t i = 0; i < p; ++i) {
      double acc = i == j ? 1.0 : 0.0;
      for (uint32_t t = 0; t < i; ++t) acc -= f.L[i * p + t] * u[t];
      u[i] = f.converged ? acc / f.L[i * p + i] : NAN;
      nq += u[i] * u[i];
    }
    beta[j] = f.converged ? f.beta[j] : NAN;
    se[j] = f.converged ? sqrt(nq) : NAN;
  }
  uint32_t k = 0;
  for (uint32_t t = 0; t < n_terms; ++t) {
    double T[3][3] = {{0}};
    for (uint32_t c = 0; c < term_q[t]; ++c)
      for (uint32_t r = c; r < term_q[t]; ++r) T[r][c] = f.theta[k++];
    for (uint32_t a = 0; a < 3; ++a)
      for (uint32_t b = 0; b < 3; ++b) {
        double s = 0.0;
        for (uint32_t c = 0; c < 3; ++c) s += T[a][c] * T[b][c];
        vars_out[t * 9 + a * 3 + b] = s;
      }
  }
  return MCP_OK;
}
