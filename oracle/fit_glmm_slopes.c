/*
 * oracle/fit_glmm_slopes.c -- CPU ORACLE (test infrastructure): clustered GLM with random slopes on the primary
 * grouping (q = 1 + slopes <= 3 random effects per cluster), Laplace approximation.  Generalises fit_glmm.c (random
 * intercept, scalar modes) to vector modes; same sources, same objective:
 *   d(theta, beta) = sum_i devres_i + sum_c |u_c|^2 + sum_c log|I + Lambda' Z_c' W_c Z_c Lambda|,
 * b_c = Lambda_theta u_c, Lambda the lower-triangular template of mixed_workspace.rs:112-198, the conditional modes by
 * per-cluster Fisher scoring  u <- (I + A'WA)^-1 A'W (z - eta0),  A = Z_c Lambda.  (theta, beta) jointly by the bounded
 * Nelder-Mead simplex; Wald SE mode "rx".
 *
 * PARITY STATUS: data bit-exact on validation/data/glmm_slope.rds / glmm_multislope.rds; beta-hat within the
 * reference's GLMM_TOL of lme4 1.1.38 glmer (tolerances.R:186-200); variance components at the looser band glmer's
 * own convergence allows on (near-)singular fits.  Optimiser, boundary rules: **parity unpinned**.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define GS_MAX_Q 3
#define GS_MAX_DIM (MCPO_MAX_P + 6)
#define GS_MAX_EVAL 12000
#define GS_THETA_MAX 32.0
#define GS_THETA_ZERO 1e-6
#define GS_PIRLS_ITERS 60

typedef struct gs_model {
  int fam;
  uint32_t n, ld, p, ncl, q, nth;
  uint32_t cols[GS_MAX_Q - 1];
  const float *x, *y;
  uint32_t *start, *rows;
  double *eta0;
  double *u;   /* ncl x q modes */
  double *Lm;  /* ncl x q x q lower Cholesky factors of I + A'WA at the modes */
  uint32_t nagq;        /* 1 = Laplace; k > 1: the k^q-point tensor adaptive Gauss-Hermite rule */
  double gz[25], gw[25];
} gs_model;

static void gs_free(gs_model* m) {
  free(m->start); free(m->rows); free(m->eta0); free(m->u); free(m->Lm);
  memset(m, 0, sizeof *m);
}

static void gs_template(const gs_model* m, const double* th, double T[GS_MAX_Q][GS_MAX_Q]) {
  uint32_t k = 0;
  for (uint32_t r = 0; r < GS_MAX_Q; ++r) for (uint32_t c = 0; c < GS_MAX_Q; ++c) T[r][c] = 0.0;
  for (uint32_t c = 0; c < m->q; ++c)
    for (uint32_t r = c; r < m->q; ++r) T[r][c] = th[k++];
}

/* lower Cholesky of the q x q matrix a (row-major); 0 on success */
static int gs_chol(const double* a, uint32_t q, double* l) {
  for (uint32_t i = 0; i < q * q; ++i) l[i] = 0.0;
  for (uint32_t j = 0; j < q; ++j) {
    double d = a[j * q + j];
    for (uint32_t k = 0; k < j; ++k) d -= l[j * q + k] * l[j * q + k];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    l[j * q + j] = sqrt(d);
    for (uint32_t i = j + 1; i < q; ++i) {
      double v = a[i * q + j];
      for (uint32_t k = 0; k < j; ++k) v -= l[i * q + k] * l[j * q + k];
      l[i * q + j] = v / l[j * q + j];
    }
  }
  return 0;
}

static double gs_deviance(gs_model* m, const double* par) {
  const uint32_t p = m->p, q = m->q;
  const double* beta = par + m->nth;
  double T[GS_MAX_Q][GS_MAX_Q];
  gs_template(m, par, T);
  for (uint32_t i = 0; i < m->n; ++i) {
    double e = 0.0;
    for (uint32_t c = 0; c < p; ++c) e += (c == 0 ? 1.0 : (double)m->x[(size_t)c * m->ld + i]) * beta[c];
    m->eta0[i] = e;
  }
  double total = 0.0;
  for (uint32_t c = 0; c < m->ncl; ++c) {
    const uint32_t lo = m->start[c], hi = m->start[c + 1];
    double* u = m->u + (size_t)c * q;
    double* Lm = m->Lm + (size_t)c * q * q;
    for (uint32_t b = 0; b < q; ++b) u[b] = 0.0;
    if (lo == hi) { for (uint32_t b = 0; b < q * q; ++b) Lm[b] = b % (q + 1) == 0 ? 1.0 : 0.0; continue; }
    /* Fisher scoring with step halving (see fit_glmm.c): (u, Lm, dev, logdet) belong to the ACCEPTED point, ut is the
     * trial; a trial whose penalised deviance is not below the accepted one's is pulled half-way back to it */
    double pd_acc = INFINITY, dev = 0.0, logdet = 0.0, ut[GS_MAX_Q] = {0};
    for (int it = 0; it < GS_PIRLS_ITERS; ++it) {
      double M[GS_MAX_Q * GS_MAX_Q] = {0}, r[GS_MAX_Q] = {0};
      double devt = 0.0;
      for (uint32_t k = lo; k < hi; ++k) {
        uint32_t i = m->rows[k];
        double z[GS_MAX_Q], a[GS_MAX_Q];
        z[0] = 1.0;
        for (uint32_t d = 1; d < q; ++d) z[d] = (double)m->x[(size_t)m->cols[d - 1] * m->ld + i];
        double au = 0.0;
        for (uint32_t b = 0; b < q; ++b) { /* a = Lambda' z */
          double s = 0.0;
          for (uint32_t d = b; d < q; ++d) s += T[d][b] * z[d];
          a[b] = s;
          au += s * ut[b];
        }
        double w, zw, dv;
        mcpo_fam_eval(m->fam, m->eta0[i] + au, (double)m->y[i], &w, &zw, &dv);
        devt += dv;
        const double res = w * (zw - m->eta0[i]);
        for (uint32_t b = 0; b < q; ++b) {
          r[b] += res * a[b];
          for (uint32_t d = 0; d <= b; ++d) M[b * q + d] += w * a[b] * a[d];
        }
      }
      for (uint32_t b = 0; b < q; ++b) { M[b * q + b] += 1.0; for (uint32_t d = b + 1; d < q; ++d) M[b * q + d] = M[d * q + b]; }
      double uu = 0.0;
      for (uint32_t b = 0; b < q; ++b) uu += ut[b] * ut[b];
      const double pd = devt + uu, tol = 1e-13 * (fabs(pd) + 1.0);
      if (!(pd <= pd_acc + tol)) {                   /* worse than the accepted point (or not finite): halve the step */
        int stuck_t = 1, stuck_a = 1;
        for (uint32_t b = 0; b < q; ++b) {
          const double half = 0.5 * (ut[b] + u[b]);
          if (half != ut[b]) stuck_t = 0;
          if (half != u[b]) stuck_a = 0;
          ut[b] = half;
        }
        if (stuck_t || stuck_a) break;
        continue;
      }
      if (gs_chol(M, q, Lm)) return INFINITY;
      logdet = 0.0;
      for (uint32_t b = 0; b < q; ++b) logdet += 2.0 * log(Lm[b * q + b]);
      const int conv = fabs(pd_acc - pd) <= tol;     /* pd_acc = inf on the first pass: never */
      for (uint32_t b = 0; b < q; ++b) u[b] = ut[b];
      dev = devt; pd_acc = pd;
      if (conv) break;
      double un[GS_MAX_Q], t[GS_MAX_Q];
      for (uint32_t b = 0; b < q; ++b) { /* forward, then back substitution */
        double v = r[b];
        for (uint32_t k2 = 0; k2 < b; ++k2) v -= Lm[b * q + k2] * t[k2];
        t[b] = v / Lm[b * q + b];
      }
      for (int b = (int)q - 1; b >= 0; --b) {
        double v = t[b];
        for (uint32_t k2 = (uint32_t)b + 1; k2 < q; ++k2) v -= Lm[k2 * q + (uint32_t)b] * un[k2];
        un[b] = v / Lm[(uint32_t)b * q + (uint32_t)b];
      }
      int same = 1;
      for (uint32_t b = 0; b < q; ++b) if (un[b] != u[b]) same = 0;
      if (same) break;
      for (uint32_t b = 0; b < q; ++b) ut[b] = un[b];
    }
    if (!(pd_acc < INFINITY)) return INFINITY;       /* not even the prior mode evaluates */
    double uu = 0.0;
    for (uint32_t b = 0; b < q; ++b) uu += u[b] * u[b];
    total += dev + uu + logdet;
    if (m->nagq > 1) {
      /* Adaptive Gauss-Hermite quadrature over the q random effects of the cluster (the reference accepts nagq > 1 for a
       * single grouping with <= 3 of them, engine-contract validate.rs:918-941; Pinheiro & Bates 1995, Pinheiro & Chao
       * 2006): the k^q tensor rule for the standard normal weight, centred at the mode and scaled by the factor of the
       * curvature there, u_j = u~ + Lm^-T z_j with Lm Lm' = I + A'WA.  As a ratio to the Laplace term (q = 1: lme4's
       * glmerAGQ, fit_glmm.c):  - 2 log sum_j w_j exp(-(d_c(u_j) - d_c(u~)) / 2 + |z_j|^2 / 2),
       * d_c(v) = sum_i devres_i(eta0_i + a_i'v) + |v|^2.  Node index 0 of the grid runs fastest. */
      const uint32_t k = m->nagq;
      const double d0 = dev + uu;
      uint32_t nn = 1;
      for (uint32_t b = 0; b < q; ++b) nn *= k;
      double mult = 0.0;
      for (uint32_t node = 0; node < nn; ++node) {
        double z[GS_MAX_Q], xs[GS_MAX_Q], v[GS_MAX_Q], wj = 1.0, zz = 0.0;
        uint32_t rem = node, centre = 1;
        for (uint32_t b = 0; b < q; ++b) {
          const uint32_t ib = rem % k;
          rem /= k;
          z[b] = m->gz[ib]; wj *= m->gw[ib]; zz += z[b] * z[b];
          if (2 * ib + 1 != k) centre = 0;
        }
        if (centre) { mult += wj; continue; }
        for (int b = (int)q - 1; b >= 0; --b) { /* Lm' xs = z */
          double t = z[b];
          for (uint32_t k2 = (uint32_t)b + 1; k2 < q; ++k2) t -= Lm[k2 * q + (uint32_t)b] * xs[k2];
          xs[b] = t / Lm[(uint32_t)b * q + (uint32_t)b];
        }
        double vv = 0.0;
        for (uint32_t b = 0; b < q; ++b) { v[b] = u[b] + xs[b]; vv += v[b] * v[b]; }
        double ds = 0.0;
        for (uint32_t kk = lo; kk < hi; ++kk) {
          uint32_t i = m->rows[kk];
          double zr[GS_MAX_Q];
          zr[0] = 1.0;
          for (uint32_t d = 1; d < q; ++d) zr[d] = (double)m->x[(size_t)m->cols[d - 1] * m->ld + i];
          double au = 0.0;
          for (uint32_t b = 0; b < q; ++b) {
            double sacc = 0.0;
            for (uint32_t d = b; d < q; ++d) sacc += T[d][b] * zr[d];
            au += sacc * v[b];
          }
          double w, zw, dv;
          mcpo_fam_eval(m->fam, m->eta0[i] + au, (double)m->y[i], &w, &zw, &dv);
          ds += dv;
        }
        mult += wj * exp(-0.5 * (ds + vv - d0) + 0.5 * zz);
      }
      total -= 2.0 * log(mult);
    }
  }
  return isfinite(total) ? total : INFINITY;
}

static void gs_diag_mask(const gs_model* m, uint8_t* is_diag) {
  uint32_t k = 0;
  for (uint32_t c = 0; c < m->q; ++c)
    for (uint32_t r = c; r < m->q; ++r) is_diag[k++] = r == c;
}

static double gs_f(gs_model* m, const uint8_t* is_diag, double* par, uint32_t* n_eval) {
  for (uint32_t k = 0; k < m->nth; ++k) {
    if (is_diag[k]) par[k] = par[k] < 0.0 ? 0.0 : (par[k] > GS_THETA_MAX ? GS_THETA_MAX : par[k]);
    else par[k] = par[k] < -GS_THETA_MAX ? -GS_THETA_MAX : (par[k] > GS_THETA_MAX ? GS_THETA_MAX : par[k]);
  }
  for (uint32_t k = m->nth; k < m->nth + m->p; ++k) par[k] = par[k] < -50.0 ? -50.0 : (par[k] > 50.0 ? 50.0 : par[k]);
  ++*n_eval;
  return gs_deviance(m, par);
}

static double gs_nelder_mead(gs_model* m, const uint8_t* is_diag, double* x0, double step, double ftol, double xtol,
                             uint32_t* n_eval) {
  const uint32_t d = m->nth + m->p;
  double sx[GS_MAX_DIM + 1][GS_MAX_DIM], fv[GS_MAX_DIM + 1];
  memcpy(sx[0], x0, sizeof(double) * d);
  fv[0] = gs_f(m, is_diag, sx[0], n_eval);
  for (uint32_t i = 0; i < d; ++i) {
    memcpy(sx[i + 1], x0, sizeof(double) * d);
    sx[i + 1][i] += step;
    fv[i + 1] = gs_f(m, is_diag, sx[i + 1], n_eval);
  }
  while (*n_eval < GS_MAX_EVAL) {
    uint32_t lo = 0, hi = 0, nh;
    for (uint32_t i = 1; i <= d; ++i) { if (fv[i] < fv[lo]) lo = i; if (fv[i] > fv[hi]) hi = i; }
    nh = lo;
    for (uint32_t i = 0; i <= d; ++i) if (i != hi && fv[i] > fv[nh]) nh = i;
    double size = 0.0;
    for (uint32_t i = 0; i <= d; ++i)
      for (uint32_t k = 0; k < d; ++k) size = fmax(size, fabs(sx[i][k] - sx[lo][k]));
    if (isfinite(fv[hi]) && fv[hi] - fv[lo] <= ftol && size <= xtol) break;
    double cen[GS_MAX_DIM], xr[GS_MAX_DIM], xe[GS_MAX_DIM];
    for (uint32_t k = 0; k < d; ++k) {
      double s = 0.0;
      for (uint32_t i = 0; i <= d; ++i) if (i != hi) s += sx[i][k];
      cen[k] = s / (double)d;
      xr[k] = cen[k] + (cen[k] - sx[hi][k]);
    }
    double fr = gs_f(m, is_diag, xr, n_eval);
    if (fr < fv[lo]) {
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 2.0 * (xr[k] - cen[k]);
      double fe = gs_f(m, is_diag, xe, n_eval);
      if (fe < fr) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fe; }
      else { memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr; }
    } else if (fr < fv[nh]) {
      memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr;
    } else {
      int outside = fr < fv[hi];
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 0.5 * ((outside ? xr[k] : sx[hi][k]) - cen[k]);
      double fc = gs_f(m, is_diag, xe, n_eval);
      if (fc < (outside ? fr : fv[hi])) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fc; }
      else {
        for (uint32_t i = 0; i <= d; ++i) {
          if (i == lo) continue;
          for (uint32_t k = 0; k < d; ++k) sx[i][k] = sx[lo][k] + 0.5 * (sx[i][k] - sx[lo][k]);
          fv[i] = gs_f(m, is_diag, sx[i], n_eval);
        }
      }
    }
  }
  uint32_t lo = 0;
  for (uint32_t i = 1; i <= d; ++i) if (fv[i] < fv[lo]) lo = i;
  memcpy(x0, sx[lo], sizeof(double) * d);
  return fv[lo];
}

typedef struct gs_cb { gs_model* m; const uint8_t* is_diag; uint32_t* n_eval; } gs_cb;
static double gs_probe_cb(void* ctx, double* par) {  /* a probe of the stationarity check: counted like any evaluation */
  gs_cb* c = (gs_cb*)ctx;
  return gs_f(c->m, c->is_diag, par, c->n_eval);
}
static double gs_dev_cb(void* ctx, double* par) {
  uint32_t scratch = 0;
  gs_cb* c = (gs_cb*)ctx;
  return gs_f(c->m, c->is_diag, par, &scratch);
}

/* theta0: nth start values (floored at 0.01 when given) or NULL for the blind start; beta0: p start values;
 * mode: MCPO_MODE_HESSIAN selects the finite-difference Hessian Wald covariance (fit_glmm.c) */
int mcpo_glmm_slopes_fit_ids(int fam, const float* x, const float* y, const uint32_t* ids, uint32_t n, uint32_t ld,
                             uint32_t p, uint32_t ncl, uint32_t q, const uint32_t* slope_cols, const double* theta0,
                             const double* beta0, uint32_t mode, mcpo_gen_fit* f) {
  if (p > MCPO_MAX_P || q < 1 || q > GS_MAX_Q) return MCP_ERR_UNSUPPORTED;
  gs_model m;
  memset(&m, 0, sizeof m);
  m.fam = fam; m.n = n; m.ld = ld; m.p = p; m.ncl = ncl; m.q = q; m.nth = q * (q + 1) / 2; m.x = x; m.y = y;
  for (uint32_t d = 0; d + 1 < q; ++d) m.cols[d] = slope_cols[d];
  m.nagq = (mode >> 8) & 0xffu;
  if (m.nagq == 0) m.nagq = 1;
  if (m.nagq > 25 || !(m.nagq & 1u)) return MCP_ERR_INVALID_SPEC;
  if (m.nagq > 1) mcpo_gh_rule(m.nagq, m.gz, m.gw);
  m.start = (uint32_t*)calloc(ncl + 2, sizeof(uint32_t));
  m.rows = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  m.eta0 = (double*)malloc(sizeof(double) * (n ? n : 1));
  m.u = (double*)calloc((size_t)(ncl ? ncl : 1) * q, sizeof(double));
  m.Lm = (double*)calloc((size_t)(ncl ? ncl : 1) * q * q, sizeof(double));
  if (!m.start || !m.rows || !m.eta0 || !m.u || !m.Lm) { gs_free(&m); return MCP_ERR_DEVICE; }
  for (uint32_t i = 0; i < n; ++i) { if (ids[i] >= ncl) { gs_free(&m); return MCP_ERR_INVALID_SPEC; } m.start[ids[i] + 1] += 1; }
  for (uint32_t c = 0; c < ncl; ++c) m.start[c + 1] += m.start[c];
  {
    uint32_t* fill = (uint32_t*)calloc(ncl ? ncl : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) { uint32_t c = ids[i]; m.rows[m.start[c] + fill[c]++] = i; }
    free(fill);
  }
  uint8_t is_diag[6];
  gs_diag_mask(&m, is_diag);
  const uint32_t d = m.nth + p;
  double par[GS_MAX_DIM];
  for (uint32_t k = 0; k < m.nth; ++k) {
    par[k] = theta0 ? theta0[k] : (is_diag[k] ? 1.0 : 0.0);
    if (theta0 && par[k] < 0.01) par[k] = 0.01;
  }
  for (uint32_t c = 0; c < p; ++c) par[m.nth + c] = beta0 ? beta0[c] : 0.0;
  memset(f, 0, sizeof *f);
  uint32_t n_eval = 0;
  double best = gs_nelder_mead(&m, is_diag, par, 0.25, 1e-10, 1e-7, &n_eval);
  /* restarts only for a point the stationarity check rejects (oracle_internal.h) */
  uint8_t lower0[GS_MAX_DIM] = {0};
  for (uint32_t k = 0; k < m.nth; ++k) lower0[k] = is_diag[k];
  gs_cb probe = {&m, is_diag, &n_eval};
  const int stationary = mcpo_stationary(gs_probe_cb, &probe, par, m.nth + p, lower0, best);
  for (int restart = 0; !stationary && restart < 4 && n_eval < GS_MAX_EVAL; ++restart) {
    double prev = best;
    best = gs_nelder_mead(&m, is_diag, par, restart == 0 ? 0.05 : 0.01, 1e-10, 1e-7, &n_eval);
    if (!(prev - best > 1e-8)) break;
  }
  double cur = gs_f(&m, is_diag, par, &n_eval);
  for (uint32_t k = 0; k < m.nth; ++k) {
    if (!is_diag[k] || par[k] == 0.0 || par[k] > 1e-4) continue;
    double keep = par[k];
    par[k] = 0.0;
    double f0 = gs_f(&m, is_diag, par, &n_eval);
    if (f0 <= cur + 1e-8) cur = f0; else par[k] = keep;
  }
  cur = gs_f(&m, is_diag, par, &n_eval);
  f->n_eval = n_eval; f->n_theta = m.nth; f->p = p; f->dev = cur; f->sigma_sq = 1.0;
  memcpy(f->theta, par, sizeof(double) * m.nth);
  memcpy(f->beta, par + m.nth, sizeof(double) * p);
  /* RX'RX = X'WX - sum_c C_c' M_c^-1 C_c,  C_c = A_c' W X_c  (q x p),  M_c = Lm Lm' */
  int ok = isfinite(cur) && n_eval < GS_MAX_EVAL;
  double A[MCPO_MAX_P * MCPO_MAX_P], T[GS_MAX_Q][GS_MAX_Q];
  memset(A, 0, sizeof A);
  gs_template(&m, par, T);
  for (uint32_t c = 0; c < ncl && ok; ++c) {
    double C[GS_MAX_Q * MCPO_MAX_P];
    memset(C, 0, sizeof C);
    const double* u = m.u + (size_t)c * q;
    for (uint32_t k = m.start[c]; k < m.start[c + 1]; ++k) {
      uint32_t i = m.rows[k];
      double z[GS_MAX_Q], a[GS_MAX_Q], au = 0.0;
      z[0] = 1.0;
      for (uint32_t dd = 1; dd < q; ++dd) z[dd] = (double)x[(size_t)m.cols[dd - 1] * ld + i];
      for (uint32_t b = 0; b < q; ++b) {
        double s = 0.0;
        for (uint32_t dd = b; dd < q; ++dd) s += T[dd][b] * z[dd];
        a[b] = s; au += s * u[b];
      }
      double w, zw, dv;
      mcpo_fam_eval(fam, m.eta0[i] + au, (double)y[i], &w, &zw, &dv);
      for (uint32_t r = 0; r < p; ++r) {
        double xr = r == 0 ? 1.0 : (double)x[(size_t)r * ld + i];
        for (uint32_t b = 0; b < q; ++b) C[b * p + r] += w * a[b] * xr;
        for (uint32_t s = 0; s <= r; ++s) A[r * p + s] += w * xr * (s == 0 ? 1.0 : (double)x[(size_t)s * ld + i]);
      }
    }
    const double* Lm = m.Lm + (size_t)c * q * q;
    for (uint32_t r = 0; r < p; ++r) { /* D = Lm^-1 C column r */
      double t[GS_MAX_Q];
      for (uint32_t b = 0; b < q; ++b) {
        double v = C[b * p + r];
        for (uint32_t k2 = 0; k2 < b; ++k2) v -= Lm[b * q + k2] * t[k2];
        t[b] = v / Lm[b * q + b];
      }
      for (uint32_t b = 0; b < q; ++b) C[b * p + r] = t[b];
    }
    for (uint32_t r = 0; r < p; ++r)
      for (uint32_t s = 0; s <= r; ++s) {
        double v = 0.0;
        for (uint32_t b = 0; b < q; ++b) v += C[b * p + r] * C[b * p + s];
        A[r * p + s] -= v;
      }
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
  for (uint32_t k = 0; k < m.nth; ++k) if (is_diag[k] && par[k] < GS_THETA_ZERO) singular = 1;
  (void)singular;
  if (ok && (mode & MCPO_MODE_HESSIAN)) {
    /* the stencil runs over the FREE covariance parameters and beta: a diagonal pinned at 0 sits at an even point of the
     * deviance (flipping the sign of a column of Lambda only relabels u), so its cross-derivatives vanish and leaving it
     * out changes nothing in the beta block; the coupling with the remaining parameters is what "rx" ignores.  A block
     * that does not factor keeps "rx". */
    uint8_t th_free[6];
    for (uint32_t k = 0; k < m.nth; ++k) th_free[k] = !(is_diag[k] && par[k] < GS_THETA_ZERO);
    {  /* EXPERIMENT KNOB HESS_PINNED (oracle_internal.h): 1 = keep pinned diagonals in the stencil (steps past the bound are
          clamped onto it), 2 = a singular fit keeps "rx" (vcov.merMod), 4 = a singular fit differences over beta alone,
          5 = a pinned diagonal takes the rest of its template column out as well */
      const int hp = (int)mcpo_knob("HESS_PINNED", 0.0);
      int singular = 0;
      for (uint32_t k = 0; k < m.nth; ++k) singular |= !th_free[k];
      if (hp == 1) for (uint32_t k = 0; k < m.nth; ++k) th_free[k] = 1;
      if (hp == 2 && singular) goto hess_done;
      if (hp == 4 && singular) for (uint32_t k = 0; k < m.nth; ++k) th_free[k] = 0;
      if (hp == 5 && singular) {
        uint32_t k = 0;
        for (uint32_t c = 0; c < q; ++c) {
          const int pinned = !th_free[k];
          for (uint32_t r = c; r < q; ++r, ++k) if (pinned) th_free[k] = 0;
        }
      }
    }
    gs_cb cb = {&m, is_diag, NULL};
    double lh[MCPO_MAX_P * MCPO_MAX_P];
    if (mcpo_fd_hessian_factor(gs_dev_cb, &cb, par, m.nth, p, th_free, cur, lh) == 0) memcpy(f->L, lh, sizeof(double) * p * p);
  hess_done:;
  }
  f->converged = ok;
  uint32_t comp = 0;
  for (uint32_t k = 0; k < m.nth; ++k) {
    if (!is_diag[k]) continue;
    if (ok && par[k] < GS_THETA_ZERO) f->pinned |= 1ull << comp;
    if (par[k] > GS_THETA_MAX - 1e-3) f->at_upper = 1;
    ++comp;
  }
  f->boundary_hit = !ok ? 0 : (f->at_upper ? 2 : (f->pinned ? 1 : 0));
  if (f->at_upper) f->converged = 0;
  (void)d;
  gs_free(&m);
  return MCP_OK;
}

/* provided-bytes entry: vars_out = Lambda Lambda' (q x q, row-major 3 x 3) */
int mcpo_glmm_slopes_fit(int family, const float* x, const float* y, const uint32_t* cluster_ids, uint32_t n, uint32_t ld,
                         uint32_t p, uint32_t n_clusters, uint32_t q, const uint32_t* slope_cols, const double* theta_start,
                         const double* beta_start, double* beta, double* se, double* theta_hat, double* vars_out,
                         uint8_t* converged, uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval, uint32_t mode) {
  mcpo_gen_fit f;
  int rc = mcpo_glmm_slopes_fit_ids(family, x, y, cluster_ids, n, ld, p, n_clusters, q, slope_cols, theta_start, beta_start, mode, &f);
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
  double T[3][3] = {{0}};
  uint32_t k = 0;
  for (uint32_t c = 0; c < q; ++c)
    for (uint32_t r = c; r < q; ++r) T[r][c] = f.theta[k++];
  for (uint32_t a = 0; a < 3; ++a)
    for (uint32_t b = 0; b < 3; ++b) {
      double s = 0.0;
      for (uint32_t c = 0; c < 3; ++c) s += T[a][c] * T[b][c];
      vars_out[a * 3 + b] = s;
    }
  return MCP_OK;
}
