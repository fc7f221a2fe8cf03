/*
 * oracle/fit_lmm_general.c -- CPU ORACLE (test infrastructure): the general linear mixed model by profiled
 * REML -- random slopes on the primary grouping, crossed and nested extra groupings, slopes on those.
 *
 * The reference routes every Mle fit through the third-party crate glmm 0.1.2 (`loop_advanced::fit_on`,
 * batch.rs:1333-1340) with bobyqa 0.1.1 as optimiser; neither is in the tree.  What IS readable:
 *   crates/engine-core/src/mixed_workspace.rs:57-110 -- the model handed to the solver: one term per
 *     grouping (primary first, extras in declaration order), each with an intercept and the listed slope
 *     columns of the design;
 *   mixed_workspace.rs:112-198 -- theta is the column-major vech of one lower-triangular q x q template
 *     per term (lme4's parametrisation: b = Lambda_theta u, u ~ N(0, sigma^2 I)); the truth start is
 *     chol(diag(tau) R diag(tau)) since the DGP has sigma = 1;
 *   web/documentation/about/acknowledgements.md:16-34 -- the fit is lme4's penalised least squares:
 *     with A = Lambda' Z'Z Lambda + I = L L', cu = L^-1 Lambda' Z'y, RZX = L^-1 Lambda' Z'X,
 *     RX'RX = X'X - RZX'RZX, the profiled REML criterion is
 *       d(theta) = 2 log|L| + 2 log|RX| + (n - p) (1 + log(2 pi r^2 / (n - p))),  r^2 = |y|^2 - |cu|^2 - |cb|^2
 *     (Bates, Maechler, Bolker, Walker 2015, J. Stat. Softw. 67(1), eqs. 39-41), beta-hat = RX^-1 cb,
 *     Var(beta-hat) = sigma-hat^2 (RX'RX)^-1, sigma-hat^2 = r^2 / (n - p).
 *
 * Factorisation order here: [blocks of the ELIMINATED term | columns of the other terms | X | y].  One term's levels
 * (one q x q block each) are eliminated analytically, the other terms form one dense block.  Which term: the likelihood
 * does not depend on it, the cost does -- `elim` (mcpo_elim_term: the grouping with the most random-effect
 * columns among those as wide as the primary one) names it; theta stays in
 * declaration order.  The device kernel (lmm_general_impl.hpp) takes the same choice and the same operation order.
 * The minimiser is a bound-constrained Nelder-Mead simplex (diagonal theta >= 0, off-diagonals free) with
 * restarts -- lme4 ships the same family of optimiser as `Nelder_Mead`; the reference uses BOBYQA.
 *
 * PARITY STATUS: beta, Wald z, variance components, RE correlations and sigma^2 pinned to lme4 1.1.38
 * (validation/data/lmm_*.golden.rds, gates of validation/tolerances.R:38-49: 1e-4 on beta / z, 1e-3 on
 * variance components).  Optimiser trajectory, blind start, boundary / pinned classification:
 * **parity unpinned** beyond those bands.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define GEN_MAX_THETA 12
#define GEN_THETA_ZERO 1e-6
#define GEN_THETA_MAX 32.0
#define GEN_MAX_EVAL 6000
#define GEN_NEAR_ZERO 1e-30

typedef struct gen_model {
  uint32_t n, p, n_terms, n_theta;
  uint32_t q[MCPO_GEN_MAX_TERMS], levels[MCPO_GEN_MAX_TERMS], theta_off[MCPO_GEN_MAX_TERMS];
  uint32_t E, n_rest, rest[MCPO_GEN_MAX_TERMS];  /* eliminated term; the other terms in declaration order */
  uint32_t L0, q0, R, W;     /* eliminated term's levels / width; rest width R; W = R + p + 1 */
  double* gzz;               /* L0 x q0 x q0      per-level Z0'Z0 */
  double* gzw;               /* L0 x q0 x W       per-level Z0'[Zr | X | y] */
  double* gww;               /* W x W             [Zr | X | y]'[Zr | X | y] */
  double *S, *B, *zw;        /* scratch: W x W, q0 x W, q0 x W */
} gen_model;

typedef struct gen_eval {
  int ok;
  double dev, sigma_sq;
  double beta[MCPO_MAX_P];
  double L[MCPO_MAX_P * MCPO_MAX_P];   /* lower Cholesky factor of RX'RX, row-major */
} gen_eval;

static void gen_free(gen_model* m) {
  free(m->gzz); free(m->gzw); free(m->gww); free(m->S); free(m->B); free(m->zw);
  memset(m, 0, sizeof *m);
}

/* sufficient statistics of rows [0, n): every quantity the criterion needs, independent of theta */
static int gen_build(gen_model* m, const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                     const mcpo_re_term* terms, uint32_t n_terms, uint32_t elim) {
  memset(m, 0, sizeof *m);
  if (n_terms == 0 || n_terms > MCPO_GEN_MAX_TERMS || p > MCPO_MAX_P || elim >= n_terms) return MCP_ERR_UNSUPPORTED;
  m->n = n; m->p = p; m->n_terms = n_terms; m->E = elim;
  const uint32_t E = elim;
  uint32_t off = 0, roff[MCPO_GEN_MAX_TERMS] = {0};
  for (uint32_t t = 0; t < n_terms; ++t) {
    m->q[t] = terms[t].q; m->levels[t] = terms[t].n_levels; m->theta_off[t] = off;
    off += terms[t].q * (terms[t].q + 1) / 2;
    if (t != E) { m->rest[m->n_rest++] = t; roff[t] = m->R; m->R += terms[t].q * terms[t].n_levels; }
  }
  if (off > GEN_MAX_THETA) return MCP_ERR_UNSUPPORTED;
  m->n_theta = off;
  m->L0 = terms[E].n_levels; m->q0 = terms[E].q; m->W = m->R + p + 1;
  const uint32_t q0 = m->q0, W = m->W;
  m->gzz = (double*)calloc((size_t)m->L0 * q0 * q0 + 1, sizeof(double));
  m->gzw = (double*)calloc((size_t)m->L0 * q0 * W + 1, sizeof(double));
  m->gww = (double*)calloc((size_t)W * W, sizeof(double));
  m->S = (double*)malloc(sizeof(double) * W * W);
  m->B = (double*)malloc(sizeof(double) * q0 * W);
  m->zw = (double*)malloc(sizeof(double) * q0 * W);
  if (!m->gzz || !m->gzw || !m->gww || !m->S || !m->B || !m->zw) { gen_free(m); return MCP_ERR_DEVICE; }
  /* sparse row of W = [Zr | X | y]: index / value pairs */
  uint32_t idx[MCPO_GEN_MAX_TERMS * (1 + MCP_MAX_RE_SLOPES) + MCPO_MAX_P + 1];
  double val[MCPO_GEN_MAX_TERMS * (1 + MCP_MAX_RE_SLOPES) + MCPO_MAX_P + 1], z0[1 + MCP_MAX_RE_SLOPES];
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t nz = 0;
    for (uint32_t r = 0; r < m->n_rest; ++r) {
      const uint32_t t = m->rest[r];
      uint32_t lvl = terms[t].ids[i];
      if (lvl >= terms[t].n_levels) { gen_free(m); return MCP_ERR_INVALID_SPEC; }
      for (uint32_t d = 0; d < terms[t].q; ++d) {
        idx[nz] = roff[t] + lvl * terms[t].q + d;
        val[nz++] = d == 0 ? 1.0 : (double)x[(size_t)terms[t].cols[d - 1] * ld + i];
      }
    }
    for (uint32_t a = 0; a < p; ++a) { idx[nz] = m->R + a; val[nz++] = a == 0 ? 1.0 : (double)x[(size_t)a * ld + i]; }
    idx[nz] = m->R + p; val[nz++] = (double)y[i];
    uint32_t c = terms[E].ids[i];
    if (c >= m->L0) { gen_free(m); return MCP_ERR_INVALID_SPEC; }
    for (uint32_t d = 0; d < q0; ++d) z0[d] = d == 0 ? 1.0 : (double)x[(size_t)terms[E].cols[d - 1] * ld + i];
    for (uint32_t a = 0; a < q0; ++a) {
      for (uint32_t b = 0; b < q0; ++b) m->gzz[((size_t)c * q0 + a) * q0 + b] += z0[a] * z0[b];
      double* row = m->gzw + ((size_t)c * q0 + a) * W;
      for (uint32_t k = 0; k < nz; ++k) row[idx[k]] += z0[a] * val[k];
    }
    for (uint32_t k = 0; k < nz; ++k)
      for (uint32_t l = 0; l < nz; ++l) m->gww[(size_t)idx[k] * W + idx[l]] += val[k] * val[l];
  }
  return MCP_OK;
}

/* template of term t: T[r][c] (lower), column-major vech in theta (mixed_workspace.rs:189-196) */
static void gen_template(const gen_model* m, uint32_t t, const double* theta, double T[3][3]) {
  const double* th = theta + m->theta_off[t];
  uint32_t q = m->q[t], k = 0;
  for (uint32_t r = 0; r < 3; ++r) for (uint32_t c = 0; c < 3; ++c) T[r][c] = 0.0;
  for (uint32_t c = 0; c < q; ++c)
    for (uint32_t r = c; r < q; ++r) T[r][c] = th[k++];
}

/* the "rest" columns of a W-wide vector v (every term but the eliminated one): v <- Lambda_r' v, in place per level block */
static void gen_lambda_t_cols(const gen_model* m, const double* theta, double* v, size_t stride) {
  uint32_t base = 0;
  for (uint32_t r = 0; r < m->n_rest; ++r) {
    const uint32_t t = m->rest[r];
    double T[3][3];
    gen_template(m, t, theta, T);
    uint32_t q = m->q[t];
    for (uint32_t l = 0; l < m->levels[t]; ++l) {
      double in[3], out[3];
      for (uint32_t d = 0; d < q; ++d) in[d] = v[(size_t)(base + l * q + d) * stride];
      for (uint32_t b = 0; b < q; ++b) { /* (Lambda' v)_b = sum_{d >= b} T[d][b] v_d */
        double s = 0.0;
        for (uint32_t d = b; d < q; ++d) s += T[d][b] * in[d];
        out[b] = s;
      }
      for (uint32_t d = 0; d < q; ++d) v[(size_t)(base + l * q + d) * stride] = out[d];
    }
    base += m->levels[t] * q;
  }
}

static void gen_evaluate(gen_model* m, const double* theta, gen_eval* e) {
  const uint32_t W = m->W, R = m->R, p = m->p, q0 = m->q0;
  double* S = m->S;
  e->ok = 0; e->dev = INFINITY;
  /* S = [Lambda_r' Gww Lambda_r (+ I on the rest block)] -- rows then columns */
  memcpy(S, m->gww, sizeof(double) * W * W);
  for (uint32_t j = 0; j < W; ++j) gen_lambda_t_cols(m, theta, S + j, W);          /* columns: act on the row index */
  for (uint32_t i = 0; i < W; ++i) gen_lambda_t_cols(m, theta, S + (size_t)i * W, 1); /* rows: act on the column index */
  for (uint32_t r = 0; r < R; ++r) S[(size_t)r * W + r] += 1.0;
  /* log-determinant: one log() per primary level (of the product of its squared pivots, each >= 1), summed in
   * level order; the dense columns multiply their squared pivots up four at a time -- the grouping the device
   * kernel uses (it would otherwise spend most of an evaluation in log()) */
  double logdet = 0.0, pd = 1.0, T0[3][3];
  uint32_t npd = 0;
#define GEN_PIVOT(dv) do { pd *= (dv); if (++npd == 4u) { logdet += log(pd); pd = 1.0; npd = 0; } } while (0)
  gen_template(m, m->E, theta, T0);
  for (uint32_t c = 0; c < m->L0; ++c) {
    /* A = T0' Gzz T0 + I, lower Cholesky in place */
    const double* G = m->gzz + (size_t)c * q0 * q0;
    double A[3][3], Lc[3][3], pdc = 1.0;
    for (uint32_t a = 0; a < q0; ++a)
      for (uint32_t b = 0; b < q0; ++b) {
        double s = 0.0;
        for (uint32_t r = a; r < q0; ++r)
          for (uint32_t cc = b; cc < q0; ++cc) s += T0[r][a] * G[r * q0 + cc] * T0[cc][b];
        A[a][b] = s + (a == b ? 1.0 : 0.0);
      }
    for (uint32_t j = 0; j < q0; ++j) {
      double d = A[j][j];
      for (uint32_t k = 0; k < j; ++k) d -= Lc[j][k] * Lc[j][k];
      if (!(d > 0.0)) return;
      Lc[j][j] = sqrt(d);
      pdc *= d;
      for (uint32_t i = j + 1; i < q0; ++i) {
        double v = A[i][j];
        for (uint32_t k = 0; k < j; ++k) v -= Lc[i][k] * Lc[j][k];
        Lc[i][j] = v / Lc[j][j];
      }
    }
    logdet += log(pdc);
    /* B = Lc^-1 T0' [Gzw Lambda_r | Gzx | Gzy]  (q0 x W);  S -= B'B */
    double* zw = m->zw;
    memcpy(zw, m->gzw + (size_t)c * q0 * W, sizeof(double) * q0 * W);
    for (uint32_t a = 0; a < q0; ++a) gen_lambda_t_cols(m, theta, zw + (size_t)a * W, 1);
    double* B = m->B;
    for (uint32_t j = 0; j < W; ++j) {
      double tv[3];
      for (uint32_t a = 0; a < q0; ++a) { /* T0' column */
        double s = 0.0;
        for (uint32_t r = a; r < q0; ++r) s += T0[r][a] * zw[(size_t)r * W + j];
        tv[a] = s;
      }
      for (uint32_t a = 0; a < q0; ++a) {
        double v = tv[a];
        for (uint32_t k = 0; k < a; ++k) v -= Lc[a][k] * B[(size_t)k * W + j];
        B[(size_t)a * W + j] = v / Lc[a][a];
      }
    }
    for (uint32_t a = 0; a < q0; ++a) {
      const double* br = B + (size_t)a * W;
      for (uint32_t i = 0; i < W; ++i) {
        if (br[i] == 0.0) continue;
        double bi = br[i];
        double* srow = S + (size_t)i * W;
        for (uint32_t j = 0; j <= i; ++j) srow[j] -= bi * br[j];
      }
    }
  }
  /* dense Cholesky over [rest | X] of the lower triangle of S; the y row is the running right-hand side */
  const uint32_t M = R + p;
  for (uint32_t j = 0; j < M; ++j) {
    double d = S[(size_t)j * W + j];
    for (uint32_t k = 0; k < j; ++k) d -= S[(size_t)j * W + k] * S[(size_t)j * W + k];
    if (!(d > 0.0) || !isfinite(d)) return;
    double ljj = sqrt(d);
    S[(size_t)j * W + j] = ljj;
    GEN_PIVOT(d);   /* log|L|^2 over the rest block and log|RX|^2 over the X block */
    for (uint32_t i = j + 1; i < W; ++i) {
      double v = S[(size_t)i * W + j];
      for (uint32_t k = 0; k < j; ++k) v -= S[(size_t)i * W + k] * S[(size_t)j * W + k];
      S[(size_t)i * W + j] = v / ljj;
    }
  }
  logdet += log(pd);
#undef GEN_PIVOT
  double pwrss = S[(size_t)M * W + M];
  for (uint32_t k = 0; k < M; ++k) pwrss -= S[(size_t)M * W + k] * S[(size_t)M * W + k];
  const double dfr = (double)m->n - (double)p;
  if (!(pwrss > 0.0) || !(dfr > 0.0)) return;
  /* rank guard on RX like the OLS arm (batch.rs:40-41) */
  double lmin = INFINITY, lmax = 0.0;
  for (uint32_t a = 0; a < p; ++a) {
    double v = fabs(S[(size_t)(R + a) * W + R + a]);
    lmin = fmin(lmin, v); lmax = fmax(lmax, v);
  }
  if (!(lmin / lmax >= 1e-12)) return;
  e->dev = logdet + dfr * (1.0 + log(2.0 * M_PI * pwrss / dfr));
  e->sigma_sq = pwrss / dfr;
  for (uint32_t a = 0; a < p; ++a)
    for (uint32_t b = 0; b < p; ++b) e->L[a * p + b] = b <= a ? S[(size_t)(R + a) * W + R + b] : 0.0;
  for (int a = (int)p - 1; a >= 0; --a) { /* beta = RX^-1 cb, RX = L' */
    double v = S[(size_t)M * W + R + (uint32_t)a];
    for (uint32_t k = (uint32_t)a + 1; k < p; ++k) v -= e->L[k * p + (uint32_t)a] * e->beta[k];
    e->beta[a] = v / e->L[(uint32_t)a * p + (uint32_t)a];
  }
  e->ok = isfinite(e->dev);
}

/* which theta entries are diagonals of their template (bounded below by 0) */
static void gen_diag_mask(const gen_model* m, uint8_t* is_diag) {
  for (uint32_t t = 0; t < m->n_terms; ++t) {
    uint32_t k = m->theta_off[t];
    for (uint32_t c = 0; c < m->q[t]; ++c)
      for (uint32_t r = c; r < m->q[t]; ++r) is_diag[k++] = r == c;
  }
}

static void gen_clamp(const gen_model* m, const uint8_t* is_diag, double* th) {
  for (uint32_t k = 0; k < m->n_theta; ++k) {
    if (is_diag[k]) th[k] = th[k] < 0.0 ? 0.0 : (th[k] > GEN_THETA_MAX ? GEN_THETA_MAX : th[k]);
    else th[k] = th[k] < -GEN_THETA_MAX ? -GEN_THETA_MAX : (th[k] > GEN_THETA_MAX ? GEN_THETA_MAX : th[k]);
  }
}

static double gen_f(gen_model* m, const uint8_t* is_diag, double* th, gen_eval* e, uint32_t* n_eval) {
  gen_clamp(m, is_diag, th);
  gen_evaluate(m, th, e);
  ++*n_eval;
  return e->ok ? e->dev : INFINITY;
}

typedef struct gen_count_cb { gen_model* m; const uint8_t* is_diag; uint32_t* n_eval; } gen_count_cb;
static double gen_probe_cb(void* ctx, double* th) {  /* a probe of the stationarity check: counted like any evaluation */
  gen_count_cb* c = (gen_count_cb*)ctx;
  gen_eval e;
  return gen_f(c->m, c->is_diag, th, &e, c->n_eval);
}

/* one Nelder-Mead run from x0 with initial edge `step`; returns the best value, x0 <- best point */
static double gen_nelder_mead(gen_model* m, const uint8_t* is_diag, double* x0, double step, double ftol, double xtol,
                              uint32_t* n_eval) {
  const uint32_t d = m->n_theta;
  double sx[GEN_MAX_THETA + 1][GEN_MAX_THETA], fv[GEN_MAX_THETA + 1];
  gen_eval e;
  memcpy(sx[0], x0, sizeof(double) * d);
  fv[0] = gen_f(m, is_diag, sx[0], &e, n_eval);
  for (uint32_t i = 0; i < d; ++i) {
    memcpy(sx[i + 1], x0, sizeof(double) * d);
    /* step away from a bound that the start sits on, otherwise toward larger values */
    sx[i + 1][i] += (is_diag[i] || x0[i] >= 0.0) ? step : -step;
    fv[i + 1] = gen_f(m, is_diag, sx[i + 1], &e, n_eval);
  }
  while (*n_eval < GEN_MAX_EVAL) {
    /* order: lo = best, hi = worst, nh = second worst */
    uint32_t lo = 0, hi = 0, nh = 0;
    for (uint32_t i = 1; i <= d; ++i) { if (fv[i] < fv[lo]) lo = i; if (fv[i] > fv[hi]) hi = i; }
    nh = lo;
    for (uint32_t i = 0; i <= d; ++i) if (i != hi && fv[i] > fv[nh]) nh = i;
    double size = 0.0;
    for (uint32_t i = 0; i <= d; ++i)
      for (uint32_t k = 0; k < d; ++k) size = fmax(size, fabs(sx[i][k] - sx[lo][k]));
    if (isfinite(fv[hi]) && fv[hi] - fv[lo] <= ftol && size <= xtol) break;
    double cen[GEN_MAX_THETA], xr[GEN_MAX_THETA], xe[GEN_MAX_THETA];
    for (uint32_t k = 0; k < d; ++k) {
      double s = 0.0;
      for (uint32_t i = 0; i <= d; ++i) if (i != hi) s += sx[i][k];
      cen[k] = s / (double)d;
      xr[k] = cen[k] + (cen[k] - sx[hi][k]);
    }
    double fr = gen_f(m, is_diag, xr, &e, n_eval);
    if (fr < fv[lo]) {
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 2.0 * (xr[k] - cen[k]);
      double fe = gen_f(m, is_diag, xe, &e, n_eval);
      if (fe < fr) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fe; }
      else { memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr; }
    } else if (fr < fv[nh]) {
      memcpy(sx[hi], xr, sizeof(double) * d); fv[hi] = fr;
    } else {
      int outside = fr < fv[hi];
      for (uint32_t k = 0; k < d; ++k) xe[k] = cen[k] + 0.5 * ((outside ? xr[k] : sx[hi][k]) - cen[k]);
      double fc = gen_f(m, is_diag, xe, &e, n_eval);
      if (fc < (outside ? fr : fv[hi])) { memcpy(sx[hi], xe, sizeof(double) * d); fv[hi] = fc; }
      else { /* shrink toward the best vertex */
        for (uint32_t i = 0; i <= d; ++i) {
          if (i == lo) continue;
          for (uint32_t k = 0; k < d; ++k) sx[i][k] = sx[lo][k] + 0.5 * (sx[i][k] - sx[lo][k]);
          fv[i] = gen_f(m, is_diag, sx[i], &e, n_eval);
        }
      }
    }
  }
  uint32_t lo = 0;
  for (uint32_t i = 1; i <= d; ++i) if (fv[i] < fv[lo]) lo = i;
  memcpy(x0, sx[lo], sizeof(double) * d);
  return fv[lo];
}

static void gen_minimise(gen_model* m, const double* theta_start, mcpo_gen_fit* f) {
  const uint32_t d = m->n_theta;
  uint8_t is_diag[GEN_MAX_THETA];
  gen_diag_mask(m, is_diag);
  double th[GEN_MAX_THETA];
  for (uint32_t k = 0; k < d; ++k) {
    th[k] = theta_start ? theta_start[k] : (is_diag[k] ? 1.0 : 0.0);  /* blind: lme4's start */
    if (theta_start && th[k] < 0.01) th[k] = 0.01;  /* THETA_TRUTH_FLOOR: every component of a given start, off-diagonals too (batch.rs:1316-1321) */
  }
  memset(f, 0, sizeof *f);
  uint32_t n_eval = 0;
  double best = gen_nelder_mead(m, is_diag, th, 0.25, 1e-10, 1e-7, &n_eval);
  /* restarts guard against a collapsed simplex: taken only for a point the stationarity check rejects (oracle_internal.h) */
  gen_count_cb probe = {m, is_diag, &n_eval};
  const int stationary = mcpo_stationary(gen_probe_cb, &probe, th, d, is_diag, best);
  for (int restart = 0; !stationary && restart < 4 && n_eval < GEN_MAX_EVAL; ++restart) {
    double prev = best;
    best = gen_nelder_mead(m, is_diag, th, restart == 0 ? 0.05 : 0.01, 1e-10, 1e-7, &n_eval);
    if (!(prev - best > 1e-8)) break;
  }
  /* snap small diagonals onto the boundary when that does not cost deviance (singular fits) */
  gen_eval e;
  double cur = gen_f(m, is_diag, th, &e, &n_eval);
  for (uint32_t k = 0; k < d; ++k) {
    if (!is_diag[k] || th[k] == 0.0 || th[k] > 1e-4) continue;
    double keep = th[k];
    th[k] = 0.0;
    gen_eval e0;
    double f0 = gen_f(m, is_diag, th, &e0, &n_eval);
    if (f0 <= cur + 1e-8) { cur = f0; e = e0; } else th[k] = keep;
  }
  gen_f(m, is_diag, th, &e, &n_eval);
  f->converged = e.ok && n_eval < GEN_MAX_EVAL;
  f->n_eval = n_eval; f->n_theta = d;
  memcpy(f->theta, th, sizeof(double) * d);
  f->dev = e.dev; f->sigma_sq = e.sigma_sq;
  memcpy(f->beta, e.beta, sizeof e.beta);
  memcpy(f->L, e.L, sizeof e.L);
  uint32_t comp = 0;
  for (uint32_t k = 0; k < d; ++k) {
    if (!is_diag[k]) continue;
    if (th[k] < GEN_THETA_ZERO) f->pinned |= 1ull << comp;
    if (th[k] > GEN_THETA_MAX - 1e-3) f->at_upper = 1;
    ++comp;
  }
  f->boundary_hit = !f->converged ? 0 : (f->at_upper ? 2 : (f->pinned ? 1 : 0));
  if (f->at_upper) f->converged = 0;
}

/* truth_theta (mixed_workspace.rs:112-198): per grouping the column-major vech of chol(diag(tau) R diag(tau)) */
uint32_t mcpo_truth_theta(const mcp_cluster* c, int include_slopes, double* theta) {
  uint32_t k = 0;
  for (uint32_t t = 0; t <= c->n_extra_groupings; ++t) {
    const mcp_re_slopes* sl = t == 0 ? &c->slopes : &c->extra[t - 1].slopes;
    double tau0 = sqrt(fmax(t == 0 ? c->tau_squared : c->extra[t - 1].tau_squared, 0.0));
    if (!include_slopes || sl->n_slopes == 0) { theta[k++] = tau0; continue; }
    uint32_t q = 1 + sl->n_slopes;
    double l[(1 + MCP_MAX_RE_SLOPES) * (1 + MCP_MAX_RE_SLOPES)];
    mcpo_slope_cholesky(sl, tau0, l);
    for (uint32_t col = 0; col < q; ++col)
      for (uint32_t row = col; row < q; ++row) theta[k++] = l[row * q + col];
  }
  return k;
}

/* the term whose levels are eliminated one block each (the fit view of lmm_general_impl.hpp / glmm_general_kernels.hip,
 * chosen by host_lowering.cpp the same way): any grouping's levels are mutually exclusive, so any term can lead; the one
 * with the most random-effect columns at the call's largest sample size leaves the smallest dense block.  Ties keep the
 * earlier term.  equal_q: only terms as wide as the primary one (the LMM kernel is instantiated on one block width). */
uint32_t mcpo_elim_term(const mcp_cluster* c, uint32_t max_n, int equal_q) {
  uint32_t E = 0, best = mcpo_n_clusters_at(c, max_n) * (1u + c->slopes.n_slopes);
  for (uint32_t t = 1; t <= c->n_extra_groupings; ++t) {
    const mcp_grouping* g = &c->extra[t - 1];
    if (equal_q && g->slopes.n_slopes != c->slopes.n_slopes) continue;
    const uint32_t cols = mcpo_extra_n_levels_at(c, t - 1, max_n) * (1u + g->slopes.n_slopes);
    if (cols > best) { best = cols; E = t; }
  }
  return E;
}

int mcpo_lmm_general_fit_terms(const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p,
                               const mcpo_re_term* terms, uint32_t n_terms, uint32_t elim, const double* theta_start, mcpo_gen_fit* f) {
  gen_model m;
  int rc = gen_build(&m, x, y, n, ld, p, terms, n_terms, elim);
  if (rc) return rc;
  gen_minimise(&m, theta_start, f);
  f->p = p;
  gen_free(&m);
  return MCP_OK;
}

/* fit provided bytes (introspect::fit_provided_data twin for the general mixed arm).
 * term_q / term_levels: per term; ids: n_terms x ld level ids; slope_cols: n_terms x MCP_MAX_RE_SLOPES design columns.
 * vars_out: per term the q x q covariance sigma^2 Lambda Lambda' (row-major, MAX 3 x 3 each). */
int mcpo_lmm_general_fit(const float* x, const float* y, uint32_t n, uint32_t ld, uint32_t p, uint32_t n_terms,
                         const uint32_t* term_q, const uint32_t* term_levels, const uint32_t* ids,
                         const uint32_t* slope_cols, uint32_t elim, const double* theta_start, double* beta, double* se, double* z_sq,
                         double* theta_hat, double* sigma_sq, double* vars_out, uint8_t* converged,
                         uint8_t* boundary_hit, uint64_t* pinned, uint32_t* n_eval) {
  if (n_terms == 0 || n_terms > MCPO_GEN_MAX_TERMS) return MCP_ERR_UNSUPPORTED;
  mcpo_re_term terms[MCPO_GEN_MAX_TERMS];
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (term_q[t] == 0 || term_q[t] > 1 + MCP_MAX_RE_SLOPES) return MCP_ERR_UNSUPPORTED;
    terms[t].q = term_q[t]; terms[t].n_levels = term_levels[t]; terms[t].ids = ids + (size_t)t * ld;
    for (uint32_t d = 0; d < MCP_MAX_RE_SLOPES; ++d) terms[t].cols[d] = slope_cols[t * MCP_MAX_RE_SLOPES + d];
  }
  mcpo_gen_fit f;
  int rc = mcpo_lmm_general_fit_terms(x, y, n, ld, p, terms, n_terms, elim, theta_start, &f);
  if (rc) return rc;
  *converged = (uint8_t)f.converged; *boundary_hit = f.boundary_hit; *pinned = f.pinned; *n_eval = f.n_eval;
  *sigma_sq = f.sigma_sq;
  memcpy(theta_hat, f.theta, sizeof(double) * f.n_theta);
  for (uint32_t j = 0; j < p; ++j) {
    /* var_j = sigma^2 |L^-1 e_j|^2 */
    double u[MCPO_MAX_P], nq = 0.0;
    for (uint32_t i = 0; i < p; ++i) {
      double acc = i == j ? 1.0 : 0.0;
      for (uint32_t t = 0; t < i; ++t) acc -= f.L[i * p + t] * u[t];
      u[i] = acc / f.L[i * p + i];
      nq += u[i] * u[i];
    }
    double var = f.sigma_sq * nq;
    beta[j] = f.converged ? f.beta[j] : NAN;
    se[j] = f.converged ? sqrt(var) : NAN;
    z_sq[j] = (f.converged && var > GEN_NEAR_ZERO) ? f.beta[j] * f.beta[j] / var : NAN;
  }
  uint32_t off = 0;
  for (uint32_t t = 0; t < n_terms; ++t) {
    double T[3][3] = {{0}};
    uint32_t q = term_q[t], k = off;
    for (uint32_t c = 0; c < q; ++c)
      for (uint32_t r = c; r < q; ++r) T[r][c] = f.theta[k++];
    off = k;
    for (uint32_t a = 0; a < 3; ++a)
      for (uint32_t b = 0; b < 3; ++b) {
        double s = 0.0;
        for (uint32_t c = 0; c < 3; ++c) s += T[a][c] * T[b][c];
        vars_out[t * 9 + a * 3 + b] = f.sigma_sq * s;
      }
  }
  return MCP_OK;
}
