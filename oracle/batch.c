/*
 * oracle/batch.c -- CPU ORACLE (test infrastructure): run_batch.
 *
 * Follows /root/reference/crates/engine-core/src/batch.rs
 *   run_batch_impl :199-776 (validation :208-330, crit tables :349, result
 *   tensors :357-385), run_one_sim :795 (OLS arm :871-1213, GLM arm :1562-2144,
 *   MLE arm :1214-1560), build_posthoc_contrasts :2189, build_crit_tables :2334,
 * and crates/engine-orchestrator/src/aggregation.rs:61-330 for the counters.
 * Sparse-factor exclusion and test_formula reduced fits (batch.rs:925-1013, :2418-2562) go through
 * exclusion.c: the OLS arm gathers the kept rows/cols of the sufficient statistics, the GLM / MLE arms
 * gather the kept design columns, and every df-dependent threshold of such a round comes from a table
 * rebuilt at p_red fitted columns.
 */
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

static __thread char g_err[512];
const char* mcpo_last_error(void) { return g_err; }
void mcpo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

/* implemented in fit_glm.c / fit_lmm.c */
int mcpo_glm_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, mcpo_red_cache* cache, uint8_t* unc, uint8_t* cor,
                 uint8_t* conv, uint8_t* overall, uint8_t* fx, double* stat_row);
int mcpo_glmm_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                  const double* crit, uint32_t crit_stride, uint8_t* unc, uint8_t* cor, uint8_t* conv, uint8_t* bh,
                  uint8_t* joint_unc, uint8_t* joint_cor, uint8_t* overall, double* tau_hat, double* stat_row,
                  uint64_t* pinned);
int mcpo_mle_arm(const mcp_spec* spec, const mcpo_ws* ws, uint32_t max_n, const uint32_t* ns, uint32_t n_ns,
                 const double* crit, uint32_t crit_stride, uint8_t* unc, uint8_t* cor, uint8_t* conv, uint8_t* bh,
                 uint8_t* joint_unc, uint8_t* joint_cor, uint8_t* fx, double* stat_row, uint64_t* pinned);

static uint32_t total_posthoc_contrasts(const mcp_spec* spec) {
  uint32_t tot = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    tot += k * (k - 1) / 2;
  }
  return tot;
}

void mcpo_counter_layout(const mcp_spec* spec, mcp_counter_layout* L) {
  uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  uint32_t npc = total_posthoc_contrasts(spec);
  uint32_t o = 0;
  L->k_unc = o; o += nt;
  L->k_cor = o; o += nt;
  L->converged = o; o += 1;
  L->overall = o; o += 1;
  L->hist_unc = o; o += nt + npc + 1;
  L->hist_cor = o; o += nt + npc + 1;
  L->posthoc_unc = o; o += npc;
  L->posthoc_cor = o; o += npc;
  L->excl_sparse = o; o += spec->n_factors;
  L->excl_sep = o; o += spec->n_factors;
  L->joint_unc = o; o += 1;
  L->joint_cor = o; o += 1;
  L->boundary_1 = o; o += 1;
  L->boundary_2 = o; o += 1;
  L->n_sims = o; o += 1;
  L->stride = o;
}

/* validation, batch.rs:208-330 */
static int validate(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns) {
  if (spec->abi_version != MCP_ABI_VERSION) { mcpo_set_error("invalid spec: abi_version"); return MCP_ERR_INVALID_SPEC; }
  if (spec->estimator == MCP_EST_MLE && !spec->cluster.present) { mcpo_set_error("invalid spec: estimator Mle requires ClusterSpec"); return MCP_ERR_INVALID_SPEC; }
  if (spec->n_posthoc && spec->estimator != MCP_EST_OLS) { mcpo_set_error("invalid spec: posthoc tests are only supported for estimator=Ols"); return MCP_ERR_INVALID_SPEC; }
  if (n_ns == 0) { mcpo_set_error("invalid spec: sample_sizes must be non-empty"); return MCP_ERR_INVALID_SPEC; }
  for (uint32_t i = 1; i < n_ns; ++i)
    if (ns[i] <= ns[i - 1]) { mcpo_set_error("invalid spec: sample_sizes must be strictly increasing; got %u then %u", ns[i - 1], ns[i]); return MCP_ERR_INVALID_SPEC; }
  if (ns[0] == 0) { mcpo_set_error("invalid spec: sample_sizes[0] must be > 0"); return MCP_ERR_INVALID_SPEC; }
  uint32_t k = spec->n_non_factor, p = 1 + k + spec->n_factor_dummies + spec->n_interactions;
  if (spec->n_effect_sizes != p) { mcpo_set_error("invalid spec: effect_sizes length %u does not match n_predictors %u", spec->n_effect_sizes, p); return MCP_ERR_INVALID_SPEC; }
  uint32_t exp_props = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    if (spec->factor_n_levels[f] < 2) { mcpo_set_error("invalid spec: factor_n_levels entries must be >= 2; got %d", spec->factor_n_levels[f]); return MCP_ERR_INVALID_SPEC; }
    exp_props += (uint32_t)spec->factor_n_levels[f];
  }
  if (spec->n_factor_proportions != exp_props) { mcpo_set_error("invalid spec: factor_proportions length"); return MCP_ERR_INVALID_SPEC; }
  for (uint32_t t = 0; t < spec->n_target_indices; ++t)
    if (spec->target_indices[t] >= p) { mcpo_set_error("invalid spec: target_indices entry %u out of range (n_predictors = %u)", spec->target_indices[t], p); return MCP_ERR_INVALID_SPEC; }
  for (uint32_t c = 0; c < 2 * spec->n_contrast_pairs; ++c)
    if (spec->contrast_pairs[c] >= p) { mcpo_set_error("invalid spec: contrast_pairs entry out of range"); return MCP_ERR_INVALID_SPEC; }
  if (k) {
    if (k > MCPO_MAX_K) { mcpo_set_error("unsupported: n_non_factor > %d", MCPO_MAX_K); return MCP_ERR_UNSUPPORTED; }
    double tmp[MCPO_MAX_K * MCPO_MAX_K];
    memcpy(tmp, spec->correlation, sizeof(double) * k * k);
    if (mcpo_chol_colmajor(tmp, k)) { mcpo_set_error("correlation matrix is not positive semi-definite"); return MCP_ERR_CORRELATION_NOT_PSD; }
  }
  if (!mcpo_spec_supported(spec) || p > MCPO_MAX_P) { mcpo_set_error("unsupported: uploads / bootstrap / slopes / extra groupings"); return MCP_ERR_UNSUPPORTED; }
  if (spec->n_factor_proportions > MCPO_MAX_LEVELS || spec->n_factors > 16) { mcpo_set_error("unsupported: too many factor levels"); return MCP_ERR_UNSUPPORTED; }
  return MCP_OK;
}

typedef struct job {
  const mcp_spec* spec;
  const uint32_t* ns;
  uint32_t n_ns, max_n, p, nt, npc;
  uint64_t base_seed, sim_offset;
  uint32_t sim_begin, sim_end;
  const double* crit;      /* n_ns x (nt + 4) */
  const double* post_corr; /* n_ns x npc */
  const double* post_contrasts; /* npc x p */
  mcp_batch_result* out;
  uint32_t flags;
  int rc;
  mcpo_red_cache cache; /* reduced-model thresholds, per worker */
} job;

static int ols_arm(job* jb, const mcpo_ws* ws, uint32_t s) {
  const mcp_spec* spec = jb->spec;
  const uint32_t p = jb->p, nt = jb->nt, n_ns = jb->n_ns, npc = jb->npc, cs = nt + 4, nf = spec->n_factors;
  mcp_batch_result* R = jb->out;
  mcpo_suff suff, sred;
  mcpo_olsfit_t fit;
  mcpo_suff_reset(&suff, p);
  uint32_t last_n = 0;
  double all_t[MCPO_MAX_P * 2];
  uint32_t prefix_counts[MCPO_MAX_LEVELS] = {0}, kept[MCPO_MAX_P];
  int32_t remap[MCPO_MAX_P];
  uint8_t flags[16] = {0};
  const int test_reduces = spec->n_fit_columns && spec->n_fit_columns < p; /* batch.rs:869 */
  for (uint32_t g = 0; g < n_ns; ++g) {
    uint32_t n = jb->ns[g];
    mcpo_suff_add_rows(&suff, ws->x, ws->y, jb->max_n, last_n, n); /* batch.rs:883-917 */
    if (nf && spec->factor_min_level_count > 0)                      /* incremental, cumulative counts */
      mcpo_update_factor_exclusions(ws->x, jb->max_n, spec, last_n, n, prefix_counts, flags);
    last_n = n;
    int any_excl = 0;
    for (uint32_t f = 0; f < nf; ++f) any_excl |= flags[f] != 0;
    const int reduced = any_excl || test_reduces;
    uint32_t p_red = p;
    const double* crow = jb->crit + (size_t)g * cs;
    const double* prow = jb->post_corr ? jb->post_corr + (size_t)g * npc : NULL;
    if (!reduced) {
      for (uint32_t c = 0; c < p; ++c) remap[c] = (int32_t)c;
      mcpo_fit_suff(&suff, &fit);
    } else { /* gather the kept rows/cols of the sufficient statistics, batch.rs:970-1013 */
      p_red = mcpo_build_exclusion_remap(spec, flags, kept, remap);
      mcpo_suff_reset(&sred, p_red);
      for (uint32_t rj = 0; rj < p_red; ++rj) {
        sred.xty[rj] = suff.xty[kept[rj]];
        for (uint32_t ri = rj; ri < p_red; ++ri) sred.xtx[ri * p_red + rj] = suff.xtx[kept[ri] * p + kept[rj]];
      }
      sred.yty = suff.yty; sred.sum_y = suff.sum_y; sred.n_rows = suff.n_rows;
      mcpo_fit_suff(&sred, &fit);
      const mcpo_red_crit* rc = mcpo_reduced_crit(&jb->cache, spec, n, p_red);
      if (!rc) return MCP_ERR_INVALID_SPEC;
      crow = rc->row;
      prow = rc->post;
    }
    size_t cell = (size_t)s * n_ns + g;
    R->converged[cell] = (uint8_t)fit.converged;
    R->boundary_hit[cell] = 0;
    uint8_t* unc = R->uncorrected + cell * nt;
    uint8_t* cor = R->corrected + cell * nt;
    for (uint32_t t = 0; t < nt; ++t) all_t[t] = NAN;
    memset(unc, 0, nt);
    if (fit.converged) { /* batch.rs:1060-1095; a dropped target keeps NaN (never significant) */
      double c[MCPO_MAX_P];
      for (uint32_t t = 0; t < spec->n_target_indices; ++t) {
        int32_t r = remap[spec->target_indices[t]];
        if (r < 0) continue;
        memset(c, 0, sizeof(double) * p_red);
        c[r] = 1.0;
        all_t[t] = mcpo_contrast_t_sq(&fit, c);
      }
      for (uint32_t q = 0; q < spec->n_contrast_pairs; ++q) {
        int32_t rp = remap[spec->contrast_pairs[2 * q]], rn = remap[spec->contrast_pairs[2 * q + 1]];
        if (rp < 0 || rn < 0) continue;
        memset(c, 0, sizeof(double) * p_red);
        c[rp] += 1.0;
        c[rn] -= 1.0;
        all_t[spec->n_target_indices + q] = mcpo_contrast_t_sq(&fit, c);
      }
      for (uint32_t t = 0; t < nt; ++t)
        if (!isnan(all_t[t]) && all_t[t] > crow[0]) unc[t] = 1;
      mcpo_apply_correction(spec->correction_method, all_t, nt, crow[0], crow + 1, cor);
    } else {
      memset(cor, 0, nt);
    }
    if (g == 0 && R->stat_capture) memcpy(R->stat_capture + (size_t)s * nt, all_t, sizeof(double) * nt);
    if (R->overall) { /* batch.rs:1118-1142: numerator df follows the fitted column count */
      uint8_t sig = 0;
      if (fit.converged) {
        double dfn = (double)((int64_t)p_red - 1), dfd = fit.df_resid;
        if (dfn >= 1.0 && fit.sst > 1e-10 && fit.rss > 0.0 && isfinite(fit.rss) && isfinite(fit.sst)) {
          double f = ((fit.sst - fit.rss) / dfn) / (fit.rss / dfd);
          if (isfinite(f) && f > crow[3 + nt]) sig = 1;
        }
      }
      R->overall[cell] = sig;
    }
    if (npc) { /* posthoc.rs:41-122; batch.rs:1164-1199 zeroes a contrast that touches a dropped column */
      uint8_t* pu = R->posthoc_unc + cell * npc;
      uint8_t* pc = R->posthoc_cor + cell * npc;
      memset(pu, 0, npc);
      memset(pc, 0, npc);
      if (fit.converged) {
        double tsq[256], rc[MCPO_MAX_P];
        double pcrit = crow[1 + nt];
        for (uint32_t c = 0; c < npc; ++c) {
          const double* full = jb->post_contrasts + (size_t)c * p;
          int dropped = 0;
          memset(rc, 0, sizeof(double) * p_red);
          for (uint32_t col = 0; col < p; ++col)
            if (full[col] != 0.0) {
              if (remap[col] >= 0) rc[remap[col]] = full[col];
              else dropped = 1;
            }
          if (dropped) memset(rc, 0, sizeof(double) * p_red);
          tsq[c] = mcpo_contrast_t_sq(&fit, rc);
          if (!isnan(tsq[c]) && tsq[c] > pcrit) pu[c] = 1;
        }
        mcpo_apply_correction(spec->correction_method, tsq, npc, pcrit, prow, pc);
      }
    }
    if (nf) memcpy(R->factor_excluded + cell * nf, flags, nf);
  }
  return MCP_OK;
}

static void* worker(void* arg) {
  job* jb = (job*)arg;
  const mcp_spec* spec = jb->spec;
  uint32_t ncl = spec->cluster.present ? mcpo_n_clusters_at(&spec->cluster, jb->max_n) : 0;
  mcpo_ws* ws = mcpo_ws_new(jb->max_n, jb->p, spec->n_non_factor, ncl);
  if (!ws || mcpo_ws_size_extras(ws, spec, jb->max_n)) { jb->rc = MCP_ERR_DEVICE; mcpo_ws_free(ws); return NULL; }
  mcp_batch_result* R = jb->out;
  const uint32_t nt = jb->nt, n_ns = jb->n_ns, cs = nt + 4;
  for (uint32_t s = jb->sim_begin; s < jb->sim_end; ++s) {
    int rc = mcpo_generate_ws(spec, jb->base_seed, jb->sim_offset + s, jb->max_n, ws);
    if (rc) { jb->rc = rc; break; }
    size_t cell0 = (size_t)s * n_ns;
    uint8_t* fx = spec->n_factors ? R->factor_excluded + cell0 * spec->n_factors : NULL;
    if (spec->estimator == MCP_EST_OLS) {
      rc = ols_arm(jb, ws, s);
      if (rc) { jb->rc = rc; break; }
    } else if (spec->estimator == MCP_EST_GLM && spec->cluster.present) { /* GLMM branch, batch.rs:1575-1830 */
      rc = mcpo_glmm_arm(spec, ws, jb->max_n, jb->ns, n_ns, jb->crit, cs, R->uncorrected + cell0 * nt,
                         R->corrected + cell0 * nt, R->converged + cell0, R->boundary_hit + cell0,
                         R->joint_unc + cell0, R->joint_cor + cell0, R->overall ? R->overall + cell0 : NULL,
                         R->tau_squared_hat + cell0, R->stat_capture ? R->stat_capture + (size_t)s * nt : NULL,
                         R->pinned_components + cell0);
      if (rc) { jb->rc = rc; break; }
    } else if (spec->estimator == MCP_EST_GLM) {
      rc = mcpo_glm_arm(spec, ws, jb->max_n, jb->ns, n_ns, jb->crit, cs, &jb->cache, R->uncorrected + cell0 * nt,
                        R->corrected + cell0 * nt, R->converged + cell0, R->overall ? R->overall + cell0 : NULL, fx,
                        R->stat_capture ? R->stat_capture + (size_t)s * nt : NULL);
      if (rc) { jb->rc = rc; break; }
    } else {
      rc = mcpo_mle_arm(spec, ws, jb->max_n, jb->ns, n_ns, jb->crit, cs, R->uncorrected + cell0 * nt,
                        R->corrected + cell0 * nt, R->converged + cell0, R->boundary_hit + cell0,
                        R->joint_unc + cell0, R->joint_cor + cell0, fx,
                        R->stat_capture ? R->stat_capture + (size_t)s * nt : NULL,
                        R->pinned_components + cell0); /* bit k = diagonal variance component k pinned at 0 (spec.rs:566-573) */
      if (rc) { jb->rc = rc; break; }
    }
  }
  mcpo_ws_free(ws);
  mcpo_red_cache_free(&jb->cache);
  return NULL;
}

void mcpo_free_result(mcp_batch_result* r) {
  if (!r) return;
  free(r->uncorrected); free(r->corrected); free(r->posthoc_unc); free(r->posthoc_cor);
  free(r->converged); free(r->boundary_hit); free(r->pinned_components); free(r->joint_unc);
  free(r->joint_cor); free(r->overall); free(r->factor_excluded); free(r->tau_squared_hat);
  free(r->stat_capture); free(r->counters);
  memset(r, 0, sizeof *r);
}

/* aggregation.rs:61-330 restricted to the additive counters */
static void fold_counters(const mcp_spec* spec, mcp_batch_result* R) {
  const mcp_counter_layout* L = &R->layout;
  uint32_t nt = R->n_targets, n_ns = R->n_sample_sizes, npc = R->n_posthoc_contrasts;
  for (uint32_t s = 0; s < R->n_sims; ++s)
    for (uint32_t g = 0; g < n_ns; ++g) {
      uint64_t* c = R->counters + (size_t)g * L->stride;
      size_t cell = (size_t)s * n_ns + g;
      uint32_t ku = 0, kc = 0;
      for (uint32_t t = 0; t < nt; ++t) {
        uint8_t u = R->uncorrected[cell * nt + t], k = R->corrected[cell * nt + t];
        c[L->k_unc + t] += u;
        c[L->k_cor + t] += k;
        ku += u != 0;
        kc += k != 0;
      }
      for (uint32_t q = 0; q < npc; ++q) {
        uint8_t u = R->posthoc_unc[cell * npc + q], k = R->posthoc_cor[cell * npc + q];
        c[L->posthoc_unc + q] += u;
        c[L->posthoc_cor + q] += k;
        ku += u != 0;
        kc += k != 0;
      }
      c[L->hist_unc + ku] += 1;
      c[L->hist_cor + kc] += 1;
      c[L->converged] += R->converged[cell];
      if (R->overall) c[L->overall] += R->overall[cell];
      c[L->joint_unc] += R->joint_unc[cell];
      c[L->joint_cor] += R->joint_cor[cell];
      c[L->boundary_1] += R->boundary_hit[cell] == 1;
      c[L->boundary_2] += R->boundary_hit[cell] == 2;
      for (uint32_t f = 0; f < R->n_factors; ++f) {
        c[L->excl_sparse + f] += R->factor_excluded[cell * R->n_factors + f] == 1;
        c[L->excl_sep + f] += R->factor_excluded[cell * R->n_factors + f] == 2;
      }
      c[L->n_sims] += 1;
    }
  (void)spec;
}

int mcpo_run_batch(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_sims, uint64_t base_seed,
                   const mcp_run_options* opts, uint32_t n_threads, mcp_batch_result* out) {
  memset(out, 0, sizeof *out);
  int rc = validate(spec, ns, n_ns);
  if (rc) return rc;
  uint32_t p = 1 + spec->n_non_factor + spec->n_factor_dummies + spec->n_interactions;
  uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  uint32_t npc = total_posthoc_contrasts(spec);
  uint32_t cs = nt + 4;
  if (npc > 256 || nt > MCPO_MAX_P * 2) { mcpo_set_error("unsupported: too many targets"); return MCP_ERR_UNSUPPORTED; }
  double* crit = (double*)malloc(sizeof(double) * n_ns * cs);
  rc = mcpo_build_crit_table(spec, ns, n_ns, crit);
  if (rc) { mcpo_set_error(rc == MCP_ERR_INVALID_SPEC ? "invalid spec: df = N - P < 1 or alpha outside (0,1)" : "unsupported correction"); free(crit); return rc; }

  /* posthoc contrasts + per-block correction rows (batch.rs:2189-2223, :2383-2405) */
  double* post_contrasts = npc ? (double*)calloc((size_t)npc * p, sizeof(double)) : NULL;
  double* post_corr = npc ? (double*)calloc((size_t)npc * n_ns, sizeof(double)) : NULL;
  if (npc) {
    uint32_t ci = 0;
    for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
      const uint32_t* d = spec->posthoc_target_indices + spec->posthoc_target_offsets[b];
      uint32_t nd = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b], k = nd + 1;
      for (uint32_t i = 0; i < nd; ++i)
        if (d[i] >= p) { mcpo_set_error("invalid spec: posthoc target index out of range"); free(crit); free(post_contrasts); free(post_corr); return MCP_ERR_INVALID_SPEC; }
      uint32_t blk = k * (k - 1) / 2, c0 = ci;
      for (uint32_t a = 0; a < k; ++a)
        for (uint32_t bb = a + 1; bb < k; ++bb) {
          double* c = post_contrasts + (size_t)ci * p;
          if (bb > 0) c[d[bb - 1]] = 1.0;
          if (a > 0) c[d[a - 1]] -= 1.0;
          ++ci;
        }
      (void)blk; (void)c0;
    }
  }

  if (npc) {
    rc = mcpo_build_posthoc_corr(spec, ns, n_ns, p, post_corr);
    if (rc) { free(crit); free(post_contrasts); free(post_corr); mcpo_set_error("invalid spec: posthoc thresholds"); return rc; }
  }

  size_t cells = (size_t)n_sims * n_ns;
  out->n_sims = n_sims; out->n_sample_sizes = n_ns; out->n_targets = nt; out->n_factors = spec->n_factors;
  out->n_posthoc_contrasts = npc;
  out->n_variance_components = (spec->estimator == MCP_EST_MLE || (spec->estimator == MCP_EST_GLM && spec->cluster.present)) ? 1 : 0;
  out->uncorrected = (uint8_t*)calloc(cells * nt + 1, 1);
  out->corrected = (uint8_t*)calloc(cells * nt + 1, 1);
  out->posthoc_unc = npc ? (uint8_t*)calloc(cells * npc + 1, 1) : NULL;
  out->posthoc_cor = npc ? (uint8_t*)calloc(cells * npc + 1, 1) : NULL;
  out->converged = (uint8_t*)calloc(cells + 1, 1);
  out->boundary_hit = (uint8_t*)calloc(cells + 1, 1);
  out->pinned_components = (uint64_t*)calloc(cells + 1, sizeof(uint64_t));
  out->joint_unc = (uint8_t*)calloc(cells + 1, 1);
  out->joint_cor = (uint8_t*)calloc(cells + 1, 1);
  out->overall = spec->report_overall ? (uint8_t*)calloc(cells + 1, 1) : NULL;
  out->factor_excluded = spec->n_factors ? (uint8_t*)calloc(cells * spec->n_factors + 1, 1) : NULL;
  out->tau_squared_hat = (double*)malloc(sizeof(double) * (cells + 1));
  for (size_t i = 0; i < cells; ++i) out->tau_squared_hat[i] = NAN;
  uint32_t flags = opts ? opts->flags : MCP_WANT_PER_SIM;
  if (flags & MCP_WANT_STAT_CAPTURE) {
    out->stat_capture = (double*)malloc(sizeof(double) * ((size_t)n_sims * nt + 1));
    for (size_t i = 0; i < (size_t)n_sims * nt; ++i) out->stat_capture[i] = NAN;
  }
  mcpo_counter_layout(spec, &out->layout);
  out->counters = (uint64_t*)calloc((size_t)n_ns * out->layout.stride, sizeof(uint64_t));

  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  if (n_threads > n_sims) n_threads = n_sims ? n_sims : 1;
  job jobs[256];
  pthread_t th[256];
  for (uint32_t t = 0; t < n_threads; ++t) {
    job* jb = &jobs[t];
    jb->spec = spec; jb->ns = ns; jb->n_ns = n_ns; jb->max_n = ns[n_ns - 1]; jb->p = p; jb->nt = nt; jb->npc = npc;
    jb->base_seed = base_seed; jb->sim_offset = opts ? opts->sim_offset : 0;
    jb->sim_begin = (uint32_t)((uint64_t)n_sims * t / n_threads);
    jb->sim_end = (uint32_t)((uint64_t)n_sims * (t + 1) / n_threads);
    jb->crit = crit; jb->post_corr = post_corr; jb->post_contrasts = post_contrasts; jb->out = out; jb->flags = flags; jb->rc = 0;
    memset(&jb->cache, 0, sizeof jb->cache);
  }
  if (n_threads == 1) worker(&jobs[0]);
  else {
    for (uint32_t t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker, &jobs[t]);
    for (uint32_t t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  for (uint32_t t = 0; t < n_threads; ++t)
    if (jobs[t].rc) rc = jobs[t].rc;
  free(crit); free(post_contrasts); free(post_corr);
  if (rc) { mcpo_set_error("oracle run failed with status %d", rc); mcpo_free_result(out); return rc; }
  fold_counters(spec, out);
  return MCP_OK;
}
