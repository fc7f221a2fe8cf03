// host_lowering.cpp -- run_batch's host prologue: validation and lowering of a
// SimulationSpec to the launch-constant block the kernels read.
//
// Restates /root/reference/crates/engine-core/src/batch.rs:208-330 (validation,
// same order and wording of the InvalidSpec messages), data_gen.rs:943-961
// (fixed_level_next, the deterministic largest-remainder allocation walk --
// consumes no RNG, so it is evaluated once per call on the host and uploaded as
// a level table), data_gen.rs:998-1005 + correlation.rs:48 (Cholesky factor of
// the correlation, identical for every replicate on the optimistic path), and
// batch.rs:2334-2407 (critical-value tables).
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <string>

#include "host_internal.hpp"
#include "mcp_params.hpp"

namespace mcp {

// k-point Gauss-Hermite rule for the weight exp(-z^2 / 2) / sqrt(2 pi): Newton on the orthonormal Hermite three-term
// recurrence from the classical asymptotic starting guesses, scaled from the e^{-x^2} rule (z = sqrt(2) x,
// w / sqrt(pi)); nodes descending, the centre node of an odd rule exactly 0 (lme4 GHrule convention, sum w = 1)
static void gauss_hermite_normal(uint32_t k, double* z, double* w) {
  const double pim4 = 0.7511255444649425;
  const uint32_t half = (k + 1) / 2;
  double x = 0.0, dp = 1.0;
  for (uint32_t i = 0; i < half; ++i) {
    if (i == 0) x = std::sqrt(2.0 * k + 1.0) - 1.85575 * std::pow(2.0 * k + 1.0, -0.16667);
    else if (i == 1) x -= 1.14 * std::pow((double)k, 0.426) / x;
    else if (i == 2) x = 1.86 * x - 0.86 * z[0];
    else if (i == 3) x = 1.91 * x - 0.91 * z[1];
    else x = 2.0 * x - z[i - 2];
    for (int it = 0; it < 100; ++it) {
      double h1 = pim4, h2 = 0.0;
      for (uint32_t j = 0; j < k; ++j) {
        const double h3 = h2;
        h2 = h1;
        h1 = x * std::sqrt(2.0 / (j + 1.0)) * h2 - std::sqrt((double)j / (j + 1.0)) * h3;
      }
      dp = std::sqrt(2.0 * k) * h2;
      const double prev = x;
      x = prev - h1 / dp;
      if (std::fabs(x - prev) <= 1e-15 * (1.0 + std::fabs(x))) break;
    }
    z[i] = x;
    z[k - 1 - i] = -x;
    w[i] = w[k - 1 - i] = 2.0 / (dp * dp);
  }
  for (uint32_t i = 0; i < k; ++i) {
    z[i] *= 1.4142135623730951;
    w[i] /= 1.7724538509055159;
  }
  if (k & 1u) z[k / 2] = 0.0;
}


static bool scenario_is_optimistic(const mcp_scenario& s) {  // spec.rs:229-236
  return s.heterogeneity == 0.0 && s.heteroskedasticity_ratio == 1.0 && s.correlation_noise_sd == 0.0 &&
         s.distribution_change_prob == 0.0 && s.residual_change_prob == 0.0 && !s.has_lme;
}

// lower Cholesky factor, column-major in, row-major out; false when a pivot is not positive
static bool chol_lower(const double* a_colmajor, uint32_t k, double* l_rowmajor, uint32_t ld) {
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < ld; ++j) l_rowmajor[i * ld + j] = 0.0;
  for (uint32_t j = 0; j < k; ++j) {
    double d = a_colmajor[j * k + j];
    for (uint32_t t = 0; t < j; ++t) d -= l_rowmajor[j * ld + t] * l_rowmajor[j * ld + t];
    if (!(d > 0.0)) return false;
    const double ljj = std::sqrt(d);
    l_rowmajor[j * ld + j] = ljj;
    for (uint32_t i = j + 1; i < k; ++i) {
      double s = a_colmajor[j * k + i];
      for (uint32_t t = 0; t < j; ++t) s -= l_rowmajor[i * ld + t] * l_rowmajor[j * ld + t];
      l_rowmajor[i * ld + j] = s / ljj;
    }
  }
  return true;
}

void counter_layout(const mcp_spec* spec, mcp_counter_layout* L) {
  const uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  uint32_t npc = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    const uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    npc += k * (k - 1) / 2;
  }
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

int validate_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, std::string& err) {
  auto bad = [&](const std::string& m) {
    err = "invalid spec: " + m;
    return (int)MCP_ERR_INVALID_SPEC;
  };
  if (!spec) return bad("null spec");
  if (spec->abi_version != MCP_ABI_VERSION) return bad("abi_version mismatch");
  if (spec->estimator == MCP_EST_MLE && !spec->cluster.present) return bad("estimator Mle requires ClusterSpec");
  if (spec->n_posthoc && spec->estimator != MCP_EST_OLS) return bad("posthoc tests are only supported for estimator=Ols");
  if (n_ns == 0 || !ns) return bad("sample_sizes must be non-empty");
  for (uint32_t i = 1; i < n_ns; ++i)
    if (ns[i] <= ns[i - 1])
      return bad("sample_sizes must be strictly increasing; got " + std::to_string(ns[i - 1]) + " then " + std::to_string(ns[i]));
  if (ns[0] == 0) return bad("sample_sizes[0] must be > 0");
  const uint32_t k = spec->n_non_factor;
  const uint32_t p = 1 + k + spec->n_factor_dummies + spec->n_interactions;
  if (spec->n_effect_sizes != p)
    return bad("effect_sizes length " + std::to_string(spec->n_effect_sizes) + " does not match n_predictors " +
               std::to_string(p) + " (1 intercept + " + std::to_string(k) + " non_factor + " +
               std::to_string(spec->n_factor_dummies) + " factor dummies)");
  uint32_t expected_props = 0, expected_dummies = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    if (spec->factor_n_levels[f] < 2)
      return bad("factor_n_levels entries must be >= 2; got " + std::to_string(spec->factor_n_levels[f]));
    expected_props += (uint32_t)spec->factor_n_levels[f];
    expected_dummies += (uint32_t)spec->factor_n_levels[f] - 1;
  }
  if (spec->n_factor_proportions != expected_props)
    return bad("factor_proportions length " + std::to_string(spec->n_factor_proportions) +
               " does not match sum(factor_n_levels) " + std::to_string(expected_props));
  if (expected_dummies != spec->n_factor_dummies) return bad("n_factor_dummies does not match factor_n_levels");
  for (uint32_t t = 0; t < spec->n_target_indices; ++t)
    if (spec->target_indices[t] >= p)
      return bad("target_indices entry " + std::to_string(spec->target_indices[t]) + " out of range (n_predictors = " +
                 std::to_string(p) + ")");
  for (uint32_t c = 0; c < 2 * spec->n_contrast_pairs; ++c)
    if (spec->contrast_pairs[c] >= p)
      return bad("contrast_pairs entry " + std::to_string(spec->contrast_pairs[c]) + " out of range (n_predictors = " +
                 std::to_string(p) + ")");
  for (uint32_t j = 0; j < spec->n_interactions; ++j)
    for (uint32_t c = spec->interaction_offsets[j]; c < spec->interaction_offsets[j + 1]; ++c)
      if (spec->interaction_components[c] == 0 || spec->interaction_components[c] >= 1 + k + spec->n_factor_dummies)
        return bad("interaction component out of range");
  if (k > 0) {  // trial Cholesky (batch.rs:320-330)
    std::vector<double> l((size_t)k * k);
    if (!chol_lower(spec->correlation, k, l.data(), k)) {
      err = "correlation matrix is not positive semi-definite";
      return MCP_ERR_CORRELATION_NOT_PSD;
    }
  }
  return MCP_OK;
}

// data_gen.rs:943-961
static uint32_t fixed_level_next(const double* probs, uint32_t* counts, uint32_t nl) {
  double total = 0.0;
  uint32_t assigned = 0;
  for (uint32_t g = 0; g < nl; ++g) {
    total += probs[g];
    assigned += counts[g];
  }
  const double t = double(assigned + 1);
  uint32_t best = 0;
  double best_def = -INFINITY;
  for (uint32_t g = 0; g < nl; ++g) {
    const double deficit = (probs[g] / total) * t - double(counts[g]);
    if (deficit > best_def) {
      best_def = deficit;
      best = g;
    }
  }
  counts[best] += 1;
  return best;
}


uint32_t total_posthoc_contrasts(const mcp_spec* spec) {
  uint32_t npc = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    const uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    npc += k * (k - 1) / 2;
  }
  return npc;
}

// batch.rs:2235-2253 tukey_k_for_kernel_col: level count of the factor owning a dummy column, NaN elsewhere
static double tukey_k_for_col(const mcp_spec* spec, uint32_t col) {
  const uint32_t base = 1 + spec->n_non_factor;
  if (col < base) return std::nan("");
  uint32_t cursor = base;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    const uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0u;
    const uint32_t width = l ? l - 1 : 0;
    if (col < cursor + width) return (double)l;
    cursor += width;
  }
  return std::nan("");
}

// batch.rs:2346-2374: marginal targets first, then contrasts (the positive side names the factor)
void tukey_k_per_target(const mcp_spec* spec, std::vector<double>& k) {
  k.clear();
  for (uint32_t t = 0; t < spec->n_target_indices; ++t) k.push_back(tukey_k_for_col(spec, spec->target_indices[t]));
  for (uint32_t q = 0; q < spec->n_contrast_pairs; ++q) k.push_back(tukey_k_for_col(spec, spec->contrast_pairs[2 * q]));
}

// build_crit_tables (batch.rs:2334-2407) at `n_fit` fitted columns: the main table, then per post-hoc block a
// table with n_targets = C(k,2) at alpha = posthoc_alpha whose correction row is appended (:2262-2313, :2383-2405)
int build_crit_rows_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, uint32_t n_fit, TukeyMemo* memo,
                         std::vector<double>& out, std::string& err) {
  const uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  const uint32_t npc = total_posthoc_contrasts(spec), stride = nt + 4 + npc;
  std::vector<double> kk, main_rows;
  if (spec->correction_method == MCP_CORR_TUKEY_HSD) tukey_k_per_target(spec, kk);
  int rc = build_crit_rows(spec->estimator, spec->correction_method, spec->alpha, spec->posthoc_alpha, n_fit, nt, ns, n_ns,
                           kk.empty() ? nullptr : kk.data(), memo, main_rows, err);
  if (rc) return rc;
  out.assign((size_t)n_ns * stride, 0.0);
  for (uint32_t g = 0; g < n_ns; ++g)
    std::memcpy(out.data() + (size_t)g * stride, main_rows.data() + (size_t)g * (nt + 4), sizeof(double) * (nt + 4));
  const double pa = std::isnan(spec->posthoc_alpha) ? spec->alpha : spec->posthoc_alpha;
  uint32_t c0 = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    const uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    const uint32_t blk = k * (k - 1) / 2;
    std::vector<double> bk(blk, (double)k), rows;
    rc = build_crit_rows(spec->estimator, spec->correction_method, pa, pa, n_fit, blk, ns, n_ns, bk.data(), memo, rows, err);
    if (rc) return rc;
    for (uint32_t g = 0; g < n_ns; ++g)
      for (uint32_t q = 0; q < blk; ++q) out[(size_t)g * stride + nt + 4 + c0 + q] = rows[(size_t)g * (blk + 4) + 1 + q];
    c0 += blk;
  }
  return MCP_OK;
}

int lower_spec(const mcp_spec* spec, const uint32_t* ns, uint32_t n_ns, SimParams& prm, Lowered& low, std::string& err) {
  auto unsupported = [&](const std::string& m) {
    err = "unsupported on the device path: " + m;
    return (int)MCP_ERR_UNSUPPORTED;
  };
  std::memset(&prm, 0, sizeof prm);
  const uint32_t K = spec->n_non_factor;
  const uint32_t P = 1 + K + spec->n_factor_dummies + spec->n_interactions;
  const uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs;
  const uint32_t max_n = ns[n_ns - 1];
  if (P < 2 || P > (uint32_t)MAX_P) return unsupported("design columns P = " + std::to_string(P) + " outside [2, " + std::to_string(MAX_P) + "]");
  if (K > (uint32_t)MAX_K) return unsupported("more than " + std::to_string(MAX_K) + " continuous predictors");
  if (nt > (uint32_t)MAX_T) return unsupported("more than " + std::to_string(MAX_T) + " tests per fit");
  if (n_ns > (uint32_t)MAX_GRID) return unsupported("more than " + std::to_string(MAX_GRID) + " sample sizes per call");
  if (spec->n_factors > (uint32_t)MAX_FACTORS) return unsupported("more than " + std::to_string(MAX_FACTORS) + " factors");
  if (spec->n_factor_proportions > (uint32_t)MAX_LEVELS) return unsupported("more than " + std::to_string(MAX_LEVELS) + " factor levels in total");
  bool lmm_general = false;  // random slopes / extra groupings: the general mixed-model kernel
  if (spec->cluster.present) {
    lmm_general = spec->cluster.slopes.n_slopes != 0 || spec->cluster.n_extra_groupings != 0;
    if (spec->cluster.slopes.n_slopes > MCP_MAX_RE_SLOPES || spec->cluster.n_extra_groupings > MCP_MAX_EXTRA_GROUPINGS)
      return unsupported("more than 2 random slopes per grouping or more than 2 extra groupings");
  }
  if (spec->estimator == MCP_EST_MLE) {
    if (spec->outcome_kind != MCP_OUTCOME_CONTINUOUS) return unsupported("Mle on Binary / Count outcomes");
    if (spec->n_contrast_pairs) return unsupported("contrast pairs under the Mle estimator");

    if (spec->cluster.sizing_value == 0) { err = "invalid spec: cluster sizing value must be > 0"; return MCP_ERR_INVALID_SPEC; }
    if (spec->cluster.sizing == MCP_CLUSTER_FIXED_SIZE)
      for (uint32_t g = 0; g < n_ns; ++g)
        if (ns[g] % spec->cluster.sizing_value) return unsupported("FixedSize clusters with a sample size that is not a multiple of cluster_size");
  }
  if (spec->estimator == MCP_EST_OLS && spec->outcome_kind != MCP_OUTCOME_CONTINUOUS)
    return unsupported("OLS on Binary / Count outcomes");
  // the IRLS kernel is instantiated up to 16 design columns / tests for unclustered GLMs on the plain data path
  // (glm_wide_kernels.hip: the reference bench's glm_wide, throughput.rs:300-303); everything else stays at 8
  const bool glm_plain = spec->estimator == MCP_EST_GLM && !spec->cluster.present;
  if (glm_plain && (P > (uint32_t)MAX_P || nt > (uint32_t)MAX_T))
    return unsupported("Glm with more than " + std::to_string(MAX_P) + " design columns or tests");
  // Mle with at most one random slope on the primary grouping: 9 .. 12 design columns run on the run-time-sized general
  // mixed-model kernel (lmm_general_impl.hpp, instantiated for them at block widths 1 and 2); the rest of this family stays at 8
  bool mle_wide = false;
  if (spec->estimator == MCP_EST_MLE && spec->cluster.present && P > (uint32_t)MAX_P_WAVE && P <= 12u)
    mle_wide = spec->cluster.slopes.n_slopes <= 1;   // the extras may carry slopes of their own: only the eliminated block width is compiled in
  if (mle_wide) lmm_general = true;
  if (spec->estimator != MCP_EST_OLS && !glm_plain && ((P > (uint32_t)MAX_P_WAVE && !mle_wide) || nt > (mle_wide ? 12u : (uint32_t)MAX_T_WAVE)))
    return unsupported("Mle / clustered Glm with more than " + std::to_string(MAX_P_WAVE) + " design columns or tests (Mle with at most one random slope on the primary grouping: 12 columns, 12 tests from 9 columns on)");
  if (spec->estimator == MCP_EST_GLM) {
    if (spec->outcome_kind == MCP_OUTCOME_CONTINUOUS) return unsupported("Glm on a Continuous outcome");
    if (spec->outcome_kind == MCP_OUTCOME_COUNT && spec->link != MCP_LINK_CANONICAL)
      return unsupported("Count outcomes under a non-log link");
    if (spec->n_contrast_pairs) return unsupported("contrast pairs under the GLM estimator");
  }
  const bool glmm = spec->cluster.present && spec->estimator == MCP_EST_GLM;  // batch.rs:1575-1830
  if (glmm && spec->cluster.n_extra_groupings && spec->nagq > 1)
    return unsupported("adaptive Gauss-Hermite quadrature (nagq > 1) for a clustered GLM with extra groupings (Laplace only)");
  if (lmm_general) {
    if (spec->estimator != MCP_EST_MLE && !glmm) return unsupported("random slopes / extra groupings outside the Mle / clustered Glm estimators");
    for (uint32_t t = 0; t <= spec->cluster.n_extra_groupings; ++t) {
      const mcp_re_slopes& sl = t == 0 ? spec->cluster.slopes : spec->cluster.extra[t - 1].slopes;
      if (sl.n_slopes > MCP_MAX_RE_SLOPES) return unsupported("more than 2 random slopes per grouping");
      for (uint32_t d = 0; d < sl.n_slopes; ++d)
        if (sl.design_cols[d] == 0 || sl.design_cols[d] >= P) { err = "invalid spec: random-slope design column out of range"; return MCP_ERR_INVALID_SPEC; }
      if (t > 0) {
        if (spec->cluster.extra[t - 1].count == 0) { err = "invalid spec: extra grouping with zero levels"; return MCP_ERR_INVALID_SPEC; }
        if (spec->cluster.extra[t - 1].relation == MCP_GROUPING_CROSSED && spec->cluster.sizing != MCP_CLUSTER_FIXED_CLUSTERS) {
          err = "invalid spec: a crossed grouping needs FixedClusters sizing";  // validate(): crossed x FixedSize
          return MCP_ERR_INVALID_SPEC;
        }
      }
    }
  }
  if (spec->cluster.present && spec->cluster.sizing_value == 0) { err = "invalid spec: cluster sizing value must be > 0"; return MCP_ERR_INVALID_SPEC; }
  if (spec->cluster.present && spec->cluster.sizing == MCP_CLUSTER_FIXED_SIZE)
    for (uint32_t g = 0; g < n_ns; ++g)
      if (ns[g] % spec->cluster.sizing_value) return unsupported("FixedSize clusters with a sample size that is not a multiple of cluster_size");
  // general ("scenario") data-generating path: anything beyond N(0,1) predictors + N(0,1) residual + fixed design
  bool general = !scenario_is_optimistic(spec->scenario) || spec->residual_dist != MCP_RESID_NORMAL;
  for (uint32_t j = 0; j < K; ++j)
    if (spec->var_types[j] != MCP_DIST_NORMAL) general = true;
  // uploaded frames (contract_adapter.rs:356-469): NORTA quantile table and / or the strict-bootstrap frame
  if (spec->upload_normal_rows && spec->upload_normal_cols != K) {
    err = "invalid spec: upload_normal must have one column per non-factor predictor";
    return MCP_ERR_INVALID_SPEC;
  }
  if (spec->n_bootstrap_frame_map) {
    if (spec->n_bootstrap_frame_map != K + spec->n_factors || spec->upload_data_rows == 0) {
      err = "invalid spec: bootstrap_frame_map length must equal n_non_factor + n_factors and the frame must be non-empty";
      return MCP_ERR_INVALID_SPEC;
    }
    for (uint32_t j = 0; j < spec->n_bootstrap_frame_map; ++j)
      if (spec->bootstrap_frame_map[j] >= (int32_t)spec->upload_data_cols) {
        err = "invalid spec: bootstrap_frame_map entry beyond the uploaded frame's columns";
        return MCP_ERR_INVALID_SPEC;
      }
    general = true;
  }
  if (general && K > 9) return unsupported("general data-generating path with more than 9 continuous predictors");
  if (spec->scenario.has_lme && spec->estimator != MCP_EST_MLE &&
      !(spec->estimator == MCP_EST_GLM && spec->cluster.present)) {  // validate.rs:458-474: Mle or clustered Glm
    err = "invalid spec: scenario.lme requires estimator == Mle";
    return MCP_ERR_INVALID_SPEC;
  }
  if (spec->n_between_var_indices && spec->cluster.present && spec->estimator == MCP_EST_OLS &&
      spec->cluster.sizing == MCP_CLUSTER_FIXED_CLUSTERS) {
    // the lane-per-replicate kernel parks the representatives' marked values in LDS: clusters x marked columns x 1 KiB per block
    // (next to the staging area, (P + 1) KiB x 4 rows per block, inside the 160 KB of a CU)
    const size_t park = (size_t)spec->cluster.sizing_value * spec->n_between_var_indices * 1024u;
    if (park > 96u * 1024u || park + (size_t)(P + 1u) * 4096u > 150u * 1024u)
      return unsupported("cluster-level (between) predictors under the OLS estimator with more than 96 cluster x column values");
  }
  for (uint32_t b = 0; b < spec->n_between_var_indices; ++b)
    if (spec->between_var_indices[b] == 0 || spec->between_var_indices[b] >= 1 + K + spec->n_factor_dummies) {
      err = "invalid spec: between_var_indices entry must name a continuous or dummy column";
      return MCP_ERR_INVALID_SPEC;
    }
  const uint32_t npc = total_posthoc_contrasts(spec);
  if (npc > (uint32_t)MAX_PC) return unsupported("more than " + std::to_string(MAX_PC) + " post-hoc contrasts");
  for (uint32_t i = 0; i < (spec->n_posthoc ? spec->posthoc_target_offsets[spec->n_posthoc] : 0u); ++i)
    if (spec->posthoc_target_indices[i] >= P) {
      err = "invalid spec: posthoc target index " + std::to_string(spec->posthoc_target_indices[i]) +
            " out of range (n_predictors = " + std::to_string(P) + ")";
      return MCP_ERR_INVALID_SPEC;
    }
  if (spec->factor_min_level_count > 15 && spec->n_factors)
    return unsupported("factor_min_level_count above 15 (per-level row counts are kept in saturating 4-bit lanes)");
  if (spec->heteroskedasticity_driver >= (int32_t)(1 + K + spec->n_factor_dummies)) {
    err = "invalid spec: heteroskedasticity_driver out of range";
    return MCP_ERR_INVALID_SPEC;
  }

  prm.max_n = max_n;
  prm.n_grid = n_ns;
  for (uint32_t g = 0; g < n_ns; ++g) prm.ns[g] = ns[g];
  prm.P = P;
  prm.K = K;
  bool ident = true;
  if (K) {
    if (!chol_lower(spec->correlation, K, prm.L, MAX_K)) {
      err = "correlation matrix is not positive semi-definite";
      return MCP_ERR_CORRELATION_NOT_PSD;
    }
    for (uint32_t i = 0; i < K; ++i)
      for (uint32_t j = 0; j < K; ++j)
        if (prm.L[i * MAX_K + j] != (i == j ? 1.0 : 0.0)) ident = false;
  }
  prm.ident_L = ident ? 1u : 0u;
  for (uint32_t c = 0; c < P; ++c) prm.beta[c] = spec->effect_sizes[c];

  // factors
  prm.n_factors = spec->n_factors;
  for (uint32_t c = 0; c < (uint32_t)MAX_P; ++c) {
    prm.col_factor[c] = 0xffffffffu;
    prm.col_level[c] = 0;
    prm.inter_mask[c] = 0;
  }
  uint32_t col = 1 + K, poff = 0, n_sampled = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    const uint32_t nl = (uint32_t)spec->factor_n_levels[f];
    prm.f_levels[f] = nl;
    prm.f_col[f] = col;
    prm.f_prop_off[f] = poff;
    int sampled = spec->scenario.sampled_factor_proportions;
    if (f < spec->n_factor_sampled && spec->factor_sampled[f] >= 0) sampled = spec->factor_sampled[f];
    prm.f_sampled[f] = sampled ? 1u : 0u;
    prm.f_seq_slot[f] = sampled ? n_sampled++ : 0u;
    for (uint32_t l = 0; l < nl; ++l) prm.f_props[poff + l] = spec->factor_proportions[poff + l];
    for (uint32_t d = 0; d + 1 < nl; ++d) {
      prm.col_factor[col + d] = f;
      prm.col_level[col + d] = d + 1;
    }
    col += nl - 1;
    poff += nl;
  }
  prm.n_seq_pre = 0;
  prm.n_sampled_factors = n_sampled;
  prm.bootstrap = spec->n_bootstrap_frame_map ? 1u : 0u;
  prm.n_seq_per_row = n_sampled + prm.bootstrap;  // categorical draws, then the bootstrap row pick (data_gen.rs:509-566)
  for (uint32_t j = 0; j < (uint32_t)(MAX_K + MAX_FACTORS); ++j) prm.boot_map[j] = -1;
  for (uint32_t j = 0; j < spec->n_bootstrap_frame_map; ++j) prm.boot_map[j < K ? j : K + (j - K)] = spec->bootstrap_frame_map[j];
  prm.upload_rows = spec->upload_normal_rows;
  prm.upload_data_rows = spec->upload_data_rows;
  prm.upload_data_cols = spec->upload_data_cols;
  if (spec->cluster.present) {  // data_gen.rs:146-161: one RE draw per cluster present at max_n, before the row loop
    prm.has_cluster = 1;
    prm.cluster_fixed_clusters = spec->cluster.sizing == MCP_CLUSTER_FIXED_CLUSTERS ? 1u : 0u;
    prm.cluster_v = spec->cluster.sizing_value;
    prm.n_clusters_max = prm.cluster_fixed_clusters ? prm.cluster_v : max_n / prm.cluster_v;
    prm.tau = std::sqrt(spec->cluster.tau_squared > 0.0 ? spec->cluster.tau_squared : 0.0);
    prm.tau_squared = spec->cluster.tau_squared;
    for (uint32_t b = 0; b < spec->n_between_var_indices; ++b) prm.between_mask |= 1u << spec->between_var_indices[b];
    // draws one cluster effect consumes on the sequential stream (draw_residual, data_gen.rs:1289-1320)
    prm.lme_re_dist = spec->scenario.has_lme ? (uint32_t)spec->scenario.lme_random_effect_dist : 0u;
    prm.lme_re_df = spec->scenario.has_lme ? spec->scenario.lme_random_effect_df : 0.0;
    prm.lme_icc_noise_sd = spec->scenario.has_lme ? spec->scenario.lme_icc_noise_sd : 0.0;
    {
      const double df = prm.lme_re_df > 3.0 ? prm.lme_re_df : 3.0;
      long dfi = std::lround(df);
      if (dfi < 1) dfi = 1;
      if (prm.lme_re_dist != 0u && prm.lme_re_dist != 4u && dfi > 64) return unsupported("random-effect df above 64");
      prm.re_draws_per_cluster = prm.lme_re_dist == 1u ? (uint32_t)dfi + 1u
                                 : ((prm.lme_re_dist == 2u || prm.lme_re_dist == 3u) ? (uint32_t)dfi : 1u);
    }
    prm.n_seq_pre = prm.n_clusters_max * prm.re_draws_per_cluster;
    if (lmm_general) {
      const mcp_cluster& cl = spec->cluster;
      prm.lmg_active = 1;
      prm.lmg_stream = (spec->estimator == MCP_EST_MLE && n_ns == 1 && prm.between_mask == 0u && spec->n_factors == 0 && max_n > 256u) ? 1u : 0u;
      prm.lmg_n_terms = 1 + cl.n_extra_groupings;
      uint32_t toff = 0, roff = 0, seq = 0;
      for (uint32_t t = 0; t < prm.lmg_n_terms; ++t) {
        const mcp_re_slopes& sl = t == 0 ? cl.slopes : cl.extra[t - 1].slopes;
        const uint32_t q = 1 + sl.n_slopes;
        prm.lmg_q[t] = q;
        prm.lmg_theta_off[t] = toff;
        toff += q * (q + 1) / 2;
        if (t == 0) {
          prm.lmg_levels[0] = prm.n_clusters_max;
        } else {
          const mcp_grouping& e = cl.extra[t - 1];
          prm.lmg_relation[t] = e.relation == MCP_GROUPING_CROSSED ? 0u : 1u;
          prm.lmg_count[t] = e.count;
          const uint32_t parents = prm.cluster_fixed_clusters ? prm.cluster_v : (max_n + prm.cluster_v - 1) / prm.cluster_v;
          prm.lmg_levels[t] = e.relation == MCP_GROUPING_CROSSED ? e.count : parents * e.count;  // extra_n_levels_at
          prm.lmg_roff[t] = roff;
          roff += prm.lmg_levels[t] * q;
        }
        for (uint32_t d = 0; d < sl.n_slopes; ++d) prm.lmg_cols[t][d] = sl.design_cols[d];
        const double tau_sq = t == 0 ? cl.tau_squared : cl.extra[t - 1].tau_squared;
        prm.lmg_tau[t] = std::sqrt(tau_sq > 0.0 ? tau_sq : 0.0);
        prm.lmg_tau_sq[t] = tau_sq;
        for (uint32_t d = 0; d < sl.n_slopes; ++d) prm.lmg_slope_sd[t][d] = std::sqrt(sl.variance[d] > 0.0 ? sl.variance[d] : 0.0);
        for (uint32_t i = 0; i < q * q; ++i) prm.lmg_recorr[t][i] = sl.re_corr[i];
        // chol_lower of D = diag(tau) R diag(tau) (data_gen.rs:55-75): the slope draws and the truth start share it
        double tv[3] = {prm.lmg_tau[t], 0.0, 0.0}, dm[9] = {0}, l[9] = {0};
        for (uint32_t d = 0; d < sl.n_slopes; ++d) tv[d + 1] = std::sqrt(sl.variance[d] > 0.0 ? sl.variance[d] : 0.0);
        for (uint32_t i = 0; i < q; ++i)
          for (uint32_t j = 0; j < q; ++j) dm[i * q + j] = tv[i] * sl.re_corr[i * q + j] * tv[j];
        for (uint32_t j = 0; j < q; ++j) {
          double diag = dm[j * q + j];
          for (uint32_t k = 0; k < j; ++k) diag -= l[j * q + k] * l[j * q + k];
          const double ljj = std::sqrt(diag > 0.0 ? diag : 0.0);
          l[j * q + j] = ljj;
          for (uint32_t i = j + 1; i < q; ++i)
            if (ljj > 0.0) {
              double sacc = dm[i * q + j];
              for (uint32_t k = 0; k < j; ++k) sacc -= l[i * q + k] * l[j * q + k];
              l[i * q + j] = sacc / ljj;
            }
        }
        for (uint32_t i = 0; i < 9; ++i) prm.lmg_chol[t][i] = i < q * q ? l[i] : 0.0;
        // truth_theta (mixed_workspace.rs:112-198): column-major lower vech of the same factor
        uint32_t k = prm.lmg_theta_off[t];
        if (sl.n_slopes == 0) prm.lmg_theta_truth[k] = prm.lmg_tau[t];
        else
          for (uint32_t col = 0; col < q; ++col)
            for (uint32_t row = col; row < q; ++row) prm.lmg_theta_truth[k++] = l[row * q + col];
      }
      prm.lmg_n_theta = toff;
      prm.lmg_R = roff;
      prm.lmg_W = roff + P + 1;
      {
        // fit view (mcp_params.hpp): any grouping's levels are mutually exclusive, so any term can be the one whose levels are
        // eliminated block by block (LMM) / lead the index layout of the joint mode system (clustered GLM); the one with the
        // most random-effect columns leaves the smallest dense block.  The LMM kernel is instantiated on one block width, so
        // there only terms as wide as the primary one qualify.  (the test oracle takes the same choice)
        uint32_t E = 0, best = prm.lmg_levels[0] * prm.lmg_q[0];
        for (uint32_t t = 1; t < prm.lmg_n_terms; ++t) {
          if (spec->estimator == MCP_EST_MLE && prm.lmg_q[t] != prm.lmg_q[0]) continue;
          const uint32_t cols = prm.lmg_levels[t] * prm.lmg_q[t];
          if (cols > best) { best = cols; E = t; }
        }
        prm.lmg_elim = E;
        const bool e_nested = E != 0u && prm.lmg_relation[E] == 1u;
        prm.lmg_cpe = e_nested ? prm.lmg_count[E] : 1u;
        uint32_t fro = 0, nr = 0;
        for (uint32_t t = 0; t < prm.lmg_n_terms; ++t) {
          if (t == E) continue;
          prm.lmg_frest[nr++] = t;
          prm.lmg_froff[t] = fro;
          fro += prm.lmg_levels[t] * prm.lmg_q[t];
          // how an eliminated level meets term t: E primary -> the spec's relation; E nested in the primary -> the primary and the
          // other nested groupings through the shared primary level, crossed ones fully; E crossed -> every term fully
          prm.lmg_frel[t] = E == 0u ? prm.lmg_relation[t] : (e_nested ? (t == 0u ? 1u : prm.lmg_relation[t]) : 0u);
          prm.lmg_fcnt[t] = t == 0u ? 1u : prm.lmg_count[t];
        }
        prm.lmg_n_rest = nr;
        prm.lmg_fR = fro;
        prm.lmg_fW = fro + P + 1;
        prm.lmg_compact = 0;   // decided below, once it is known whether reduced rounds can occur
        prm.lmg_Wc = prm.lmg_fW;
        uint32_t co = 0;
        for (uint32_t r = 0; r < nr; ++r) {
          const uint32_t t = prm.lmg_frest[r];
          prm.lmg_coff[t] = co;
          co += (prm.lmg_frel[t] == 1u ? prm.lmg_fcnt[t] : prm.lmg_levels[t]) * prm.lmg_q[t];
        }
        prm.lmg_coff_xy = co;
      }
      // sequential-stream layout (data_gen.rs:146-284): primary intercepts, primary slopes, every extra grouping's
      // intercepts, then every extra grouping's slopes -- in units of one random-effect draw
      // (re_draws_per_cluster words each: 1 for Normal, df or 1 + df for the scenario's other shapes)
      prm.lmg_seq[0] = 0;
      seq = prm.lmg_levels[0];
      prm.lmg_seq[1] = seq;
      seq += prm.lmg_levels[0] * (prm.lmg_q[0] - 1);
      for (uint32_t t = 1; t < prm.lmg_n_terms; ++t) { prm.lmg_seq[2 * t] = seq; seq += prm.lmg_levels[t]; }
      for (uint32_t t = 1; t < prm.lmg_n_terms; ++t) { prm.lmg_seq[2 * t + 1] = seq; seq += prm.lmg_levels[t] * (prm.lmg_q[t] - 1); }
      prm.n_seq_pre = seq * prm.re_draws_per_cluster;
      if (prm.lmg_n_theta > (uint32_t)LMG_MAX_THETA) return unsupported("random-effect structures with more than 12 covariance parameters");
      if ((spec->estimator == MCP_EST_MLE ? prm.lmg_fW : prm.lmg_W) > 128u) return unsupported("extra groupings with more than ~120 random-effect columns in total");
    }
  }
  prm.truth_start = spec->scenario.truth_start ? 1u : 0u;
  low.fixed_levels.assign((size_t)max_n * (spec->n_factors ? spec->n_factors : 1), 0);
  {
    std::vector<uint32_t> counts(spec->n_factor_proportions ? spec->n_factor_proportions : 1, 0u);
    for (uint32_t i = 0; i < max_n; ++i)
      for (uint32_t f = 0; f < spec->n_factors; ++f)
        if (!prm.f_sampled[f])
          low.fixed_levels[(size_t)i * spec->n_factors + f] =
              (uint8_t)fixed_level_next(spec->factor_proportions + prm.f_prop_off[f], counts.data() + prm.f_prop_off[f],
                                        prm.f_levels[f]);
  }
  // interactions: ascending component lists become bitmasks (two-component products are exact in f64 and
  // commutative; longer ones must already be listed ascending, the order the reference multiplies in)
  for (uint32_t j = 0; j < spec->n_interactions; ++j) {
    uint32_t mask = 0, prev = 0;
    const uint32_t b = spec->interaction_offsets[j], e = spec->interaction_offsets[j + 1];
    for (uint32_t c = b; c < e; ++c) {
      const uint32_t comp = spec->interaction_components[c];
      if (e - b > 2 && comp <= prev) return unsupported("interaction with >2 components not listed in ascending column order");
      if (mask & (1u << comp)) return unsupported("interaction repeating a component (powers)");
      prev = comp;
      mask |= 1u << comp;
    }
    if (mask == 0) return unsupported("empty interaction");
    prm.inter_mask[1 + K + spec->n_factor_dummies + j] = mask;
  }
  // tests: marginal targets then contrast pairs (batch.rs:1045-1088)
  prm.n_tests = nt;
  for (uint32_t t = 0; t < spec->n_target_indices; ++t) prm.contrast[t * MAX_P + spec->target_indices[t]] = 1.0;
  for (uint32_t q = 0; q < spec->n_contrast_pairs; ++q) {
    double* c = prm.contrast + (spec->n_target_indices + q) * MAX_P;
    c[spec->contrast_pairs[2 * q]] += 1.0;
    c[spec->contrast_pairs[2 * q + 1]] -= 1.0;
  }
  prm.correction = spec->correction_method;
  prm.report_overall = spec->report_overall ? 1u : 0u;
  prm.outcome_kind = (uint32_t)spec->outcome_kind;
  prm.estimator = (uint32_t)spec->estimator;
  prm.glm_family = spec->outcome_kind == MCP_OUTCOME_COUNT ? 2u : (spec->link == MCP_LINK_PROBIT ? 1u : 0u);

  // ---- general path constants (data_gen.rs:709-776, 990-1093) ----
  prm.scen_active = general ? 1u : 0u;
  if (general && spec->estimator == MCP_EST_GLM && (P > (uint32_t)MAX_P_WAVE || nt > (uint32_t)MAX_T_WAVE))
    return unsupported("Glm on the general (scenario) data path with more than " + std::to_string(MAX_P_WAVE) + " design columns or tests");
  prm.scen_optimistic = scenario_is_optimistic(spec->scenario) ? 1u : 0u;
  if (general) {
    const mcp_scenario& sc = spec->scenario;
    for (uint32_t i = 0; i < K * K; ++i) prm.corr[i] = spec->correlation[i];
    for (uint32_t j = 0; j < K; ++j) {
      prm.var_types[j] = (uint32_t)spec->var_types[j];
      prm.var_pinned[j] = (j < spec->n_var_pinned && spec->var_pinned[j]) ? 1u : 0u;
      prm.var_params[j] = j < spec->n_var_params ? spec->var_params[j] : 0.5;
    }
    prm.resid_spec_dist = (uint32_t)spec->residual_dist;
    prm.residual_pinned = spec->residual_pinned ? 1u : 0u;
    prm.corr_noise_sd = sc.correlation_noise_sd;
    prm.dist_change_prob = sc.distribution_change_prob;
    prm.resid_change_prob = sc.residual_change_prob;
    if (sc.n_new_distributions > 8 || sc.n_residual_dists > 8) return unsupported("swap menus longer than 8 entries");
    prm.n_new_dist = sc.n_new_distributions;
    prm.n_resid_dists = sc.n_residual_dists;
    bool need_t3 = false, may_chi = false, may_unif = false;
    for (uint32_t i = 0; i < sc.n_new_distributions; ++i) {
      prm.new_dist[i] = sc.new_distributions[i];
      if (sc.new_distributions[i] >= MCP_DIST_UPLOADED_DATA) {
        err = "invalid spec: uploaded-data marginals cannot appear in the scenario swap menu";
        return MCP_ERR_INVALID_SPEC;
      }
      if (sc.distribution_change_prob > 0.0 && sc.new_distributions[i] == MCP_DIST_HIGH_KURTOSIS) need_t3 = true;
    }
    for (uint32_t j = 0; j < K; ++j)
      if (spec->var_types[j] == MCP_DIST_HIGH_KURTOSIS) need_t3 = true;
    auto note_resid = [&](int rd) {
      if (rd == MCP_RESID_HIGH_KURTOSIS || rd == MCP_RESID_RIGHT_SKEWED || rd == MCP_RESID_LEFT_SKEWED) may_chi = true;
      if (rd == MCP_RESID_UNIFORM) may_unif = true;
    };
    note_resid(spec->residual_dist);
    for (uint32_t i = 0; i < sc.n_residual_dists; ++i) {
      prm.resid_dists[i] = sc.residual_dists[i];
      if (sc.residual_change_prob > 0.0 && spec->residual_dist == MCP_RESID_NORMAL && !spec->residual_pinned)
        note_resid(sc.residual_dists[i]);
    }
    if (spec->outcome_kind != MCP_OUTCOME_CONTINUOUS) may_chi = may_unif = false;
    const double df = sc.residual_df > 3.0 ? sc.residual_df : 3.0;  // data_gen.rs:424
    prm.resid_df = df;
    long dfi = std::lround(df);
    if (dfi < 1) dfi = 1;
    if (may_chi && dfi > 64) return unsupported("residual df above 64");
    prm.resid_extra_cols = may_chi ? (uint32_t)dfi + 1u : 0u;
    prm.resid_may_uniform = may_unif ? 1u : 0u;
    low.need_t3 = need_t3;
    prm.het_total = sc.heterogeneity > 0.0 ? sc.heterogeneity : 0.0;
    low.need_beta_eff = prm.het_total > 1e-30;
    // heteroskedasticity (data_gen.rs:753-776) with the population moments of spec.rs:436-515
    const double lambda = sc.heteroskedasticity_ratio > 1.0 ? sc.heteroskedasticity_ratio : 1.0;
    const double gamma = std::log(lambda) / 4.0;
    double center = 0.0, sd = 0.0;
    if (gamma > 1e-30 && spec->outcome_kind == MCP_OUTCOME_CONTINUOUS) {
      const uint32_t pm = 1 + K + spec->n_factor_dummies;
      std::vector<double> mu(pm, 0.0), sdv(pm, 0.0);
      mu[0] = 1.0;
      for (uint32_t j = 0; j < K; ++j) {
        if (spec->var_types[j] == MCP_DIST_BINARY) {
          const double pp = j < spec->n_var_params ? spec->var_params[j] : 0.5;
          mu[1 + j] = pp;
          sdv[1 + j] = std::sqrt(std::fmax(pp * (1.0 - pp), 0.0));
        } else {
          sdv[1 + j] = 1.0;
        }
      }
      uint32_t c2 = 1 + K, off = 0;
      for (uint32_t f = 0; f < spec->n_factors; ++f) {
        const uint32_t nl = (uint32_t)spec->factor_n_levels[f];
        for (uint32_t d = 0; d + 1 < nl; ++d) {
          const double pp = spec->factor_proportions[off + d + 1];
          mu[c2 + d] = pp;
          sdv[c2 + d] = std::sqrt(std::fmax(pp * (1.0 - pp), 0.0));
        }
        c2 += nl - 1;
        off += nl;
      }
      if (spec->heteroskedasticity_driver < 0) {
        double m = 0.0, v = 0.0;
        for (uint32_t j = 0; j < pm; ++j) m += spec->effect_sizes[j] * mu[j];
        for (uint32_t j = 0; j < pm; ++j) v += spec->effect_sizes[j] * spec->effect_sizes[j] * sdv[j] * sdv[j];
        for (uint32_t j = 0; j < K; ++j)
          for (uint32_t t = j + 1; t < K; ++t)
            v += 2.0 * spec->effect_sizes[1 + j] * spec->effect_sizes[1 + t] * spec->correlation[t * K + j] * sdv[1 + j] * sdv[1 + t];
        center = m;
        sd = std::sqrt(std::fmax(v, 0.0));
      } else {
        center = mu[spec->heteroskedasticity_driver];
        sd = sdv[spec->heteroskedasticity_driver];
      }
    }
    prm.apply_hsk = (spec->outcome_kind == MCP_OUTCOME_CONTINUOUS && gamma > 1e-30 && sd > 1e-30) ? 1u : 0u;
    prm.hsk_driver = spec->heteroskedasticity_driver;
    prm.hsk_gamma = gamma;
    prm.hsk_center = center;
    prm.hsk_inv_std = prm.apply_hsk ? 1.0 / sd : 0.0;
    prm.hsk_inv_norm = std::exp(-gamma * gamma / 2.0);
  }

  // ---- reduced-design path: sparse-factor exclusion, test_formula subsets, post-hoc contrasts ----
  TukeyMemo memo;
  int rc = build_crit_rows_spec(spec, ns, n_ns, P, &memo, low.crit, err);
  if (rc) return rc;
  prm.crit_stride = nt + 4 + npc;
  prm.n_posthoc = npc;
  prm.min_level_count = spec->n_factors ? spec->factor_min_level_count : 0u;
  for (uint32_t t = 0; t < (uint32_t)MAX_T; ++t) prm.test_pos[t] = prm.test_neg[t] = 0xff;
  for (uint32_t t = 0; t < spec->n_target_indices; ++t) prm.test_pos[t] = (uint8_t)spec->target_indices[t];
  for (uint32_t q = 0; q < spec->n_contrast_pairs; ++q) {
    prm.test_pos[spec->n_target_indices + q] = (uint8_t)spec->contrast_pairs[2 * q];
    prm.test_neg[spec->n_target_indices + q] = (uint8_t)spec->contrast_pairs[2 * q + 1];
  }
  {  // build_posthoc_contrasts (batch.rs:2189-2223): per block, for a < b over levels, L[b] - L[a] with L[0] = reference
    uint32_t ci = 0;
    for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
      const uint32_t* d = spec->posthoc_target_indices + spec->posthoc_target_offsets[b];
      const uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
      for (uint32_t a = 0; a < k; ++a)
        for (uint32_t bb = a + 1; bb < k; ++bb) {
          prm.ph_pos[ci] = (uint8_t)d[bb - 1];
          prm.ph_neg[ci] = a > 0 ? (uint8_t)d[a - 1] : (uint8_t)0xff;
          ++ci;
        }
    }
  }
  const bool test_reduces = spec->n_fit_columns && spec->n_fit_columns < P;  // batch.rs:869
  // kept-column mask of every combination of per-factor exclusion flags (build_exclusion_remap, batch.rs:2506-2562)
  auto in_test = [&](uint32_t c) {
    if (spec->n_fit_columns == 0) return true;
    for (uint32_t i = 0; i < spec->n_fit_columns; ++i)
      if (spec->fit_columns[i] == c) return true;
    return false;
  };
  const uint32_t dummy_base = 1 + K, inter_base = dummy_base + spec->n_factor_dummies;
  for (uint32_t fb = 0; fb < 16u; ++fb) {
    uint32_t dropped_dummies = 0, keep = 0;
    for (uint32_t c = dummy_base; c < inter_base; ++c)
      if ((fb >> prm.col_factor[c]) & 1u) dropped_dummies |= 1u << c;
    for (uint32_t c = 0; c < P; ++c) {
      bool kept = in_test(c) && !((dropped_dummies >> c) & 1u);
      if (c >= inter_base && (prm.inter_mask[c] & dropped_dummies)) kept = false;
      if (kept) keep |= 1u << c;
    }
    prm.keep_mask[fb] = keep;
  }
  // which flag combinations can occur at each grid N: a fixed-allocation factor's level counts are the same in
  // every replicate (the host walks them); a sampled factor's are per replicate, so both values stay open
  low.red_row.assign((size_t)n_ns * 16u, 0xffffu);
  low.crit_red.clear();
  bool any_excl_possible = false;
  {
    std::vector<uint32_t> counts(spec->n_factor_proportions ? spec->n_factor_proportions : 1, 0u);
    std::vector<uint32_t> fixed_flag((size_t)n_ns * (spec->n_factors ? spec->n_factors : 1), 0u);
    uint32_t g = 0;
    for (uint32_t i = 0; i < max_n && g < n_ns && prm.min_level_count > 0; ++i) {
      for (uint32_t f = 0; f < spec->n_factors; ++f)
        if (!prm.f_sampled[f])
          counts[prm.f_prop_off[f] + low.fixed_levels[(size_t)i * spec->n_factors + f]] += 1;
      while (g < n_ns && i + 1 == ns[g]) {
        for (uint32_t f = 0; f < spec->n_factors; ++f)
          if (!prm.f_sampled[f])
            for (uint32_t l = 0; l < prm.f_levels[f]; ++l)
              if (counts[prm.f_prop_off[f] + l] < prm.min_level_count) fixed_flag[(size_t)g * spec->n_factors + f] = 1;
        ++g;
      }
    }
    struct Key { uint32_t n, p_red, row; };
    std::vector<Key> keys;
    for (g = 0; g < n_ns; ++g)
      for (uint32_t fb = 0; fb < (1u << spec->n_factors); ++fb) {
        bool reachable = true;
        for (uint32_t f = 0; f < spec->n_factors; ++f) {
          const uint32_t bit = (fb >> f) & 1u;
          if (prm.min_level_count == 0) reachable = reachable && bit == 0;
          else if (spec->estimator == MCP_EST_GLM) continue;  // the separation fallback may drop any included factor
          else if (!prm.f_sampled[f] && prm.boot_map[K + f] < 0)  // a bootstrapped factor's counts are per replicate
            reachable = reachable && bit == fixed_flag[(size_t)g * spec->n_factors + f];
        }
        if (!reachable) continue;
        if (fb) any_excl_possible = true;
        const uint32_t p_red = (uint32_t)__builtin_popcount(prm.keep_mask[fb]);
        if (p_red == 0 || (int64_t)ns[g] - (int64_t)p_red < 1) continue;
        uint32_t row = 0xffffu;
        for (const Key& kx : keys)
          if (kx.n == ns[g] && kx.p_red == p_red) row = kx.row;
        if (row == 0xffffu) {
          std::vector<double> one;
          rc = build_crit_rows_spec(spec, &ns[g], 1, p_red, &memo, one, err);
          if (rc) return rc;
          row = (uint32_t)(low.crit_red.size() / prm.crit_stride);
          low.crit_red.insert(low.crit_red.end(), one.begin(), one.end());
          keys.push_back({ns[g], p_red, row});
        }
        low.red_row[(size_t)g * 16u + fb] = (uint16_t)row;
      }
  }
  // designs beyond the hot kernels' register budget (more than 8 columns or 8 tests) run on the reduced-design
  // kernel too: its solve and decisions are sized at run time
  // (the fused OLS kernel is instantiated up to MAX_P_FUSED = 12 columns / 12 tests; the wave-per-replicate estimators
  // keep the 8 / 8 of their own kernels)
  // (general data path: 9 columns -- the scenario variant of the fused kernel spills from P = 10)
  const uint32_t hot_limit = spec->estimator == MCP_EST_OLS ? (general ? 9u : (uint32_t)MAX_P_FUSED) : (mle_wide ? 12u : (uint32_t)MAX_P_WAVE);
  const bool wide = P > hot_limit || nt > (spec->estimator == MCP_EST_OLS ? (uint32_t)MAX_P_FUSED : (mle_wide ? 12u : (uint32_t)MAX_T_WAVE));
  // clustered data fitted by OLS ignoring the clustering (orchestrator tests/clustered_ols_smoke.rs): the random
  // intercept of a row's cluster is re-drawn per row on the reduced-design kernel
  const bool ols_cluster = spec->cluster.present && spec->estimator == MCP_EST_OLS;
  // ... except the plain wide OLS design (P = 13 .. 16, <= 16 tests: the reference bench's ols_wide), which has a kernel of
  // its own: four waves share 64 replicates, every accumulator in an architectural register; a grid of sample sizes runs as
  // its split launch (ols_wide_kernels.hip)
  const bool wide_pair = wide && spec->estimator == MCP_EST_OLS && !general && P >= 13u && P <= (uint32_t)MAX_P &&
                         nt <= (uint32_t)MAX_T && npc == 0 && !test_reduces && !any_excl_possible && !ols_cluster;
  prm.red_active = (npc > 0 || test_reduces || any_excl_possible || (wide && !wide_pair) || ols_cluster) ? 1u : 0u;
  if (glmm) {
    prm.glmm_active = 1;
    if (spec->wald_se > MCP_WALD_SE_HESSIAN) { err = "invalid spec: wald_se out of range"; return MCP_ERR_INVALID_SPEC; }
    // validate.rs:918-945: odd, <= 25
    if (spec->nagq > 25 || (spec->nagq > 1 && !(spec->nagq & 1u))) { err = "invalid spec: nagq must be odd and in 1..=25"; return MCP_ERR_INVALID_SPEC; }
    // validate.rs:918-941: one grouping, <= 3 random effects (the k^q tensor rule of glmm_slopes_kernel); extra groupings never
    if (spec->nagq > 1 && spec->cluster.n_extra_groupings) return unsupported("adaptive Gauss-Hermite quadrature (nagq > 1) with extra groupings (the reference admits one grouping with <= 3 random effects)");
    prm.glmm_nagq = spec->nagq > 1 ? spec->nagq : 1u;
    if (prm.glmm_nagq > 1) gauss_hermite_normal(prm.glmm_nagq, prm.glmm_gz, prm.glmm_gw);
    prm.glmm_hessian = spec->wald_se == MCP_WALD_SE_HESSIAN ? 1u : 0u;
    prm.glmm_keep = prm.keep_mask[0];  // every column, or the test_formula subset (flag combination 0: nothing excluded)
    prm.glmm_pirls_cap = 60;  // = oracle GG_PIRLS_ITERS; the diagnostic knob below only ever lowers it, so that a test can
    if (const char* cap = std::getenv("MCP_DEBUG_GLMM_PIRLS_CAP")) {  // drive every evaluation into the cap-exit branch
      const long v = std::strtol(cap, nullptr, 10);
      if (v >= 1 && v <= 60) prm.glmm_pirls_cap = (uint32_t)v;
    }
    if (!(prm.glmm_keep & 1u)) return unsupported("intercept-free test_formula fit of a clustered GLM");
    for (uint32_t t = 0; t <= spec->cluster.n_extra_groupings; ++t) {
      const mcp_re_slopes& sl = t == 0 ? spec->cluster.slopes : spec->cluster.extra[t - 1].slopes;
      for (uint32_t d = 0; d < sl.n_slopes; ++d)
        if (!((prm.glmm_keep >> sl.design_cols[d]) & 1u)) return unsupported("test_formula dropping a column that carries a random slope");
    }
    prm.red_active = 0;  // no factor exclusion on clustered specs (batch.rs:1583)
  }
  if (prm.red_active && spec->estimator != MCP_EST_OLS && !((prm.keep_mask[0] >> 0) & 1u))
    return unsupported("intercept-free reduced fit under Glm / Mle");
  // general LMM: compact strips (mcp_params.hpp) unless a reduced round can occur -- that round reads whole strips
  if (spec->estimator == MCP_EST_MLE && prm.lmg_active && !prm.red_active && prm.lmg_coff_xy + P + 1 < prm.lmg_fW) {
    prm.lmg_compact = 1;
    prm.lmg_Wc = prm.lmg_coff_xy + P + 1;
  }
  for (uint32_t k = 0; k <= (uint32_t)MAX_T; ++k)
    prm.joint_crit_k[k] = k == 0 ? std::numeric_limits<double>::infinity() : chi2_ppf(1.0 - spec->alpha, (double)k);
  low.P = P;
  low.n_tests = nt;
  return MCP_OK;
}

}  // namespace mcp
