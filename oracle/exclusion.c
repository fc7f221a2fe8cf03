/*
 * oracle/exclusion.c -- CPU ORACLE (test infrastructure): sparse-factor exclusion and the reduced
 * (test_formula) design bookkeeping shared by the three estimator arms.
 *
 * Follows /root/reference/crates/engine-core/src/batch.rs
 *   update_factor_exclusions :2451-2500 (per-level prefix counts read back from the dummy blocks;
 *     flag 1 = some level of the factor has fewer than factor_min_level_count rows),
 *   build_exclusion_remap :2506-2562 (kept columns ascending: intercept + continuous, dummies of
 *     non-flagged factors, interactions none of whose components is a dropped dummy; all intersected
 *     with fit_columns when that is non-empty),
 *   included_factor_with_smallest_min_count :2573-2600 (GLM separation fallback candidate),
 *   reduced_crit_entry :2418-2440 (df-dependent thresholds rebuilt at p_red columns by the same
 *     table constructor as the full-model ones).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

void mcpo_update_factor_exclusions(const float* x, uint32_t ld, const mcp_spec* spec, uint32_t from, uint32_t to,
                                   uint32_t* prefix_counts, uint8_t* flags) {
  uint32_t prop_off = 0, col = 1 + spec->n_non_factor;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    uint32_t width = l ? l - 1 : 0;
    for (uint32_t i = from; i < to; ++i) {
      uint32_t lvl = 0;
      for (uint32_t d = 0; d < width; ++d)
        if (x[(size_t)(col + d) * ld + i] == 1.0f) { lvl = d + 1; break; }
      prefix_counts[prop_off + lvl] += 1;
    }
    prop_off += l;
    col += width;
  }
  prop_off = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    uint32_t min_c = l ? UINT32_MAX : 0;
    for (uint32_t g = 0; g < l; ++g)
      if (prefix_counts[prop_off + g] < min_c) min_c = prefix_counts[prop_off + g];
    flags[f] = (uint8_t)(spec->factor_min_level_count > 0 && min_c < spec->factor_min_level_count);
    prop_off += l;
  }
}

static int in_test_(const mcp_spec* spec, uint32_t col) {
  if (spec->n_fit_columns == 0) return 1;
  for (uint32_t i = 0; i < spec->n_fit_columns; ++i)
    if (spec->fit_columns[i] == col) return 1;
  return 0;
}

uint32_t mcpo_build_exclusion_remap(const mcp_spec* spec, const uint8_t* flags, uint32_t* kept_cols,
                                    int32_t* col_remap) {
  const uint32_t dummy_base = 1 + spec->n_non_factor, n_fd = spec->n_factor_dummies;
  const uint32_t inter_base = dummy_base + n_fd, p = inter_base + spec->n_interactions;
  uint8_t keep_dummy[MCPO_MAX_P];
  for (uint32_t c = 0; c < p; ++c) col_remap[c] = -1;
  for (uint32_t d = 0; d < n_fd; ++d) keep_dummy[d] = 1;
  uint32_t col = dummy_base, n_kept = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    uint32_t width = l ? l - 1 : 0;
    if (flags && flags[f] != 0)
      for (uint32_t d = 0; d < width; ++d) keep_dummy[col + d - dummy_base] = 0;
    col += width;
  }
  for (uint32_t c = 0; c < dummy_base; ++c)
    if (in_test_(spec, c)) { col_remap[c] = (int32_t)n_kept; kept_cols[n_kept++] = c; }
  for (uint32_t d = 0; d < n_fd; ++d) {
    uint32_t c = dummy_base + d;
    if (keep_dummy[d] && in_test_(spec, c)) { col_remap[c] = (int32_t)n_kept; kept_cols[n_kept++] = c; }
  }
  for (uint32_t j = 0; j < spec->n_interactions; ++j) {
    uint32_t c = inter_base + j;
    int dropped = 0;
    for (uint32_t q = spec->interaction_offsets[j]; q < spec->interaction_offsets[j + 1]; ++q) {
      uint32_t cu = spec->interaction_components[q];
      if (cu >= dummy_base && cu < inter_base && !keep_dummy[cu - dummy_base]) dropped = 1;
    }
    if (!dropped && in_test_(spec, c)) { col_remap[c] = (int32_t)n_kept; kept_cols[n_kept++] = c; }
  }
  return n_kept;
}

int mcpo_smallest_min_count_factor(const mcp_spec* spec, const uint32_t* prefix_counts, const uint8_t* flags) {
  int best = -1;
  uint32_t best_min = 0, prop_off = 0;
  for (uint32_t f = 0; f < spec->n_factors; ++f) {
    uint32_t l = spec->factor_n_levels[f] > 0 ? (uint32_t)spec->factor_n_levels[f] : 0;
    if (flags[f] == 0) {
      uint32_t min_c = l ? UINT32_MAX : 0;
      for (uint32_t g = 0; g < l; ++g)
        if (prefix_counts[prop_off + g] < min_c) min_c = prefix_counts[prop_off + g];
      if (best < 0 || min_c < best_min) { best = (int)f; best_min = min_c; }
    }
    prop_off += l;
  }
  return best;
}

/* (n, p_red) -> thresholds, memoised per worker */
const mcpo_red_crit* mcpo_reduced_crit(mcpo_red_cache* cache, const mcp_spec* spec, uint32_t n, uint32_t p_red) {
  for (uint32_t i = 0; i < cache->count; ++i)
    if (cache->e[i].n == n && cache->e[i].p_red == p_red) return &cache->e[i];
  if (cache->count == cache->cap) {
    uint32_t cap = cache->cap ? cache->cap * 2 : 16;
    mcpo_red_crit* ne = (mcpo_red_crit*)realloc(cache->e, sizeof(mcpo_red_crit) * cap);
    if (!ne) return NULL;
    cache->e = ne;
    cache->cap = cap;
  }
  uint32_t nt = spec->n_target_indices + spec->n_contrast_pairs, npc = 0;
  for (uint32_t b = 0; b < spec->n_posthoc; ++b) {
    uint32_t k = spec->posthoc_target_offsets[b + 1] - spec->posthoc_target_offsets[b] + 1;
    npc += k * (k - 1) / 2;
  }
  mcpo_red_crit* e = &cache->e[cache->count];
  e->n = n;
  e->p_red = p_red;
  e->row = (double*)malloc(sizeof(double) * (nt + 4));
  e->post = (double*)malloc(sizeof(double) * (npc ? npc : 1));
  if (!e->row || !e->post) return NULL;
  if (mcpo_build_crit_table_p(spec, &n, 1, p_red, NULL, e->row) != MCP_OK) return NULL;
  if (npc && mcpo_build_posthoc_corr(spec, &n, 1, p_red, e->post) != MCP_OK) return NULL;
  cache->count += 1;
  return e;
}

void mcpo_red_cache_free(mcpo_red_cache* cache) {
  for (uint32_t i = 0; i < cache->count; ++i) { free(cache->e[i].row); free(cache->e[i].post); }
  free(cache->e);
  memset(cache, 0, sizeof *cache);
}
