// ols_reduced_kernels.hip -- gfx950 OLS kernel for rounds whose fitted model can differ from the generated
// design: per-(replicate, N) sparse-factor exclusion, a test_formula column subset, post-hoc pairwise contrasts.
//
// Same mapping as ols_kernels.hip (one lane per replicate, 4-row Philox blocks staged in LDS, sufficient
// statistics in registers, grid-incremental fits).  What changes is the solve:
//   * per-level row counts of every factor are kept in saturating 4-bit lanes of one u64 per factor and
//     compared with factor_min_level_count at each grid N (update_factor_exclusions, batch.rs:2451-2500);
//   * the flag combination selects a kept-column bitmask (build_exclusion_remap :2506-2562, fit_columns folded
//     in on the host).  Instead of gathering a p_red x p_red system, dropped columns are replaced by identity
//     rows/columns of X'X (and a zero of X'y): the Cholesky factor, beta, rss and every ||L^-1 c||^2 of the
//     kept columns come out of exactly the same operations as the gathered system would perform, plus adds
//     and multiplies by exact zeros, so the statistics equal the reference's reduced fit (batch.rs:970-1013);
//   * df = N - p_red, and every df-dependent threshold of the round is read from the row the host built at
//     p_red fitted columns (reduced_crit_entry :2418-2440);
//   * a test or post-hoc contrast touching a dropped column carries NaN and can never pass (:1060-1088,
//     :1164-1199); the post-hoc family is corrected over its own contrast set (posthoc.rs:41-122).
// Flags are written per replicate; fold_flags_kernel turns them into the additive counters.
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_lmg_row.hpp"
#include "mcp_suff.hpp"

#pragma clang fp contract(off)

namespace mcp {

constexpr int MAX_ALL = MAX_T > MAX_PC ? MAX_T : MAX_PC;

// correction.rs:16-115 over a runtime-sized family held in private memory (cold path; NaN never passes)
__device__ __noinline__ uint32_t decide_family(const double* tsq, uint32_t n, double crit_unc, const double* corr,
                                               int method, uint32_t& unc_bits) {
  unc_bits = 0;
  uint32_t cor_bits = 0;
  for (uint32_t t = 0; t < n; ++t)
    if (tsq[t] > crit_unc) unc_bits |= 1u << t;
  if (method == 0) return unc_bits;
  if (method == 1 || method == 4) {  // Bonferroni / Tukey: per-target thresholds
    for (uint32_t t = 0; t < n; ++t)
      if (tsq[t] > corr[t]) cor_bits |= 1u << t;
    return cor_bits;
  }
  // rank of every statistic in the descending order (ties by index, NaN last)
  uint32_t rank[MAX_ALL];
  uint32_t okb = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t rk = 0;
    const bool in = tsq[i] != tsq[i];
    for (uint32_t j = 0; j < n; ++j) {
      if (j == i) continue;
      const bool jn = tsq[j] != tsq[j];
      bool before;
      if (in && jn) before = j < i;
      else if (jn) before = false;
      else if (in) before = true;
      else before = (tsq[j] > tsq[i]) || (tsq[j] == tsq[i] && j < i);
      rk += before ? 1u : 0u;
    }
    rank[i] = rk;
    if (tsq[i] > corr[rk]) okb |= 1u << i;
  }
  if (method == 2) {  // Holm step-down
    for (uint32_t t = 0; t < n; ++t) {
      bool pass = true;
      for (uint32_t j = 0; j < n; ++j)
        if (rank[j] <= rank[t] && !((okb >> j) & 1u)) pass = false;
      if (pass) cor_bits |= 1u << t;
    }
  } else {  // BH step-up
    int last = -1;
    for (uint32_t t = 0; t < n; ++t)
      if (((okb >> t) & 1u) && (int)rank[t] > last) last = (int)rank[t];
    for (uint32_t t = 0; t < n; ++t)
      if ((int)rank[t] <= last) cor_bits |= 1u << t;
  }
  return cor_bits;
}

template <int P>
__device__ __forceinline__ void fit_decide_reduced(const SimParams& prm, const Suff<P>& s, uint32_t g, uint32_t n,
                                                   uint32_t flag_bits, bool active, uint32_t out_row) {
  const uint32_t keep = prm.keep_mask[flag_bits & 15u];
  const uint32_t p_red = (uint32_t)__popc(keep);
  // ---- X'X (lower) and X'y with identity rows/columns at the dropped positions ----
  double A[P][P];
  double rhs[P];
  {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < P; ++a) {
      const bool ka = (keep >> a) & 1u;
      const double ra = a == 0 ? s.sy : s.sxy[a > 0 ? a - 1 : 0];
      rhs[a] = ka ? ra : 0.0;
#pragma unroll
      for (int b = 0; b <= a; ++b) {
        double v;
        if (a == 0) v = (double)n;
        else if (b == 0) v = s.sx[a - 1];
        else v = s.sxx[idx++];
        const bool kb = (keep >> b) & 1u;
        A[a][b] = (ka && kb) ? v : (a == b ? 1.0 : 0.0);
      }
    }
  }
  double Lm[P][P];
  double invd[P];
  bool ok = true;
  double dmin = __builtin_inf(), dmax = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int t = 0; t < j; ++t) d = __builtin_fma(-Lm[j][t], Lm[j][t], d);
    ok = ok && (d > 0.0) && (d < __builtin_inf());
    const double ljj = __builtin_sqrt(d);
    Lm[j][j] = ljj;
    invd[j] = 1.0 / ljj;
    if ((keep >> j) & 1u) {  // the rank test runs over the fitted columns only
      dmin = ljj < dmin ? ljj : dmin;
      dmax = ljj > dmax ? ljj : dmax;
    }
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int t = 0; t < j; ++t) v = __builtin_fma(-Lm[i][t], Lm[j][t], v);
      Lm[i][j] = v * invd[j];
    }
  }
  const bool conv = ok && p_red > 0u && (dmin / dmax >= EPS_RANK);
  double zf[P], beta[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = rhs[i];
#pragma unroll
    for (int t = 0; t < i; ++t) v = __builtin_fma(-Lm[i][t], zf[t], v);
    zf[i] = v * invd[i];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = zf[i];
#pragma unroll
    for (int t = i + 1; t < P; ++t) v = __builtin_fma(-Lm[t][i], beta[t], v);
    beta[i] = v * invd[i];
  }
  double bxty = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) bxty = __builtin_fma(beta[i], rhs[i], bxty);
  const double nd = (double)n;
  const double rss = s.syy - bxty;
  const double df = nd - (double)p_red;
  const double sigma_sq = rss / df;
  const double sst = s.syy - s.sy * s.sy / nd;

  // squared statistic of e_pos - e_neg (neg = 0xff: a single column); NaN when a side is not fitted
  auto pair_t_sq = [&](uint32_t pos, uint32_t neg) -> double {
    const bool has_neg = neg != 0xffu;
    if (!((keep >> pos) & 1u) || (has_neg && !((keep >> neg) & 1u))) return __builtin_nan("");
    double est = 0.0, nq = 0.0;
    double u[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double ci = ((uint32_t)i == pos ? 1.0 : 0.0);
      ci = (has_neg && (uint32_t)i == neg) ? ci - 1.0 : ci;
      est = __builtin_fma(ci, beta[i], est);
      double acc = ci;
#pragma unroll
      for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lm[i][q], u[q], acc);
      u[i] = acc * invd[i];
      nq = __builtin_fma(u[i], u[i], nq);
    }
    const double se_sq = sigma_sq * nq;
    return (conv && se_sq > FLOAT_NEAR_ZERO && se_sq < __builtin_inf()) ? est * est / se_sq : __builtin_nan("");
  };

  const uint32_t row_idx = prm.red_row[(size_t)g * 16u + (flag_bits & 15u)];
  const double* crow = prm.crit_red + (size_t)row_idx * prm.crit_stride;
  const uint32_t nt = prm.n_tests, npc = prm.n_posthoc;
  double tsq[MAX_ALL];
  for (uint32_t t = 0; t < nt; ++t) tsq[t] = pair_t_sq(prm.test_pos[t], prm.test_neg[t]);
  uint32_t unc_bits = 0, cor_bits = 0;
  if (conv) cor_bits = decide_family(tsq, nt, crow[0], crow + 1, prm.correction, unc_bits);
  bool ov = false;
  if (prm.report_overall && conv) {  // batch.rs:1118-1142, numerator df = p_red - 1
    const double dfn = (double)p_red - 1.0;
    if (dfn >= 1.0 && sst > 1e-10 && rss > 0.0 && rss < __builtin_inf() && sst < __builtin_inf()) {
      const double f = ((sst - rss) / dfn) / (rss / df);
      ov = (f < __builtin_inf()) && (f > crow[3 + nt]);
    }
  }
  if (active) {
    const size_t cell = (size_t)out_row * prm.n_grid + g;
    for (uint32_t t = 0; t < nt; ++t) {
      prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
      prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
    }
    prm.conv[cell] = conv ? 1 : 0;
    if (prm.overall) prm.overall[cell] = ov ? 1 : 0;
    if (prm.stat && g == 0)
      for (uint32_t t = 0; t < nt; ++t) prm.stat[(size_t)out_row * nt + t] = tsq[t];
    for (uint32_t f = 0; f < prm.n_factors; ++f)
      prm.factor_excluded[cell * prm.n_factors + f] = (uint8_t)((flag_bits >> f) & 1u);
  }
  if (npc) {  // posthoc.rs:41-122
    uint32_t pu = 0, pc = 0;
    if (conv) {
      for (uint32_t c = 0; c < npc; ++c) tsq[c] = pair_t_sq(prm.ph_pos[c], prm.ph_neg[c]);
      pc = decide_family(tsq, npc, crow[1 + nt], crow + 4 + nt, prm.correction, pu);
    }
    if (active) {
      const size_t cell = (size_t)out_row * prm.n_grid + g;
      for (uint32_t c = 0; c < npc; ++c) {
        prm.posthoc_unc[cell * npc + c] = (uint8_t)((pu >> c) & 1u);
        prm.posthoc_cor[cell * npc + c] = (uint8_t)((pc >> c) & 1u);
      }
    }
  }
}

// FLAVOR 1: factor dummies on the fast data path; FLAVOR 2: general ("scenario") data path
template <int P, int FLAVOR>
__global__ __launch_bounds__(BLOCK) void ols_reduced_kernel(const SimParams* __restrict__ pp) {
  constexpr bool SCEN = FLAVOR == 2;
  extern __shared__ float stage_lds[];
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = local < prm.n_sims;
  // general path: lanes take the replicates of the launch in transform-class order (scenario_kernels.hip)
  const uint32_t lsim = active ? ((SCEN && prm.scen_perm) ? prm.scen_perm[local] : local) : (prm.n_sims - 1u);
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t out_row = prm.out_base + lsim;
  float* st = stage_lds + threadIdx.x;
  RowCtx<P> cx;
  row_ctx_init<P, true>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);

  double tau_rep = prm.tau;
  if (SCEN && prm.scen_tau_sq) {
    const double t2 = prm.scen_tau_sq[lsim];
    tau_rep = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
  }
  // cluster-level (between) predictors under OLS on clustered data (data_gen.rs:606-636): a row takes the marked columns from
  // its cluster's representative row, then interactions and the outcome follow from the broadcast values.  Rows come in
  // order and the representative never comes after the row, so one lane can do it on the fly: FixedSize clusters are
  // contiguous (the current cluster's values stay in registers), FixedClusters interleave (row i belongs to cluster
  // i mod v, its representative is row i mod v: the first v rows park their marked values in LDS behind the staging area)
  const uint32_t bmask = prm.has_cluster ? prm.between_mask : 0u;
  const uint32_t nb = (uint32_t)__popc(bmask);
  float xrep[P];
#pragma unroll
  for (int c = 0; c < P; ++c) xrep[c] = 0.0f;
  float* bl = stage_lds + (size_t)(SCEN ? (P + 1) : P) * 4u * BLOCK + threadIdx.x;
  Suff<P> s;
  suff_reset(s);
  unsigned long long level_counts[MAX_FACTORS] = {0ull, 0ull, 0ull, 0ull};  // 4-bit saturating count per level
  const uint32_t min_c = prm.min_level_count;
  uint32_t row = 0;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n_end = prm.ns[g];
    while (row < n_end) {
      const uint32_t base = row & ~3u;
      if ((row & 3u) == 0u) stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, row >> 2, st, SCEN ? sl.meta : 0u);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t rr = base + (uint32_t)r;
        if (rr >= row && rr < n_end) {
          float x[P], y;
          double u_re = 0.0;
          if (prm.has_cluster) {  // cluster random intercept: draw c of the sequential stream (data_gen.rs:146-161)
            const uint32_t cid = prm.cluster_fixed_clusters ? rr % prm.cluster_v : rr / prm.cluster_v;
            u_re = (double)(float)(draw_cluster_effect(prm, k0, k1, cid < prm.n_clusters_max ? cid : 0u) * tau_rep);
          }
          if (bmask == 0u) {
            build_row<P, true, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, u_re, &sl);
          } else {
            const float e = build_x<P, true, BLOCK, SCEN>(prm, cx, k0, k1, st, r, rr, x, &sl);
            const uint32_t cvb = prm.cluster_v;
            const bool fixed = prm.cluster_fixed_clusters != 0u;
            const uint32_t repr = fixed ? rr % cvb : rr - rr % cvb;
            uint32_t slot = 0;
#pragma unroll
            for (int c = 1; c < P; ++c)
              if ((bmask >> c) & 1u) {
                if (fixed) {
                  float* cell = bl + (size_t)(repr * nb + slot) * BLOCK;
                  if (rr == repr) *cell = x[c];
                  else x[c] = *cell;
                } else {
                  if (rr == repr) xrep[c] = x[c];
                  else x[c] = xrep[c];
                }
                ++slot;
              }
            rebuild_interactions<P>(cx, x);
            finish_row<P, 0, SCEN>(prm, cx, x, e, y, u_re, &sl);
          }
          suff_add_row<P>(s, x, y);
          if (min_c) {  // level of every factor read back from its dummy block (exact 0.0 / 1.0)
#pragma unroll
            for (int f = 0; f < MAX_FACTORS; ++f) {
              if ((uint32_t)f < prm.n_factors) {
                uint32_t lv = 0;
#pragma unroll
                for (int c = 1; c < P; ++c)
                  if (prm.col_factor[c] == (uint32_t)f && x[c] == 1.0f) lv = prm.col_level[c];
                const unsigned long long cur = (level_counts[f] >> (4u * lv)) & 15ull;
                if (cur < 15ull) level_counts[f] += 1ull << (4u * lv);
              }
            }
          }
        }
      }
      const uint32_t nxt = base + 4u;
      row = nxt < n_end ? nxt : n_end;
    }
    uint32_t flag_bits = 0;
    if (min_c) {
#pragma unroll
      for (int f = 0; f < MAX_FACTORS; ++f) {
        if ((uint32_t)f < prm.n_factors) {
          bool sparse = false;
          for (uint32_t l = 0; l < prm.f_levels[f]; ++l)
            sparse = sparse || (uint32_t)((level_counts[f] >> (4u * l)) & 15ull) < min_c;
          flag_bits |= sparse ? (1u << f) : 0u;
        }
      }
    }
    fit_decide_reduced<P>(prm, s, g, n_end, flag_bits, active, out_row);
  }
}

// ---- introspect::capture_sim0_data twin (introspect.rs:104-155) for every outcome kind: the generated design
// and outcome of a replicate range in the staged tile layout ([block][column][row quad][256 lanes][4 rows]).
// Discrete outcomes (logit / probit / Poisson) and the cluster random intercept (draw c of the sequential
// stream, data_gen.rs:146-161) go through the same row builder the simulate kernels use. ----
template <int P, bool SCEN>
__global__ __launch_bounds__(BLOCK) void design_capture_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ float stage_lds[];
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = local < prm.n_sims;
  const uint32_t lsim = active ? local : (prm.n_sims - 1u);
  const uint64_t key = pcg_mix64(prm.base_seed, prm.sim_begin + lsim);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  float* st = stage_lds + threadIdx.x;
  RowCtx<P> cx;
  row_ctx_init<P, true>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  const bool discrete = prm.outcome_kind != 0u;
  const uint32_t n_quads = (prm.max_n + 3u) >> 2;
  for (uint32_t ctr = 0; ctr < n_quads; ++ctr) {
    if (discrete) stage_group<P, BLOCK, 1, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u);
    else stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u);
    float q[P][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      double u_re = 0.0;
      if (prm.has_cluster && !prm.lmg_active) {
        const uint32_t cid = prm.cluster_fixed_clusters ? rr % prm.cluster_v : rr / prm.cluster_v;
        double tau_rep = prm.tau;
        if (SCEN && prm.scen_tau_sq) {
          const double t2 = prm.scen_tau_sq[lsim];
          tau_rep = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
        }
        u_re = (double)(float)(draw_cluster_effect(prm, k0, k1, cid < prm.n_clusters_max ? cid : 0u) * tau_rep);
      }
      float x[P], y;
      if (prm.lmg_active) {  // random slopes / extra groupings: the row's effect needs its own columns (slopes)
        const float e = build_x<P, true, BLOCK, SCEN>(prm, cx, k0, k1, st, r, rr, x, &sl);
        const double u_gen = lmg_row_effect_redraw<P, SCEN>(prm, k0, k1, lsim, rr, x);
        if (discrete) finish_row<P, 1, SCEN>(prm, cx, x, e, y, u_gen, &sl);
        else finish_row<P, 0, SCEN>(prm, cx, x, e, y, u_gen, &sl);
      } else if (discrete) build_row<P, true, BLOCK, 1, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, u_re, &sl);
      else build_row<P, true, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, u_re, &sl);
#pragma unroll
      for (int c = 1; c < P; ++c) q[c - 1][r] = x[c];
      q[P - 1][r] = y;
    }
#pragma unroll
    for (int c = 0; c < P; ++c)
      *reinterpret_cast<float4*>(prm.tile + ((((size_t)blockIdx.x * P + c) * n_quads + ctr) * BLOCK) * 4u + threadIdx.x * 4u) =
          make_float4(q[c][0], q[c][1], q[c][2], q[c][3]);
  }
}

hipError_t launch_design_capture(int P, const SimParams* d_prm, uint32_t n_sims, bool scen, hipStream_t st) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1) / BLOCK);
#define LC(PP)                                                                                                   \
  case PP: {                                                                                                     \
    const size_t lds = (size_t)(PP + 1) * 4u * BLOCK * sizeof(float);                                            \
    if (scen) hipLaunchKernelGGL((design_capture_kernel<PP, true>), grid, block, lds, st, d_prm);                \
    else hipLaunchKernelGGL((design_capture_kernel<PP, false>), grid, block, lds, st, d_prm);                    \
  } break;
  switch (P) {
    LC(2) LC(3) LC(4) LC(5) LC(6) LC(7) LC(8) LC(9) LC(10) LC(11) LC(12) LC(13) LC(14) LC(15) LC(16)
    default: return hipErrorInvalidValue;
  }
#undef LC
  return hipGetLastError();
}

// between_floats: floats per lane of the cluster-level predictor park (clusters x marked columns under FixedClusters, else 0)
template <int FLAVOR>
static hipError_t launch_reduced_f(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t between_floats, hipStream_t st) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1) / BLOCK);
#define LR(PP)                                                                                                   \
  case PP: {                                                                                                     \
    const size_t lds = ((size_t)(FLAVOR == 2 ? (PP + 1) : PP) * 4u + between_floats) * BLOCK * sizeof(float);    \
    if (lds > 48 * 1024) {                                                                                       \
      hipError_t e = hipFuncSetAttribute((const void*)ols_reduced_kernel<PP, FLAVOR>,                            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      if (e != hipSuccess) return e;                                                                             \
    }                                                                                                            \
    hipLaunchKernelGGL((ols_reduced_kernel<PP, FLAVOR>), grid, block, lds, st, d_prm);                           \
  } break;
  switch (P) {
    LR(2) LR(3) LR(4) LR(5) LR(6) LR(7) LR(8) LR(9) LR(10) LR(11) LR(12) LR(13) LR(14) LR(15) LR(16)
    default: return hipErrorInvalidValue;
  }
#undef LR
  return hipGetLastError();
}

hipError_t launch_ols_reduced(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, uint32_t between_floats, hipStream_t st) {
  return flavor == 2 ? launch_reduced_f<2>(P, d_prm, n_sims, between_floats, st) : launch_reduced_f<1>(P, d_prm, n_sims, between_floats, st);
}

}  // namespace mcp
