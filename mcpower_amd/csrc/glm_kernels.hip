// glm_kernels.hip -- gfx950 kernels of the logistic-GLM (IRLS) simulate path.
//
// Mapping: ONE WAVEFRONT PER REPLICATE.  The replicate's n x p design (f32) and outcome are
// generated straight into LDS ([column][row], lanes read consecutive rows: conflict-free) and stay
// there for every IRLS pass, so the re-weighting loop never touches HBM.  Rows are spread over the
// 64 lanes; each pass ends in one butterfly all-reduce of the p(p+1)/2 + p + 1 accumulators
// (X'WX, X'Wz, deviance), after which every lane holds identical sums and runs the p x p Cholesky
// solve redundantly -- the iteration count is wave-uniform, so replicates that need more passes
// do not hold other replicates back (unlike a lane-per-replicate mapping).
//
// Reference: crates/engine-core/src/batch.rs:1562-2134 (GLM arm of run_one_sim: refit per grid N
// from the truth start beta0 = effect_sizes, Wald z^2 vs norm_ppf^2, LRT overall),
// data_gen.rs:390-398, 859-911 (uniform residual slot, Bernoulli draw).  The IRLS arithmetic
// replaces glmm 0.1.2's glm_irls_fit (out of tree): published Fisher scoring, R glm.fit's
// convergence rule |dev - dev_old| / (|dev| + 0.1) < 1e-8, <= 25 passes, |beta| > 30 -> not converged.
#ifndef MCP_GLM_WAVES
#define MCP_GLM_WAVES 1  // waves per SIMD floor for the register allocator: 2..4 change C3 by < 3% (profiles/r01_j_occupancy.txt)
#endif
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"
#include "mcp_glm_family.hpp"

#pragma clang fp contract(off)

namespace mcp {

constexpr int WAVE = 64;
constexpr int MAX_IRLS_ITERS = 25;
constexpr double IRLS_EPS = 1e-8;
constexpr double BETA_CAP = 30.0;  // batch.rs:4550
constexpr double GLM_NEAR_ZERO = 1e-30;

// All-reduce over the wave without LDS traffic: rotations inside each row of 16 lanes (DPP row_ror 8, 4, 2, 1 --
// every lane ends with its row's total, bitwise the same in all 16 because each step adds a commuting pair),
// then the four row totals read through scalar registers and summed in one order by every lane.  The previous
// xor butterfly cost 12 ds_bpermute per value (180 per IRLS pass at p = 4) and their LDS latency with 2-3 waves
// per SIMD.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_allreduce_sum(double v) {
  v += dpp_f64<0x128>(v);  // row_ror:8
  v += dpp_f64<0x124>(v);  // row_ror:4
  v += dpp_f64<0x122>(v);  // row_ror:2
  v += dpp_f64<0x121>(v);  // row_ror:1
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// IRLS on the first n rows of the LDS-resident design, all lanes returning identical values.
// MASKED: `keep` selects the fitted columns; dropped columns become identity rows/columns of X'WX and zeros of
// X'Wz, so their coefficients stay exactly 0 and the kept block is solved by the same operations as the gathered
// p_red x p_red system of the reference's reduced round (batch.rs:1893-1925).
// LOGIT: the family is known at compile time (the probit / Poisson arms and their ocml expansions stay out of the
// kernel's register allocation: 180 -> fewer VGPRs, one more wave per SIMD).
template <int P, bool MASKED, bool LOGIT>
__device__ __forceinline__ void irls_fit(const float* design, uint32_t n_pad, uint32_t n, uint32_t lane, uint32_t keep,
                                         uint32_t fam_rt, double (&beta)[P], double (&Lm)[P][P], double (&invd)[P],
                                         double& dev, bool& conv) {
  const uint32_t fam = LOGIT ? 0u : fam_rt;
  dev = __builtin_nan("");
  double dev_prev = __builtin_inf();
  conv = false;
  for (int it = 0; it < MAX_IRLS_ITERS; ++it) {
    double xtwx[P * (P + 1) / 2], xtwz[P], d = 0.0;
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) xtwx[i] = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) xtwz[i] = 0.0;
    for (uint32_t i = lane; i < n; i += WAVE) {
      double xd[P];
      xd[0] = 1.0;
#pragma unroll
      for (int c = 1; c < P; ++c) xd[c] = (double)design[(size_t)(c - 1) * n_pad + i];
      const double yi = (double)design[(size_t)(P - 1) * n_pad + i];
      double eta = 0.0;
#pragma unroll
      for (int c = 0; c < P; ++c) eta = __builtin_fma(xd[c], beta[c], eta);
      double w, wz, di;
      fam_eval(fam, eta, yi, w, wz, di);
      d += di;
      int idx = 0;
#pragma unroll
      for (int a = 0; a < P; ++a) {
        const double wxa = w * xd[a];
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          xtwx[idx] = __builtin_fma(wxa, xd[b], xtwx[idx]);
          ++idx;
        }
        xtwz[a] = __builtin_fma(xd[a], wz, xtwz[a]);
      }
    }
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) xtwx[i] = wave_allreduce_sum(xtwx[i]);
#pragma unroll
    for (int i = 0; i < P; ++i) xtwz[i] = wave_allreduce_sum(xtwz[i]);
    dev = wave_allreduce_sum(d);
    if (!(dev < __builtin_inf())) break;  // NaN / inf deviance
    const bool done = it > 0 && __builtin_fabs(dev - dev_prev) / (__builtin_fabs(dev) + 0.1) < IRLS_EPS;
    // Cholesky of X'WX and the weighted least-squares update (all lanes, identical values)
    bool ok = true;
    {
      int idx = 0;
      double A[P][P];
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          const double v = xtwx[idx++];
          A[a][b] = (!MASKED || (((keep >> a) & 1u) && ((keep >> b) & 1u))) ? v : (a == b ? 1.0 : 0.0);
        }
#pragma unroll
      for (int j = 0; j < P; ++j) {
        double dd = A[j][j];
#pragma unroll
        for (int t = 0; t < j; ++t) dd = __builtin_fma(-Lm[j][t], Lm[j][t], dd);
        ok = ok && (dd > 0.0) && (dd < __builtin_inf());
        const double ljj = __builtin_sqrt(dd);
        Lm[j][j] = ljj;
        invd[j] = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
          double v = A[i][j];
#pragma unroll
          for (int t = 0; t < j; ++t) v = __builtin_fma(-Lm[i][t], Lm[j][t], v);
          Lm[i][j] = v * invd[j];
        }
      }
    }
    if (!ok) break;
    if (done) {
      conv = true;
      break;
    }
    double zf[P], nb[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double v = (!MASKED || ((keep >> i) & 1u)) ? xtwz[i] : 0.0;
#pragma unroll
      for (int t = 0; t < i; ++t) v = __builtin_fma(-Lm[i][t], zf[t], v);
      zf[i] = v * invd[i];
    }
    bool capped = false;
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
      double v = zf[i];
#pragma unroll
      for (int t = i + 1; t < P; ++t) v = __builtin_fma(-Lm[t][i], nb[t], v);
      nb[i] = v * invd[i];
      capped = capped || !(__builtin_fabs(nb[i]) <= BETA_CAP);
    }
    if (capped) break;
#pragma unroll
    for (int i = 0; i < P; ++i) beta[i] = nb[i];
    dev_prev = dev;
  }
}

// LDS layout per block (one wave): design[c][row] for columns x1..x_{P-1} then y[row] (stride n_pad),
// followed by the 4-row staging area [slot][lane] of stage_group.
// RED: sparse-factor exclusion recounted from rows [0, N) at every grid point (batch.rs:1836-1853), reduced
// fits, and the separation fallback -- when attempt 0 does not converge, drop the included factor with the
// smallest minimum level count (code 2) and refit once from a cold start (:1976-2030).
// SCEN: general data path -- per-replicate correlation factor / marginals from the scenario design records and
// per-study effects (data_gen.rs:709-751, 990-1093); the fit still starts from the spec's effects.
template <int P, bool FACTORS, bool RED, bool SCEN, bool LOGIT>
__global__ __launch_bounds__(WAVE, MCP_GLM_WAVES) void glm_sim_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ float lds[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;  // grid = replicates of this launch
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t out_row = prm.out_base + lsim;
  const uint32_t max_n = prm.max_n;
  const uint32_t n_pad = (max_n + 3u) & ~3u;
  float* design = lds;                       // P columns x n_pad
  float* st = lds + (size_t)P * n_pad + lane;  // staging, stride WAVE

  // ---- generate the replicate into LDS: lane takes Philox blocks lane, lane + 64, ... ----
  RowCtx<P> cx;
  row_ctx_init<P, FACTORS>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  for (uint32_t ctr = lane; 4u * ctr < max_n; ctr += WAVE) {
    stage_group<P, WAVE, 1, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      if (rr < max_n) {
        float x[P], y;
        build_row<P, FACTORS, WAVE, 1, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, 0.0, &sl);
#pragma unroll
        for (int c = 1; c < P; ++c) design[(size_t)(c - 1) * n_pad + rr] = x[c];
        design[(size_t)(P - 1) * n_pad + rr] = y;
      }
    }
  }
  __syncthreads();  // single-wave block: orders the LDS writes above before the cross-lane reads below

  const uint32_t nt = prm.n_tests;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    double sum_y = 0.0;
    for (uint32_t i = lane; i < n; i += WAVE) sum_y += (double)design[(size_t)(P - 1) * n_pad + i];
    sum_y = wave_allreduce_sum(sum_y);
    const double nd = (double)n;
    const double pbar = sum_y / nd;
    double dev_null = (pbar > 0.0 && pbar < 1.0) ? -2.0 * (sum_y * log(pbar) + (nd - sum_y) * log(1.0 - pbar)) : 0.0;
    if (!LOGIT && prm.glm_family == 2u) {  // Poisson: the intercept-only fit has mu = mean(y)
      double dn = 0.0;
      for (uint32_t i = lane; i < n; i += WAVE) {
        const double yi = (double)design[(size_t)(P - 1) * n_pad + i];
        dn += 2.0 * (((yi > 0.0 && pbar > 0.0) ? yi * log(yi / pbar) : 0.0) - (yi - pbar));
      }
      dev_null = wave_allreduce_sum(dn);
    }

    double beta[P], Lm[P][P], invd[P];
    double dev;
    bool conv;
    uint32_t keep = (1u << P) - 1u, codes = 0;  // codes: 2 bits per factor (0 kept, 1 sparse, 2 separation)
    if (RED) {
      uint32_t fmin[MAX_FACTORS] = {0u, 0u, 0u, 0u};
      uint32_t flag_bits = 0;
      if (prm.min_level_count) {  // update_factor_exclusions (batch.rs:2451-2500), rows [0, n)
#pragma unroll
        for (int f = 0; f < MAX_FACTORS; ++f) {
          if ((uint32_t)f < prm.n_factors) {
            uint32_t mn = 0xffffffffu;
            for (uint32_t l = 0; l < prm.f_levels[f]; ++l) {
              uint32_t cnt = 0;
              for (uint32_t i = lane; i < n; i += WAVE) {
                uint32_t lv = 0;
#pragma unroll
                for (int c = 1; c < P; ++c)
                  if (prm.col_factor[c] == (uint32_t)f && design[(size_t)(c - 1) * n_pad + i] == 1.0f) lv = prm.col_level[c];
                cnt += lv == l ? 1u : 0u;
              }
#pragma unroll
              for (int m = 1; m < WAVE; m <<= 1) cnt += __shfl_xor(cnt, m, WAVE);
              mn = cnt < mn ? cnt : mn;
            }
            fmin[f] = mn;
            if (mn < prm.min_level_count) {
              flag_bits |= 1u << f;
              codes |= 1u << (2 * f);
            }
          }
        }
      }
      for (int attempt = 0; attempt < 2; ++attempt) {
        keep = prm.keep_mask[flag_bits & 15u];
#pragma unroll
        for (int c = 0; c < P; ++c) beta[c] = (attempt == 0 && ((keep >> c) & 1u)) ? prm.beta[c] : 0.0;
        irls_fit<P, true, LOGIT>(design, n_pad, n, lane, keep, prm.glm_family, beta, Lm, invd, dev, conv);
        if (conv || attempt == 1 || prm.min_level_count == 0u) break;
        int drop = -1;  // included_factor_with_smallest_min_count (batch.rs:2573-2600)
        uint32_t best = 0;
#pragma unroll
        for (int f = 0; f < MAX_FACTORS; ++f)
          if ((uint32_t)f < prm.n_factors && !((flag_bits >> f) & 1u) && (drop < 0 || fmin[f] < best)) {
            drop = f;
            best = fmin[f];
          }
        if (drop < 0) break;
        flag_bits |= 1u << drop;
        codes |= 2u << (2 * drop);
      }
      codes |= flag_bits << 16;
    } else {
      // ---- IRLS from the truth start (batch.rs:1945-1951) ----
#pragma unroll
      for (int c = 0; c < P; ++c) beta[c] = prm.beta[c];
      irls_fit<P, false, LOGIT>(design, n_pad, n, lane, keep, prm.glm_family, beta, Lm, invd, dev, conv);
    }

    // ---- Wald z^2 per target (cov = (X'WX)^-1), decisions, LRT overall ----
    const double* crow = prm.crit + (size_t)g * prm.crit_stride;
    double tsq[MAX_T_WAVE];
#pragma unroll
    for (int t = 0; t < MAX_T_WAVE; ++t) {
      tsq[t] = __builtin_nan("");
      if ((uint32_t)t < nt && conv && (!RED || ((keep >> prm.test_pos[t]) & 1u))) {
        const double* c = prm.contrast + t * MAX_P;
        double est = 0.0, nq = 0.0, u[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
          est = __builtin_fma(c[i], beta[i], est);
          double acc = c[i];
#pragma unroll
          for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lm[i][q], u[q], acc);
          u[i] = acc * invd[i];
          nq = __builtin_fma(u[i], u[i], nq);
        }
        if (nq > GLM_NEAR_ZERO && nq < __builtin_inf()) tsq[t] = est * est / nq;
      }
    }
    uint32_t unc_bits = 0, cor_bits = 0;
    decide_tests(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
    if (!conv) {
      unc_bits = 0;
      cor_bits = 0;
    }
    bool ov = false;
    if (prm.report_overall && conv && dev < __builtin_inf() && dev_null < __builtin_inf()) {
      double crit_overall = crow[3 + nt];
      if (RED)  // LRT df follows the fitted column count (batch.rs:2096-2113)
        crit_overall = prm.crit_red[(size_t)prm.red_row[(size_t)g * 16u + ((codes >> 16) & 15u)] * prm.crit_stride + 3 + nt];
      const double lrt = dev_null - dev;
      ov = (lrt < __builtin_inf()) && (lrt > crit_overall);
    }
    if (lane == 0) {  // the wave's replicate: one writer
      const size_t cell = (size_t)out_row * prm.n_grid + g;
#pragma unroll
      for (int t = 0; t < MAX_T_WAVE; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
      prm.conv[cell] = conv ? 1 : 0;
      if (prm.overall) prm.overall[cell] = ov ? 1 : 0;
      if (prm.stat && g == 0) {
#pragma unroll
        for (int t = 0; t < MAX_T_WAVE; ++t)
          if ((uint32_t)t < nt) prm.stat[(size_t)out_row * nt + t] = tsq[t];
      }
      if (RED && prm.factor_excluded)
        for (uint32_t f = 0; f < prm.n_factors; ++f)
          prm.factor_excluded[cell * prm.n_factors + f] = (uint8_t)((codes >> (2 * f)) & 3u);
    }
  }
}

// Fold the per-replicate flag tensors of one launch into the additive counters
// (aggregation.rs:84-284): lane per replicate, ballot + popcount, one atomic per wave per counter.
__global__ __launch_bounds__(256) void fold_flags_kernel(const SimParams* __restrict__ pp) {
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = local < prm.n_sims;
  const uint32_t row = prm.out_base + (active ? local : 0u);
  const uint32_t nt = prm.n_tests;
  const bool lane0 = (threadIdx.x & 63u) == 0u;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const size_t cell = (size_t)row * prm.n_grid + g;
    unsigned long long* ctr = prm.counters +
                            (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % CTR_COPIES) * prm.ctr_copy_stride +
                            (size_t)g * prm.ctr_stride;
    uint32_t ku = 0, kc = 0;
    for (uint32_t t = 0; t < nt; ++t) {
      const uint32_t u = active ? prm.unc[cell * nt + t] : 0u, c = active ? prm.cor[cell * nt + t] : 0u;
      ku += u != 0u;
      kc += c != 0u;
      const unsigned long long mu = __ballot(u != 0u), mc = __ballot(c != 0u);
      if (lane0) {
        if (mu) atomicAdd(ctr + prm.ctr_k_unc + t, (unsigned long long)__popcll(mu));
        if (mc) atomicAdd(ctr + prm.ctr_k_cor + t, (unsigned long long)__popcll(mc));
      }
    }
    // post-hoc contrasts count into their own slots and into the joint histogram (aggregation.rs:134-222)
    const uint32_t npc = prm.posthoc_unc ? prm.n_posthoc : 0u;
    for (uint32_t q = 0; q < npc; ++q) {
      const uint32_t u = active ? prm.posthoc_unc[cell * npc + q] : 0u, c = active ? prm.posthoc_cor[cell * npc + q] : 0u;
      ku += u != 0u;
      kc += c != 0u;
      const unsigned long long mu = __ballot(u != 0u), mc = __ballot(c != 0u);
      if (lane0) {
        if (mu) atomicAdd(ctr + prm.ctr_posthoc_unc + q, (unsigned long long)__popcll(mu));
        if (mc) atomicAdd(ctr + prm.ctr_posthoc_cor + q, (unsigned long long)__popcll(mc));
      }
    }
    for (uint32_t k = 0; k <= nt + npc; ++k) {
      const unsigned long long hu = __ballot(active && ku == k), hc = __ballot(active && kc == k);
      if (lane0) {
        if (hu) atomicAdd(ctr + prm.ctr_hist_unc + k, (unsigned long long)__popcll(hu));
        if (hc) atomicAdd(ctr + prm.ctr_hist_cor + k, (unsigned long long)__popcll(hc));
      }
    }
    if (prm.factor_excluded)  // codes: 1 = sparse level, 2 = dropped by the GLM separation fallback (aggregation.rs:244-262)
      for (uint32_t f = 0; f < prm.n_factors; ++f) {
        const uint32_t code = active ? prm.factor_excluded[cell * prm.n_factors + f] : 0u;
        const unsigned long long m1 = __ballot(code == 1u), m2 = __ballot(code == 2u);
        if (lane0) {
          if (m1) atomicAdd(ctr + prm.ctr_excl_sparse + f, (unsigned long long)__popcll(m1));
          if (m2) atomicAdd(ctr + prm.ctr_excl_sep + f, (unsigned long long)__popcll(m2));
        }
      }
    const unsigned long long mcv = __ballot(active && prm.conv[cell] != 0);
    const unsigned long long mov = __ballot(active && prm.overall && prm.overall[cell] != 0);
    const unsigned long long mac = __ballot(active);
    const unsigned long long mj = __ballot(active && prm.joint && prm.joint[cell] != 0);
    const unsigned long long mb1 = __ballot(active && prm.boundary && prm.boundary[cell] == 1);
    const unsigned long long mb2 = __ballot(active && prm.boundary && prm.boundary[cell] == 2);
    if (lane0) {
      if (mcv) atomicAdd(ctr + prm.ctr_conv, (unsigned long long)__popcll(mcv));
      if (mov) atomicAdd(ctr + prm.ctr_overall, (unsigned long long)__popcll(mov));
      if (mac) atomicAdd(ctr + prm.ctr_n_sims, (unsigned long long)__popcll(mac));
      if (mj) {
        atomicAdd(ctr + prm.ctr_joint_unc, (unsigned long long)__popcll(mj));
        atomicAdd(ctr + prm.ctr_joint_cor, (unsigned long long)__popcll(mj));
      }
      if (mb1) atomicAdd(ctr + prm.ctr_boundary_1, (unsigned long long)__popcll(mb1));
      if (mb2) atomicAdd(ctr + prm.ctr_boundary_2, (unsigned long long)__popcll(mb2));
    }
  }
}

// final[i] = sum over copies of counters[copy][i]
__global__ void reduce_counter_copies_kernel(const unsigned long long* __restrict__ copies, uint32_t copy_stride,
                                             uint32_t n, unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long s = 0;
  for (int c = 0; c < CTR_COPIES; ++c) s += copies[(size_t)c * copy_stride + i];
  out[i] = s;
}

hipError_t launch_reduce_counters(const unsigned long long* copies, uint32_t copy_stride, uint32_t n,
                                  unsigned long long* out, hipStream_t st) {
  hipLaunchKernelGGL(reduce_counter_copies_kernel, dim3((n + 127u) / 128u), dim3(128), 0, st, copies, copy_stride, n, out);
  return hipGetLastError();
}

size_t glm_lds_bytes(int P, uint32_t max_n) {  // staging sized for the general path's scratch column too
  const size_t n_pad = (max_n + 3u) & ~3u;
  return ((size_t)P * n_pad + (size_t)4 * (P + 1) * WAVE) * sizeof(float);
}

template <bool FACTORS, bool RED, bool SCEN, bool LOGIT>
static hipError_t launch_glm_f(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, hipStream_t st) {
  const size_t lds = glm_lds_bytes(P, max_n);
  const dim3 block(WAVE), grid(n_sims);
#define LG(PP)                                                                                                        \
  case PP: {                                                                                                          \
    if (lds > 48 * 1024) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)glm_sim_kernel<PP, FACTORS, RED, SCEN, LOGIT>,                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return e;                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((glm_sim_kernel<PP, FACTORS, RED, SCEN, LOGIT>), grid, block, lds, st, d_prm);                 \
  } break;
  switch (P) {
    LG(2) LG(3) LG(4) LG(5) LG(6) LG(7) LG(8)
    default: return hipErrorInvalidValue;
  }
#undef LG
  return hipGetLastError();
}

hipError_t launch_glm_sim(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, bool factors, bool reduced,
                          bool scen, bool logit, hipStream_t st) {
  if (!logit) {  // probit / Poisson: one run-time-family variant (the superset kernel)
    return scen ? launch_glm_f<true, true, true, false>(P, d_prm, n_sims, max_n, st)
                : launch_glm_f<true, true, false, false>(P, d_prm, n_sims, max_n, st);
  }
  if (scen) return launch_glm_f<true, true, true, true>(P, d_prm, n_sims, max_n, st);
  if (reduced) return launch_glm_f<true, true, false, true>(P, d_prm, n_sims, max_n, st);
  return factors ? launch_glm_f<true, false, false, true>(P, d_prm, n_sims, max_n, st)
                 : launch_glm_f<false, false, false, true>(P, d_prm, n_sims, max_n, st);
}

hipError_t launch_fold_flags(const SimParams* d_prm, uint32_t n_sims, hipStream_t st) {
  hipLaunchKernelGGL(fold_flags_kernel, dim3((n_sims + 255u) / 256u), dim3(256), 0, st, d_prm);
  return hipGetLastError();
}

}  // namespace mcp
