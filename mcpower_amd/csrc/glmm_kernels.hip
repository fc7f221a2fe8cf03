// glmm_kernels.hip -- gfx950 kernel of the clustered GLM: a GLMM with a random intercept, fitted by the Laplace
// approximation (the clustered branch of the Glm arm, crates/engine-core/src/batch.rs:1575-1830).
//
// The reference hands the fit to glmm 0.1.2 (`fit_on`, family Binomial{logit, probit} / Poisson{log}, nagq = 1,
// Wald SE mode "rx"), out of tree; this is lme4's glmer(nAGQ = 1) objective
//   d(theta, beta) = sum_i devres_i(mu_i(u~)) + |u~|^2 + sum_c log(theta^2 sum_{i in c} w_i + 1)
// with the conditional modes u~ from per-cluster Fisher scoring, minimised jointly over (theta, beta) by the bounded
// Nelder-Mead simplex -- the algorithm, constants and control flow of oracle/fit_glmm.c.
//
// ONE WAVEFRONT PER REPLICATE, design resident in LDS.  One evaluation has two phases of opposite shape:
//   A. lane = row:      eta0_i = x_i' beta                                   (n x p MACs, written to LDS)
//   B. lane = cluster:  the scalar Fisher-scoring loop for u_c over the cluster's rows (a random intercept decouples
//                       the clusters), its deviance term, weight sum and log-determinant term;
// the cluster terms are then summed in cluster order by every lane, so the objective -- and with it every branch of
// the simplex -- is wave-uniform.  The Wald covariance ("rx": (X'WX - sum_c theta^2 a_c a_c' / (theta^2 sw_c + 1))^-1,
// a_c = X_c' w_c) is formed once at the optimum.
#ifndef MCP_GLMM_WAVES
#define MCP_GLMM_WAVES 2  // waves per SIMD the register allocator leaves room for: 2 beats 1 by 1.35-1.5x on MI355X (profiles/r01_j_occupancy.txt)
#endif
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_scenario.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"
#include "mcp_glm_family.hpp"

#pragma clang fp contract(off)

namespace mcp {
namespace {

constexpr int QW = 64;
constexpr uint32_t GLMM_MAX_EVAL = 8000;
constexpr double GLMM_THETA_MAX = 32.0;
constexpr double GLMM_THETA_ZERO = 1e-6;
constexpr int GLMM_PIRLS_ITERS = 60;
constexpr double GLMM_NEAR_ZERO = 1e-30;
constexpr int GLMM_ROW = 10;  // simplex row stride (d = P + 1 <= 9)

typedef __attribute__((address_space(3))) double gdd;
typedef __attribute__((address_space(3))) float gdf;

struct GlmmLds {
  gdd *eta0, *tcl, *ucl, *swcl, *simplex, *fv, *tail;
  gdf *design, *cluster_u;
  float* st;
};

__device__ __forceinline__ void gq_sync() { __syncthreads(); }

__device__ __forceinline__ double gq_wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < QW; m <<= 1) v += __shfl_xor(v, m, QW);
  return v;
}

// Laplace deviance at par = (theta, beta); leaves the modes and weight sums of every cluster in LDS
template <int P>
__device__ __forceinline__ double glmm_eval(const SimParams& prm, const GlmmLds m, const gdd* par, uint32_t lane, uint32_t n,
                                         uint32_t ncl, uint32_t n_pad, uint32_t keep) {
  const uint32_t fam = prm.glm_family;
  const double theta = par[0];
  // par = (theta, coefficients of the fitted columns in column order); a column outside a test_formula subset has
  // coefficient 0 (batch.rs:1588-1640 fits the gathered kept columns: the same sums plus exact zeros)
  double beta[P];
  {
    uint32_t k = 1;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const bool kept = (keep >> c) & 1u;
      beta[c] = kept ? par[k] : 0.0;
      k += kept ? 1u : 0u;
    }
  }
  for (uint32_t i = lane; i < n; i += QW) {
    double e = beta[0];
#pragma unroll
    for (int c = 1; c < P; ++c) e += (double)m.design[(size_t)(c - 1) * n_pad + i] * beta[c];
    m.eta0[i] = e;
  }
  gq_sync();
  const uint32_t cv = prm.cluster_v;
  // few clusters leave lanes idle: 2^ls lanes share a cluster, each takes every 2^ls-th row, and an xor butterfly
  // leaves the same sums (bit for bit: a + b == b + a) in all of them, so the group iterates in lockstep
  const uint32_t ls = ncl <= 8u ? 3u : (ncl <= 16u ? 2u : (ncl <= 32u ? 1u : 0u));
  const uint32_t sub = lane & ((1u << ls) - 1u);
  for (uint32_t c = lane >> ls; c < ncl; c += QW >> ls) {
    const uint32_t first = prm.cluster_fixed_clusters ? c : c * cv;
    const uint32_t step = prm.cluster_fixed_clusters ? cv : 1u;
    const uint32_t last = prm.cluster_fixed_clusters ? n : ((c + 1u) * cv < n ? (c + 1u) * cv : n);
    // conditional mode by Fisher scoring with step halving (lme4's PIRLS; oracle/fit_glmm.c): a trial point whose
    // penalised deviance is not below the accepted point's is pulled half-way back to it -- an undamped step 2-cycles
    // on clusters far from their prior (15 events of 20 at a 10 % baseline).  (u, sw, dev) = the accepted point
    double u = 0.0, sw = 0.0, dev = 0.0, ut = 0.0, pd_acc = __builtin_inf();
    if (first < last) {
      for (int it = 0; it < GLMM_PIRLS_ITERS; ++it) {
        double swz = 0.0, swt = 0.0, devt = 0.0;
        for (uint32_t i = first + sub * step; i < last; i += step << ls) {
          const double e0 = m.eta0[i];
          double w, wz, d;
          fam_eval(fam, e0 + theta * ut, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, d);
          swt += w;
          swz += wz - w * e0;  // w (z - eta0)
          devt += d;
        }
        for (uint32_t o = 1u; o < (1u << ls); o <<= 1) {
          swt += __shfl_xor(swt, (int)o);
          swz += __shfl_xor(swz, (int)o);
          devt += __shfl_xor(devt, (int)o);
        }
        const double pd = devt + ut * ut, tol = 1e-13 * (__builtin_fabs(pd) + 1.0);
        if (!(pd <= pd_acc + tol)) {  // worse than the accepted point (or not finite): halve the step
          const double half = 0.5 * (ut + u);
          if (half == ut || half == u) break;
          ut = half;
          continue;
        }
        const bool conv = __builtin_fabs(pd_acc - pd) <= tol;  // pd_acc = inf on the first pass: never
        u = ut;
        sw = swt;
        dev = devt;
        pd_acc = pd;
        if (conv) break;
        const double un = theta * swz / (theta * theta * swt + 1.0);
        if (un == u) break;
        ut = un;
      }
      if (!(pd_acc < __builtin_inf())) dev = __builtin_inf();  // not even the prior mode evaluates
      double term = dev + u * u + log(theta * theta * sw + 1.0);
      if (prm.glmm_nagq > 1u) {
        // adaptive Gauss-Hermite (lme4 glmerAGQ): the k-point rule centred at the mode, scaled by the conditional sd,
        // as a ratio to its Laplace term; every lane of the group sweeps its rows once per off-centre node
        const double sd = 1.0 / __builtin_sqrt(theta * theta * sw + 1.0), d0 = dev + u * u;
        double mult = 0.0;
        for (uint32_t j = 0; j < prm.glmm_nagq; ++j) {
          const double zj = prm.glmm_gz[j], wj = prm.glmm_gw[j];
          if (2u * j + 1u == prm.glmm_nagq) {
            mult += wj;
            continue;
          }
          const double v = u + sd * zj;
          double ds = 0.0;
          for (uint32_t i = first + sub * step; i < last; i += step << ls) {
            double w, wz, d;
            fam_eval(fam, m.eta0[i] + theta * v, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, d);
            ds += d;
          }
          for (uint32_t o = 1u; o < (1u << ls); o <<= 1) ds += __shfl_xor(ds, (int)o);
          mult += wj * exp(-0.5 * (ds + v * v - d0) + 0.5 * zj * zj);
        }
        term -= 2.0 * log(mult);
      }
      if (sub == 0u) m.tcl[c] = term;
    } else if (sub == 0u) {
      m.tcl[c] = 0.0;
    }
    if (sub == 0u) {
      m.ucl[c] = u;
      m.swcl[c] = sw;
    }
  }
  gq_sync();
  double total = 0.0;
  for (uint32_t c = 0; c < ncl; ++c) total += m.tcl[c];
  gq_sync();
  return (total < __builtin_inf() && total > -__builtin_inf()) ? total : __builtin_inf();
}

// SCEN: general data path (scenario design records, non-normal marginals, per-study effects) plus the LME knobs --
// random-effect shape and per-replicate tau^2 (data_gen.rs:146-161, 1069-1076)
#include "mcp_stationary.hpp"  // MCP_STAT_H / MCP_STAT_TOL
#include "mcp_fit_fsm.hpp"
#include "mcp_glmm_tail.hpp"

template <int P, bool SCEN>
__global__ __launch_bounds__(QW, MCP_GLMM_WAVES) void glmm_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ double lds_q[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = replicate_key(prm, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t max_n = prm.max_n, n_pad = (max_n + 3u) & ~3u;
  const uint32_t ncl_max = prm.n_clusters_max, ncl_pad = (ncl_max + 1u) & ~1u;
  const uint32_t cv = prm.cluster_v;
  constexpr uint32_t d = P + 1;
  GlmmLds m;
  {
    gdd* q = (gdd*)lds_q;
    m.eta0 = q; q += n_pad;
    m.tcl = q; q += ncl_pad;
    m.ucl = q; q += ncl_pad;
    m.swcl = q; q += ncl_pad;
    m.simplex = q; q += (size_t)(d + 4u) * GLMM_ROW;
    m.fv = q; q += d + 1u + (d & 1u ? 0u : 1u);
    m.tail = q; q += 8 + 64 + 64 + 64 + 8 + 8;
    gdf* f = (gdf*)q;
    m.design = f; f += (size_t)P * n_pad;
    m.cluster_u = f; f += ncl_pad;
    m.st = (float*)f + lane;
  }
  // ---- cluster random intercepts (data_gen.rs:146-161), then the rows: the effect sits inside the link ----
  double tau_rep = prm.tau;
  if (SCEN && prm.scen_tau_sq) {
    const double t2 = prm.scen_tau_sq[lsim];
    tau_rep = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
  }
  for (uint32_t c = lane; c < ncl_max; c += QW) {
    if (SCEN) {
      m.cluster_u[c] = (float)(draw_cluster_effect(prm, k0, k1, c) * tau_rep);
    } else {
      const float z = norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, (uint64_t)c)));
      m.cluster_u[c] = (float)((double)z * prm.tau);
    }
  }
  gq_sync();
  RowCtx<P> cx;
  row_ctx_init<P, true>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  const uint32_t bmask = prm.between_mask;
  for (uint32_t ctr = lane; 4u * ctr < max_n; ctr += QW) {
    stage_group<P, QW, 1, SCEN>(prm, k0, k1, ctr, m.st, SCEN ? sl.meta : 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      if (rr < max_n) {
        const uint32_t cid = prm.cluster_fixed_clusters ? rr % cv : rr / cv;
        float x[P], y;
        const float e = build_x<P, true, QW, SCEN>(prm, cx, k0, k1, m.st, r, rr, x, &sl);
        if (bmask == 0u) finish_row<P, 1, SCEN>(prm, cx, x, e, y, (double)m.cluster_u[cid < ncl_max ? cid : 0u], &sl);
        else y = e;  // the outcome draw waits until the cluster-level columns are broadcast
#pragma unroll
        for (int c = 1; c < P; ++c) m.design[(size_t)(c - 1) * n_pad + rr] = x[c];
        m.design[(size_t)(P - 1) * n_pad + rr] = y;
      }
    }
  }
  gq_sync();
  if (bmask != 0u) {
    // cluster-level predictors (data_gen.rs:606-636), as in lmm_kernels.hip: broadcast from the representative row,
    // rebuild interactions and the outcome, rewrite the marked columns last
    for (uint32_t rr = lane; rr < max_n; rr += QW) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cv : rr - (rr % cv);
      float x[P];
      x[0] = 1.0f;
#pragma unroll
      for (int c = 1; c < P; ++c) {
        const uint32_t src = ((bmask >> c) & 1u) ? rep : rr;
        x[c] = m.design[(size_t)(c - 1) * n_pad + src];
      }
      rebuild_interactions<P>(cx, x);
      const float e = m.design[(size_t)(P - 1) * n_pad + rr];
      const uint32_t cid = prm.cluster_fixed_clusters ? rr % cv : rr / cv;
      float y;
      finish_row<P, 1, SCEN>(prm, cx, x, e, y, (double)m.cluster_u[cid < ncl_max ? cid : 0u], &sl);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 1; c < P; ++c)
        if (((bmask >> c) & 1u) == 0u && ((cx.inter_any >> c) & 1u)) m.design[(size_t)(c - 1) * n_pad + rr] = x[c];
      m.design[(size_t)(P - 1) * n_pad + rr] = y;
    }
    gq_sync();
    for (uint32_t rr = lane; rr < max_n; rr += QW) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cv : rr - (rr % cv);
#pragma unroll
      for (int c = 1; c < P; ++c)
        if ((bmask >> c) & 1u) {
          const float v = m.design[(size_t)(c - 1) * n_pad + rep];
          __builtin_amdgcn_wave_barrier();
          m.design[(size_t)(c - 1) * n_pad + rr] = v;
        }
    }
    gq_sync();
  }

  bool has_prev = false;
  double carry = 0.0;
  const uint32_t nt = prm.n_tests;
  // fitted columns: all of them, or the test_formula subset (no factor exclusion on clustered specs, batch.rs:1583)
  const uint32_t keep = prm.glmm_keep;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    const uint32_t ncl = prm.cluster_fixed_clusters ? cv : (n + cv - 1u) / cv;
    gdd* x0 = m.simplex;
    if (lane == 0u) {
      double v = has_prev ? carry : (prm.truth_start ? prm.tau : 1.0);  // carried theta-hat / truth / blind THETA0
      if ((has_prev || prm.truth_start) && v < 0.01) v = 0.01;
      x0[0] = v;
      uint32_t k = 1;
      for (uint32_t c = 0; c < (uint32_t)P; ++c)
        if ((keep >> c) & 1u) x0[k++] = prm.beta[c];                      // beta starts at the truth, unconditionally
    }
    gq_sync();
    // ---- the fit: one wave-uniform state machine around the single inlined evaluation (mcp_fit_fsm.hpp) ----
    const uint32_t d_fit = 1u + (uint32_t)__popc(keep);
    bool ok = false;
    double theta = 0.0;
    auto eval_pt = [&](gdd* x) -> double {  // clamp the point onto the bounds, then the Laplace deviance
      if (lane < d_fit) {
        double v = x[lane];
        if (lane == 0u) v = v < 0.0 ? 0.0 : (v > GLMM_THETA_MAX ? GLMM_THETA_MAX : v);
        else v = v < -50.0 ? -50.0 : (v > 50.0 ? 50.0 : v);
        x[lane] = v;
      }
      gq_sync();
      return glmm_eval<P>(prm, m, x, lane, n, ncl, n_pad, keep);
    };
    auto on_final = [&](double cur, uint32_t n_eval) -> uint32_t {
      theta = x0[0];
      ok = cur < __builtin_inf() && n_eval < GLMM_MAX_EVAL;
      // ---- RX'RX = X'WX - sum_c theta^2 a_c a_c' / (theta^2 sw_c + 1), a_c = X_c'w_c, at the optimum, in LDS (mcp_glmm_tail.hpp):
      //      lane = row leaves the IRLS weight at the final modes in eta0's slot (every evaluation rebuilds eta0), then
      //      lane = entry (r, s) of the lower triangle sums its products cluster by cluster, rows ascending ----
      for (uint32_t i = lane; i < n; i += QW) {
        const uint32_t cid = prm.cluster_fixed_clusters ? i % cv : i / cv;
        double w, wz, dv;
        fam_eval(prm.glm_family, m.eta0[i] + theta * m.ucl[cid], (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
        m.eta0[i] = w;
      }
      gq_sync();
      if (lane < (uint32_t)(P * (P + 1) / 2)) {
        uint32_t r = 0;
        while ((r + 1u) * (r + 2u) / 2u <= lane) ++r;
        const uint32_t s = lane - r * (r + 1u) / 2u;
        double acc = 0.0;
        for (uint32_t c = 0; c < ncl; ++c) {
          const uint32_t first = prm.cluster_fixed_clusters ? c : c * cv;
          const uint32_t step = prm.cluster_fixed_clusters ? cv : 1u;
          const uint32_t last = prm.cluster_fixed_clusters ? n : ((c + 1u) * cv < n ? (c + 1u) * cv : n);
          double loc = 0.0, ar = 0.0, as = 0.0;
          for (uint32_t i = first; i < last; i += step) {
            const double w = m.eta0[i];
            const double xr = r == 0u ? 1.0 : (double)m.design[(size_t)(r - 1u) * n_pad + i];
            const double xs = s == 0u ? 1.0 : (double)m.design[(size_t)(s - 1u) * n_pad + i];
            ar += w * xr;
            as += w * xs;
            loc += w * xr * xs;
          }
          acc += loc - theta * theta * ar * as / (theta * theta * m.swcl[c] + 1.0);
        }
        m.tail[GT_LX + r * 8u + s] = acc;
      }
      if (lane == 0u) {  // fitted coefficients in column order; a column outside a test_formula subset has coefficient 0
        uint32_t k = 1;
        for (uint32_t c = 0; c < (uint32_t)P; ++c) {
          const bool kept = (keep >> c) & 1u;
          m.tail[GT_BETA + c] = kept ? x0[k] : 0.0;
          k += kept ? 1u : 0u;
        }
      }
      gq_sync();
      if (lane == 0u) m.tail[GT_SPARE] = glmm_tail_factor(m.tail + GT_LX, m.tail + GT_LX, (uint32_t)P, keep) ? 1.0 : 0.0;
      gq_sync();
      ok = (m.tail[GT_SPARE] != 0.0) && ok;
      // Wald SE mode "hessian": lme4's deriv12 stencil over (theta, beta); a singular fit, or a block that does not
      // factor, keeps "rx"
      return (prm.glmm_hessian && ok && !(theta < GLMM_THETA_ZERO)) ? (FIT_FSM_HESSIAN | 1u) : 0u;
    };
    const FitFsmOut fit = simplex_fit_fsm(m.simplex, m.fv, d_fit, (uint32_t)GLMM_ROW, 1u, 1u, GLMM_MAX_EVAL, lane, eval_pt, on_final);
    if (fit.hess_ok && lane == 0u) glmm_tail_take_hessian(m.tail, m.simplex + GLMM_ROW, (uint32_t)GLMM_ROW, 1u, (uint32_t)P, keep);
    bool conv = ok;
    const bool at_upper = theta > GLMM_THETA_MAX - 1e-3;
    const uint32_t pinned = (conv && theta < GLMM_THETA_ZERO) ? 1u : 0u;
    const uint32_t bh = !conv ? 0u : (at_upper ? 2u : pinned);
    if (at_upper) conv = false;
    if (conv) {
      carry = theta;
      has_prev = true;
    }
    // ---- Wald z^2 (dispersion 1), joint Wald chi2 over the targets vs chi2_ppf(1 - alpha, k), decisions: one lane, in LDS ----
    if (lane == 0u)
      glmm_tail_decide(prm, m.tail, (uint32_t)P, nt, conv, bh, pinned, theta * theta, prm.crit + (size_t)g * prm.crit_stride,
                       (size_t)(prm.out_base + lsim) * prm.n_grid + g, (size_t)(prm.out_base + lsim), g == 0u);
    gq_sync();
  }
}

}  // namespace

size_t glmm_lds_bytes(int P, uint32_t max_n, uint32_t ncl_max) {
  const size_t n_pad = ((size_t)max_n + 3u) & ~(size_t)3u;
  const size_t ncl_pad = ((size_t)ncl_max + 1u) & ~(size_t)1u;
  const size_t d = (size_t)P + 1;
  const size_t dbl = n_pad + 3 * ncl_pad + (d + 4) * GLMM_ROW + d + 2 + (8 + 64 + 64 + 64 + 8 + 8);
  const size_t flt = (size_t)P * n_pad + ncl_pad + (size_t)4 * (P + 1) * QW;
  return dbl * 8 + flt * 4 + 64;
}

hipError_t launch_glmm(int P, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool scen,
                       hipStream_t st) {
  const size_t lds = glmm_lds_bytes(P, max_n, ncl_max);
  const dim3 block(QW), grid(n_sims);
#define GQ(PP)                                                                                                         \
  case PP: {                                                                                                           \
    if (lds > 48 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute(scen ? (const void*)glmm_kernel<PP, true> : (const void*)glmm_kernel<PP, false>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
    }                                                                                                                  \
    if (scen) hipLaunchKernelGGL((glmm_kernel<PP, true>), grid, block, lds, st, d_prm);                                \
    else hipLaunchKernelGGL((glmm_kernel<PP, false>), grid, block, lds, st, d_prm);                                    \
  } break;
  switch (P) {
    GQ(2) GQ(3) GQ(4) GQ(5) GQ(6) GQ(7) GQ(8)
    default: return hipErrorInvalidValue;
  }
#undef GQ
  return hipGetLastError();
}

}  // namespace mcp
