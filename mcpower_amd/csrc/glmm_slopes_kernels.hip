// glmm_slopes_kernels.hip -- gfx950 kernel of the clustered GLM with RANDOM SLOPES on the primary grouping (Q = 1 + slopes
// <= 3 random effects per cluster): the vector-mode generalisation of glmm_kernels.hip (same sources, same objective),
// following oracle/fit_glmm_slopes.c.  Lane = cluster carries the Q-vector mode, the Q x Q penalised weight matrix
// I + A'WA (A = Z_c Lambda) and its Cholesky factor in registers.
// (derived from glmm_kernels.hip: the clustered branch of the Glm arm, crates/engine-core/src/batch.rs:1575-1830)
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
constexpr uint32_t GLMM_MAX_EVAL = 12000;
constexpr double GLMM_THETA_MAX = 32.0;
constexpr double GLMM_THETA_ZERO = 1e-6;
constexpr int GLMM_PIRLS_ITERS = 60;
constexpr double GLMM_NEAR_ZERO = 1e-30;
constexpr int GLMM_ROW = 16;  // simplex row stride (d = Q (Q + 1) / 2 + P <= 14)

typedef __attribute__((address_space(3))) double gdd;
typedef __attribute__((address_space(3))) float gdf;

struct GlmmLds {
  gdd *eta0, *tcl, *ucl, *lmcl, *simplex, *fv, *tail, *carry;
  gdf *design, *cluster_u, *slope_u;
  float* st;
};

__device__ __forceinline__ void gq_sync() { __syncthreads(); }

__device__ __forceinline__ double gq_wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < QW; m <<= 1) v += __shfl_xor(v, m, QW);
  return v;
}

// Laplace deviance at par = (theta vech, beta); leaves the modes and the Cholesky factors of I + A'WA in LDS
template <int P, int Q>
__device__ __forceinline__ double glmm_eval(const SimParams& prm, const GlmmLds m, const gdd* par, uint32_t lane, uint32_t n,
                                         uint32_t ncl, uint32_t n_pad, uint32_t keep) {
  constexpr int NTH = Q * (Q + 1) / 2;
  const uint32_t fam = prm.glm_family;
  double T[Q][Q];
  {
    int k = 0;
#pragma unroll
    for (int c = 0; c < Q; ++c)
#pragma unroll
      for (int r = 0; r < Q; ++r) {
        T[r][c] = r >= c ? par[k] : 0.0;
        k += r >= c ? 1 : 0;
      }
  }
  double beta[P];
  {
    uint32_t k = NTH;
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
  bool ok = true;
  // 2^ls lanes share a cluster (see glmm_kernels.hip): every 2^ls-th row each, xor butterfly over the partial sums
  const uint32_t ls = ncl <= 8u ? 3u : (ncl <= 16u ? 2u : (ncl <= 32u ? 1u : 0u));
  const uint32_t sub = lane & ((1u << ls) - 1u);
  for (uint32_t c = lane >> ls; c < ncl; c += QW >> ls) {
    const uint32_t first = prm.cluster_fixed_clusters ? c : c * cv;
    const uint32_t step = prm.cluster_fixed_clusters ? cv : 1u;
    const uint32_t last = prm.cluster_fixed_clusters ? n : ((c + 1u) * cv < n ? (c + 1u) * cv : n);
    double u[Q], Lm[Q][Q];
#pragma unroll
    for (int b = 0; b < Q; ++b) {
      u[b] = 0.0;
#pragma unroll
      for (int dd = 0; dd < Q; ++dd) Lm[b][dd] = b == dd ? 1.0 : 0.0;
    }
    // Fisher scoring with step halving (oracle/fit_glmm_slopes.c): (u, Lm, dev, logdet) belong to the ACCEPTED point,
    // ut is the trial; a trial whose penalised deviance is not below the accepted one's is pulled half-way back
    double pd_acc = __builtin_inf(), dev = 0.0, logdet = 0.0, ut[Q];
#pragma unroll
    for (int b = 0; b < Q; ++b) ut[b] = 0.0;
    if (first < last) {
      for (int it = 0; it < GLMM_PIRLS_ITERS; ++it) {
        double M[Q][Q], r[Q], devt = 0.0;
#pragma unroll
        for (int b = 0; b < Q; ++b) {
          r[b] = 0.0;
#pragma unroll
          for (int dd = 0; dd < Q; ++dd) M[b][dd] = 0.0;
        }
        for (uint32_t i = first + sub * step; i < last; i += step << ls) {
          double z[Q], a[Q], au = 0.0;
          z[0] = 1.0;
#pragma unroll
          for (int dd = 1; dd < Q; ++dd) z[dd] = (double)m.design[(size_t)(prm.lmg_cols[0][dd - 1] - 1u) * n_pad + i];
#pragma unroll
          for (int b = 0; b < Q; ++b) {  // a = Lambda' z
            double sacc = 0.0;
#pragma unroll
            for (int dd = 0; dd < Q; ++dd)
              if (dd >= b) sacc += T[dd][b] * z[dd];
            a[b] = sacc;
            au += sacc * ut[b];
          }
          const double e0 = m.eta0[i];
          double w, wz, dv;
          fam_eval(fam, e0 + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
          devt += dv;
          const double res = wz - w * e0;  // w (z - eta0)
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            r[b] += res * a[b];
#pragma unroll
            for (int dd = 0; dd < Q; ++dd)
              if (dd <= b) M[b][dd] += w * a[b] * a[dd];
          }
        }
        for (uint32_t o = 1u; o < (1u << ls); o <<= 1) {
          devt += __shfl_xor(devt, (int)o);
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            r[b] += __shfl_xor(r[b], (int)o);
#pragma unroll
            for (int dd = 0; dd < Q; ++dd)
              if (dd <= b) M[b][dd] += __shfl_xor(M[b][dd], (int)o);
          }
        }
        double uu = 0.0;
#pragma unroll
        for (int b = 0; b < Q; ++b) {
          M[b][b] += 1.0;
          uu += ut[b] * ut[b];
        }
        const double pd = devt + uu, tol = 1e-13 * (__builtin_fabs(pd) + 1.0);
        if (!(pd <= pd_acc + tol)) {  // worse than the accepted point (or not finite): halve the step
          bool stuck_t = true, stuck_a = true;
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            const double half = 0.5 * (ut[b] + u[b]);
            stuck_t = stuck_t && (half == ut[b]);
            stuck_a = stuck_a && (half == u[b]);
            ut[b] = half;
          }
          if (stuck_t || stuck_a) break;
          continue;
        }
        bool cok = true;  // lower Cholesky of I + A'WA at the point about to be accepted
#pragma unroll
        for (int j = 0; j < Q; ++j) {
          double dj = M[j][j];
#pragma unroll
          for (int k = 0; k < Q; ++k)
            if (k < j) dj -= Lm[j][k] * Lm[j][k];
          cok = cok && (dj > 0.0) && (dj < __builtin_inf());
          Lm[j][j] = __builtin_sqrt(dj);
#pragma unroll
          for (int i2 = 0; i2 < Q; ++i2)
            if (i2 > j) {
              double v = M[i2][j];
#pragma unroll
              for (int k = 0; k < Q; ++k)
                if (k < j) v -= Lm[i2][k] * Lm[j][k];
              Lm[i2][j] = v / Lm[j][j];
            }
        }
        if (!cok) { ok = false; break; }
        logdet = 0.0;
#pragma unroll
        for (int b = 0; b < Q; ++b) logdet += 2.0 * log(Lm[b][b]);
        const bool conv = __builtin_fabs(pd_acc - pd) <= tol;  // pd_acc = inf on the first pass: never
#pragma unroll
        for (int b = 0; b < Q; ++b) u[b] = ut[b];
        dev = devt;
        pd_acc = pd;
        if (conv) break;
        double un[Q], t[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) {
          double v = r[b];
#pragma unroll
          for (int k = 0; k < Q; ++k)
            if (k < b) v -= Lm[b][k] * t[k];
          t[b] = v / Lm[b][b];
        }
#pragma unroll
        for (int b = Q - 1; b >= 0; --b) {
          double v = t[b];
#pragma unroll
          for (int k = 0; k < Q; ++k)
            if (k > b) v -= Lm[k][b] * un[k];
          un[b] = v / Lm[b][b];
        }
        bool same = true;
#pragma unroll
        for (int b = 0; b < Q; ++b) same = same && (un[b] == u[b]);
        if (same) break;
#pragma unroll
        for (int b = 0; b < Q; ++b) ut[b] = un[b];
      }
      if (!(pd_acc < __builtin_inf())) ok = false;  // not even the prior mode evaluates
      double uu = 0.0;
#pragma unroll
      for (int b = 0; b < Q; ++b) uu += u[b] * u[b];
      double term = dev + uu + logdet;
      if (prm.glmm_nagq > 1u) {
        // adaptive Gauss-Hermite quadrature over the Q random effects of the cluster (validate.rs:918-941 admits it for one
        // grouping with <= 3 of them): the k^Q tensor rule centred at the mode, scaled by the factor of the curvature
        // there, u_j = u~ + Lm^-T z_j, as a ratio to the Laplace term -- oracle/fit_glmm_slopes.c, node index 0 fastest.
        // Every lane of the group sweeps its rows once per off-centre node.
        const uint32_t kq = prm.glmm_nagq;
        const double d0 = dev + uu;
        uint32_t nn = 1u;
#pragma unroll
        for (int b = 0; b < Q; ++b) nn *= kq;
        double mult = 0.0;
        for (uint32_t node = 0; node < nn; ++node) {
          double zn[Q], xs[Q], v[Q], wj = 1.0, zz = 0.0;
          uint32_t rem = node;
          bool centre = true;
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            const uint32_t ib = rem % kq;
            rem /= kq;
            zn[b] = prm.glmm_gz[ib];
            wj *= prm.glmm_gw[ib];
            zz += zn[b] * zn[b];
            centre = centre && (2u * ib + 1u == kq);
          }
          if (centre) {
            mult += wj;
            continue;
          }
#pragma unroll
          for (int b = Q - 1; b >= 0; --b) {  // Lm' xs = z
            double t = zn[b];
#pragma unroll
            for (int k2 = 0; k2 < Q; ++k2)
              if (k2 > b) t -= Lm[k2][b] * xs[k2];
            xs[b] = t / Lm[b][b];
          }
          double vv = 0.0;
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            v[b] = u[b] + xs[b];
            vv += v[b] * v[b];
          }
          double ds = 0.0;
          for (uint32_t i = first + sub * step; i < last; i += step << ls) {
            double z[Q], au = 0.0;
            z[0] = 1.0;
#pragma unroll
            for (int dd = 1; dd < Q; ++dd) z[dd] = (double)m.design[(size_t)(prm.lmg_cols[0][dd - 1] - 1u) * n_pad + i];
#pragma unroll
            for (int b = 0; b < Q; ++b) {
              double sacc = 0.0;
#pragma unroll
              for (int dd = 0; dd < Q; ++dd)
                if (dd >= b) sacc += T[dd][b] * z[dd];
              au += sacc * v[b];
            }
            double w, wz, dv;
            fam_eval(fam, m.eta0[i] + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
            ds += dv;
          }
          for (uint32_t o = 1u; o < (1u << ls); o <<= 1) ds += __shfl_xor(ds, (int)o);
          mult += wj * exp(-0.5 * (ds + vv - d0) + 0.5 * zz);
        }
        term -= 2.0 * log(mult);
      }
      if (sub == 0u) m.tcl[c] = term;
    } else if (sub == 0u) {
      m.tcl[c] = 0.0;
    }
    if (sub == 0u) {
#pragma unroll
      for (int b = 0; b < Q; ++b) {
        m.ucl[(size_t)c * Q + b] = u[b];
#pragma unroll
        for (int dd = 0; dd < Q; ++dd) m.lmcl[((size_t)c * Q + b) * Q + dd] = Lm[b][dd];
      }
    }
  }
  ok = __ballot(!ok) == 0ull;
  gq_sync();
  double total = 0.0;
  for (uint32_t c = 0; c < ncl; ++c) total += m.tcl[c];
  gq_sync();
  return (ok && total < __builtin_inf() && total > -__builtin_inf()) ? total : __builtin_inf();
}

// theta entry k of the column-major lower vech of a Q x Q template is a diagonal iff it starts a column
template <int Q>
__device__ __forceinline__ bool glmm_is_diag(uint32_t k) {
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < Q; ++c) {
    if (k == o) return true;
    o += (uint32_t)(Q - c);
  }
  return false;
}

// SCEN: general data path (scenario design records, non-normal marginals, per-study effects) plus the LME knobs --
// random-effect shape and per-replicate tau^2 (data_gen.rs:146-161, 1069-1076)
#include "mcp_stationary.hpp"  // MCP_STAT_H / MCP_STAT_TOL
#include "mcp_fit_fsm.hpp"
#include "mcp_glmm_tail.hpp"

template <int P, int Q, bool SCEN>
__global__ __launch_bounds__(QW, MCP_GLMM_WAVES) void glmm_slopes_kernel(const SimParams* __restrict__ pp) {
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
  constexpr uint32_t NTH = Q * (Q + 1) / 2;
  constexpr uint32_t d = NTH + P;
  GlmmLds m;
  {
    gdd* q = (gdd*)lds_q;
    m.eta0 = q; q += n_pad;
    m.tcl = q; q += ncl_pad;
    m.ucl = q; q += (size_t)ncl_pad * Q;
    m.lmcl = q; q += (size_t)ncl_pad * Q * Q;
    m.simplex = q; q += (size_t)(d + 4u) * GLMM_ROW;
    m.fv = q; q += d + 1u + (d & 1u ? 0u : 1u);
    m.tail = q; q += 8 + 64 + 64 + 64 + 8 + 8;
    gdf* f = (gdf*)q;
    m.design = f; f += (size_t)P * n_pad;
    m.cluster_u = f; f += ncl_pad;
    m.slope_u = f; f += (size_t)ncl_pad * (Q - 1);
    m.st = (float*)f + lane;
  }
  // ---- cluster random intercepts, then the correlated slopes through chol(diag(tau) R diag(tau)) on the sequential
  //      stream (data_gen.rs:146-284; same order as lmm_general_impl.hpp), then the rows: the effect sits inside the link
  auto re_draw = [&](uint32_t ordinal) -> double {
    if (SCEN) return draw_cluster_effect(prm, k0, k1, ordinal);
    return (double)norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, (uint64_t)ordinal)));
  };
  double tau_rep = prm.lmg_tau[0];
  if (SCEN && prm.scen_tau_sq) {
    const double t2 = prm.scen_tau_sq[lsim];
    tau_rep = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
  }
  for (uint32_t c = lane; c < ncl_max; c += QW) m.cluster_u[c] = (float)(re_draw(prm.lmg_seq[0] + c) * tau_rep);
  gq_sync();
  {
    double l0[9];
    if (!SCEN) {
      for (uint32_t i = 0; i < 9u; ++i) l0[i] = prm.lmg_chol[0][i];
    } else {  // the replicate's own tau^2 (ICC jitter) enters D before the factorisation (data_gen.rs:55-75, 1069-1076)
      const double tv[3] = {tau_rep, prm.lmg_slope_sd[0][0], prm.lmg_slope_sd[0][1]};
      double dm[9];
      for (uint32_t i = 0; i < 9u; ++i) l0[i] = 0.0;
      for (uint32_t i = 0; i < (uint32_t)Q; ++i)
        for (uint32_t j = 0; j < (uint32_t)Q; ++j) dm[i * Q + j] = tv[i] * prm.lmg_recorr[0][i * Q + j] * tv[j];
      for (uint32_t j = 0; j < (uint32_t)Q; ++j) {
        double diag = dm[j * Q + j];
        for (uint32_t k = 0; k < j; ++k) diag -= l0[j * Q + k] * l0[j * Q + k];
        const double ljj = __builtin_sqrt(diag > 0.0 ? diag : 0.0);
        l0[j * Q + j] = ljj;
        for (uint32_t i = j + 1u; i < (uint32_t)Q; ++i)
          if (ljj > 0.0) {
            double sacc = dm[i * Q + j];
            for (uint32_t k = 0; k < j; ++k) sacc -= l0[i * Q + k] * l0[j * Q + k];
            l0[i * Q + j] = sacc / ljj;
          }
      }
    }
    for (uint32_t c = lane; c < ncl_max; c += QW) {
      double z[3];
      z[0] = tau_rep > 0.0 ? (double)m.cluster_u[c] / tau_rep : 0.0;
#pragma unroll
      for (int j = 1; j < Q; ++j) z[j] = re_draw(prm.lmg_seq[1] + c * (uint32_t)(Q - 1) + (uint32_t)(j - 1));
#pragma unroll
      for (int k = 1; k < Q; ++k) {
        double uk = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j)
          if (j <= k) uk += l0[k * Q + j] * z[j];
        m.slope_u[(size_t)c * (Q - 1) + (k - 1)] = (float)uk;
      }
    }
  }
  gq_sync();
  // latent-scale random effect of a row: primary intercept, then the primary slopes (data_gen.rs:812-842)
  auto row_effect = [&](uint32_t rr, const float (&x)[P]) -> double {
    uint32_t cid = prm.cluster_fixed_clusters ? rr % cv : rr / cv;
    cid = cid < ncl_max ? cid : 0u;
    double u_re = (double)m.cluster_u[cid];
#pragma unroll
    for (int s2 = 1; s2 < Q; ++s2) {
      float xv = 0.0f;
#pragma unroll
      for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[0][s2 - 1] ? x[c] : xv;
      u_re += (double)m.slope_u[(size_t)cid * (Q - 1) + (s2 - 1)] * (double)xv;
    }
    return u_re;
  };
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
        float x[P], y;
        const float e = build_x<P, true, QW, SCEN>(prm, cx, k0, k1, m.st, r, rr, x, &sl);
        if (bmask == 0u) finish_row<P, 1, SCEN>(prm, cx, x, e, y, row_effect(rr, x), &sl);
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
      float y;
      finish_row<P, 1, SCEN>(prm, cx, x, e, y, row_effect(rr, x), &sl);
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
  gdd* carry = m.tail + GT_SPARE + 1;  // theta-hat of the last converged grid point (NTH <= 6 of the 7 spare doubles)
  const uint32_t nt = prm.n_tests;
  // fitted columns: all of them, or the test_formula subset (no factor exclusion on clustered specs, batch.rs:1583)
  const uint32_t keep = prm.glmm_keep;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    const uint32_t ncl = prm.cluster_fixed_clusters ? cv : (n + cv - 1u) / cv;
    gdd* x0 = m.simplex;
    if (lane == 0u) {
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k) {  // carried theta-hat / truth / blind THETA0 (identity template)
        double v = has_prev ? carry[k] : (prm.truth_start ? prm.lmg_theta_truth[k] : (glmm_is_diag<Q>(k) ? 1.0 : 0.0));
        if ((has_prev || prm.truth_start) && v < 0.01) v = 0.01;
        x0[k] = v;
      }
      uint32_t k = NTH;
      for (uint32_t c = 0; c < (uint32_t)P; ++c)
        if ((keep >> c) & 1u) x0[k++] = prm.beta[c];                      // beta starts at the truth, unconditionally
    }
    gq_sync();
    // ---- the fit: one wave-uniform state machine around the single inlined evaluation (mcp_fit_fsm.hpp) ----
    const uint32_t d_fit = NTH + (uint32_t)__popc(keep);
    uint32_t lower0 = 0u;
#pragma unroll
    for (uint32_t k = 0; k < NTH; ++k) lower0 |= glmm_is_diag<Q>(k) ? (1u << k) : 0u;
    bool ok = false;
    double th[NTH];
#pragma unroll
    for (uint32_t k = 0; k < NTH; ++k) th[k] = 0.0;
    auto eval_pt = [&](gdd* x) -> double {  // clamp the point onto the bounds, then the Laplace deviance
      if (lane < d_fit) {
        double v = x[lane];
        if (lane < NTH) {
          if (glmm_is_diag<Q>(lane)) v = v < 0.0 ? 0.0 : (v > GLMM_THETA_MAX ? GLMM_THETA_MAX : v);
          else v = v < -GLMM_THETA_MAX ? -GLMM_THETA_MAX : (v > GLMM_THETA_MAX ? GLMM_THETA_MAX : v);
        } else {
          v = v < -50.0 ? -50.0 : (v > 50.0 ? 50.0 : v);
        }
        x[lane] = v;
      }
      gq_sync();
      return glmm_eval<P, Q>(prm, m, x, lane, n, ncl, n_pad, keep);
    };
    auto on_final = [&](double cur, uint32_t n_eval) -> uint32_t {
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k) th[k] = x0[k];
      double T[Q][Q];
      {
        int k = 0;
#pragma unroll
        for (int c = 0; c < Q; ++c)
#pragma unroll
          for (int r = 0; r < Q; ++r) {
            T[r][c] = r >= c ? th[k] : 0.0;
            k += r >= c ? 1 : 0;
          }
      }
      ok = cur < __builtin_inf() && n_eval < GLMM_MAX_EVAL;
      // ---- RX'RX = X'WX - sum_c C_c' M_c^-1 C_c, C_c = A_c'W X_c (Q x P), M_c = Lm Lm', at the optimum, in LDS
      //      (mcp_glmm_tail.hpp): lane = row leaves the IRLS weight at the final modes in eta0's slot (every evaluation
      //      rebuilds eta0); lane = entry (r, s) of the lower triangle then sums, cluster by cluster and rows ascending,
      //      its product and the two columns r, s of C_c, solves them against the cluster's factor and subtracts ----
      auto a_of_row = [&](uint32_t i, double (&a)[Q]) {
        double z[Q];
        z[0] = 1.0;
#pragma unroll
        for (int dd = 1; dd < Q; ++dd) z[dd] = (double)m.design[(size_t)(prm.lmg_cols[0][dd - 1] - 1u) * n_pad + i];
#pragma unroll
        for (int b = 0; b < Q; ++b) {
          double sacc = 0.0;
#pragma unroll
          for (int dd = 0; dd < Q; ++dd)
            if (dd >= b) sacc += T[dd][b] * z[dd];
          a[b] = sacc;
        }
      };
      for (uint32_t i = lane; i < n; i += QW) {
        const uint32_t cid = prm.cluster_fixed_clusters ? i % cv : i / cv;
        double a[Q], au = 0.0;
        a_of_row(i, a);
#pragma unroll
        for (int b = 0; b < Q; ++b) au += a[b] * m.ucl[(size_t)cid * Q + b];
        double w, wz, dv;
        fam_eval(prm.glm_family, m.eta0[i] + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
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
          double loc = 0.0, cr[Q], cs[Q];
#pragma unroll
          for (int b = 0; b < Q; ++b) cr[b] = cs[b] = 0.0;
          for (uint32_t i = first; i < last; i += step) {
            const double w = m.eta0[i];
            const double xr = r == 0u ? 1.0 : (double)m.design[(size_t)(r - 1u) * n_pad + i];
            const double xs = s == 0u ? 1.0 : (double)m.design[(size_t)(s - 1u) * n_pad + i];
            double a[Q];
            a_of_row(i, a);
#pragma unroll
            for (int b = 0; b < Q; ++b) {
              cr[b] += w * a[b] * xr;
              cs[b] += w * a[b] * xs;
            }
            loc += w * xr * xs;
          }
          double v = 0.0;  // D = Lm^-1 C, columns r and s
#pragma unroll
          for (int b = 0; b < Q; ++b) {
            double vr = cr[b], vs = cs[b];
#pragma unroll
            for (int k2 = 0; k2 < Q; ++k2)
              if (k2 < b) {
                const double l = m.lmcl[((size_t)c * Q + b) * Q + k2];
                vr -= l * cr[k2];
                vs -= l * cs[k2];
              }
            const double lbb = m.lmcl[((size_t)c * Q + b) * Q + b];
            cr[b] = vr / lbb;
            cs[b] = vs / lbb;
            v += cr[b] * cs[b];
          }
          acc += loc - v;
        }
        m.tail[GT_LX + r * 8u + s] = acc;
      }
      if (lane == 0u) {  // fitted coefficients in column order; a column outside a test_formula subset has coefficient 0
        uint32_t k = NTH;
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
      // Wald SE mode "hessian" (mcp_fit_fsm.hpp) over the FREE covariance parameters and beta: a pinned diagonal is an even
      // point of the deviance and drops out, the coupling with the other parameters (what "rx" ignores) stays; a block
      // that does not factor keeps "rx"
      uint32_t th_free = 0u;
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k) th_free |= (glmm_is_diag<Q>(k) && th[k] < GLMM_THETA_ZERO) ? 0u : (1u << k);
      return (prm.glmm_hessian && ok) ? (FIT_FSM_HESSIAN | th_free) : 0u;
    };
    const FitFsmOut fit = simplex_fit_fsm(m.simplex, m.fv, d_fit, (uint32_t)GLMM_ROW, lower0, NTH, GLMM_MAX_EVAL, lane, eval_pt, on_final);
    if (fit.hess_ok && lane == 0u) glmm_tail_take_hessian(m.tail, m.simplex + GLMM_ROW, (uint32_t)GLMM_ROW, NTH, (uint32_t)P, keep);
    bool conv = ok;
    bool at_upper = false;
    uint32_t pinned = 0u;
    {
      uint32_t comp = 0;
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k)
        if (glmm_is_diag<Q>(k)) {
          if (conv && th[k] < GLMM_THETA_ZERO) pinned |= 1u << comp;
          at_upper = at_upper || th[k] > GLMM_THETA_MAX - 1e-3;
          ++comp;
        }
    }
    const uint32_t bh = !conv ? 0u : (at_upper ? 2u : (pinned ? 1u : 0u));
    if (at_upper) conv = false;
    if (conv) {
      if (lane == 0u) {
#pragma unroll
        for (uint32_t k = 0; k < NTH; ++k) carry[k] = th[k];
      }
      has_prev = true;
    }
    // ---- Wald z^2 (dispersion 1), joint Wald chi2 over the targets vs chi2_ppf(1 - alpha, k), decisions: one lane, in LDS ----
    if (lane == 0u)
      glmm_tail_decide(prm, m.tail, (uint32_t)P, nt, conv, bh, pinned, th[0] * th[0], prm.crit + (size_t)g * prm.crit_stride,
                       (size_t)(prm.out_base + lsim) * prm.n_grid + g, (size_t)(prm.out_base + lsim), g == 0u);
    gq_sync();
  }
}

}  // namespace

size_t glmm_slopes_lds_bytes(int P, int Q, uint32_t max_n, uint32_t ncl_max) {
  const size_t n_pad = ((size_t)max_n + 3u) & ~(size_t)3u;
  const size_t ncl_pad = ((size_t)ncl_max + 1u) & ~(size_t)1u;
  const size_t d = (size_t)(Q * (Q + 1) / 2) + (size_t)P;
  const size_t dbl = n_pad + (1 + (size_t)Q + (size_t)Q * Q) * ncl_pad + (d + 4) * GLMM_ROW + d + 2 + (8 + 64 + 64 + 64 + 8 + 8);
  const size_t flt = (size_t)P * n_pad + ncl_pad * (size_t)Q + (size_t)4 * (P + 1) * QW;
  return dbl * 8 + flt * 4 + 64;
}

hipError_t launch_glmm_slopes(int P, int Q, const SimParams* d_prm, uint32_t n_sims, uint32_t max_n, uint32_t ncl_max, bool scen,
                              hipStream_t st) {
  const size_t lds = glmm_slopes_lds_bytes(P, Q, max_n, ncl_max);
  const dim3 block(QW), grid(n_sims);
#define GQ(PP, QQ)                                                                                                     \
  if (P == PP && Q == QQ) {                                                                                            \
    if (lds > 48 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute(scen ? (const void*)glmm_slopes_kernel<PP, QQ, true>                          \
                                              : (const void*)glmm_slopes_kernel<PP, QQ, false>,                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
    }                                                                                                                  \
    if (scen) hipLaunchKernelGGL((glmm_slopes_kernel<PP, QQ, true>), grid, block, lds, st, d_prm);                     \
    else hipLaunchKernelGGL((glmm_slopes_kernel<PP, QQ, false>), grid, block, lds, st, d_prm);                         \
    return hipGetLastError();                                                                                          \
  }
  GQ(2, 2) GQ(3, 2) GQ(3, 3) GQ(4, 2) GQ(4, 3) GQ(5, 2) GQ(5, 3) GQ(6, 2) GQ(6, 3) GQ(7, 2) GQ(7, 3) GQ(8, 2) GQ(8, 3)
#undef GQ
  return hipErrorInvalidValue;
}

}  // namespace mcp
