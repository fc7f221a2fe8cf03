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
__device__ __noinline__ double glmm_eval(const SimParams& prm, const GlmmLds m, const gdd* par, uint32_t lane, uint32_t n,
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
      if (sub == 0u) m.tcl[c] = dev + uu + logdet;
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

template <int P, int Q>
__device__ __forceinline__ double glmm_f(const SimParams& prm, const GlmmLds& m, gdd* x, uint32_t lane, uint32_t n, uint32_t ncl,
                                         uint32_t n_pad, uint32_t keep, uint32_t& n_eval) {
  constexpr uint32_t NTH = Q * (Q + 1) / 2;
  if (lane < NTH + (uint32_t)__popc(keep)) {
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
  ++n_eval;
  return glmm_eval<P, Q>(prm, m, x, lane, n, ncl, n_pad, keep);
}

// one Nelder-Mead run from simplex row 0 with initial edge `step`; the best point ends in row 0
template <int P, int Q>
__device__ __noinline__ double glmm_nelder_mead(const SimParams& prm, const GlmmLds m, double step, double ftol, double xtol,
                                                uint32_t lane, uint32_t n, uint32_t ncl, uint32_t n_pad, uint32_t keep,
                                                uint32_t& n_eval) {
  const uint32_t d = (uint32_t)(Q * (Q + 1) / 2) + (uint32_t)__popc(keep);
  gdd* sx = m.simplex;  // vertices 0..d, then centroid, xr, xe
  gdd* fv = m.fv;
  gdd* cen = sx + (size_t)(d + 1u) * GLMM_ROW;
  gdd* xr = cen + GLMM_ROW;
  gdd* xe = xr + GLMM_ROW;
  const double inf = __builtin_inf();
  const double f0 = glmm_f<P, Q>(prm, m, sx, lane, n, ncl, n_pad, keep, n_eval);
  if (lane == 0u) fv[0] = f0;
  for (uint32_t i = 0; i < d; ++i) {
    gdd* v = sx + (size_t)(i + 1u) * GLMM_ROW;
    if (lane < d) v[lane] = sx[lane] + (lane == i ? step : 0.0);
    gq_sync();
    const double fi = glmm_f<P, Q>(prm, m, v, lane, n, ncl, n_pad, keep, n_eval);
    if (lane == 0u) fv[i + 1u] = fi;
  }
  gq_sync();
  while (n_eval < GLMM_MAX_EVAL) {
    uint32_t lo = 0, hi = 0;
    for (uint32_t i = 1; i <= d; ++i) {
      if (fv[i] < fv[lo]) lo = i;
      if (fv[i] > fv[hi]) hi = i;
    }
    uint32_t nh = lo;
    for (uint32_t i = 0; i <= d; ++i)
      if (i != hi && fv[i] > fv[nh]) nh = i;
    double size = 0.0;
    for (uint32_t i = 0; i <= d; ++i)
      for (uint32_t k = 0; k < d; ++k) {
        const double dd = __builtin_fabs(sx[(size_t)i * GLMM_ROW + k] - sx[(size_t)lo * GLMM_ROW + k]);
        size = dd > size ? dd : size;
      }
    if (fv[hi] < inf && fv[hi] - fv[lo] <= ftol && size <= xtol) break;
    const double flo = fv[lo], fhi = fv[hi], fnh = fv[nh];
    gq_sync();
    if (lane < d) {
      double s = 0.0;
      for (uint32_t i = 0; i <= d; ++i)
        if (i != hi) s += sx[(size_t)i * GLMM_ROW + lane];
      const double c = s / (double)d;
      cen[lane] = c;
      xr[lane] = c + (c - sx[(size_t)hi * GLMM_ROW + lane]);
    }
    gq_sync();
    const double fr = glmm_f<P, Q>(prm, m, xr, lane, n, ncl, n_pad, keep, n_eval);
    gdd* worst = sx + (size_t)hi * GLMM_ROW;
    if (fr < flo) {
      if (lane < d) xe[lane] = cen[lane] + 2.0 * (xr[lane] - cen[lane]);
      gq_sync();
      const double fe = glmm_f<P, Q>(prm, m, xe, lane, n, ncl, n_pad, keep, n_eval);
      if (lane < d) worst[lane] = fe < fr ? xe[lane] : xr[lane];
      if (lane == 0u) fv[hi] = fe < fr ? fe : fr;
    } else if (fr < fnh) {
      if (lane < d) worst[lane] = xr[lane];
      if (lane == 0u) fv[hi] = fr;
    } else {
      const bool outside = fr < fhi;
      if (lane < d) xe[lane] = cen[lane] + 0.5 * ((outside ? xr[lane] : worst[lane]) - cen[lane]);
      gq_sync();
      const double fc = glmm_f<P, Q>(prm, m, xe, lane, n, ncl, n_pad, keep, n_eval);
      if (fc < (outside ? fr : fhi)) {
        if (lane < d) worst[lane] = xe[lane];
        if (lane == 0u) fv[hi] = fc;
      } else {
        for (uint32_t i = 0; i <= d; ++i) {
          if (i == lo) continue;
          gdd* v = sx + (size_t)i * GLMM_ROW;
          if (lane < d) v[lane] = sx[(size_t)lo * GLMM_ROW + lane] + 0.5 * (v[lane] - sx[(size_t)lo * GLMM_ROW + lane]);
          gq_sync();
          const double fi = glmm_f<P, Q>(prm, m, v, lane, n, ncl, n_pad, keep, n_eval);
          if (lane == 0u) fv[i] = fi;
          gq_sync();
        }
      }
    }
    gq_sync();
  }
  uint32_t lo = 0;
  for (uint32_t i = 1; i <= d; ++i)
    if (fv[i] < fv[lo]) lo = i;
  const double best = fv[lo];
  gq_sync();
  double best_k = 0.0;
  if (lane < d) best_k = sx[(size_t)lo * GLMM_ROW + lane];
  gq_sync();
  if (lane < d) sx[lane] = best_k;
  gq_sync();
  return best;
}

// SCEN: general data path (scenario design records, non-normal marginals, per-study effects) plus the LME knobs --
// random-effect shape and per-replicate tau^2 (data_gen.rs:146-161, 1069-1076)
#include "mcp_glmm_hessian.hpp"
#include "mcp_stationary.hpp"

template <int P, int Q, bool SCEN>
__global__ __launch_bounds__(QW, MCP_GLMM_WAVES) void glmm_slopes_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ double lds_q[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);
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
  double carry[NTH];
#pragma unroll
  for (uint32_t k = 0; k < NTH; ++k) carry[k] = 0.0;
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
    uint32_t n_eval = 0;
    double best = glmm_nelder_mead<P, Q>(prm, m, 0.25, 1e-10, 1e-7, lane, n, ncl, n_pad, keep, n_eval);
    // restarts only for a point the stationarity check rejects (mcp_stationary.hpp)
    uint32_t lower0 = 0u;
#pragma unroll
    for (uint32_t k = 0; k < NTH; ++k) lower0 |= glmm_is_diag<Q>(k) ? (1u << k) : 0u;
    auto fprobe = [&](gdd* pt) -> double { return glmm_f<P, Q>(prm, m, pt, lane, n, ncl, n_pad, keep, n_eval); };
    const bool stationary = mixed_stationary(fprobe, m.simplex, (uint32_t)GLMM_ROW, NTH + (uint32_t)__popc(keep), lower0, best, lane);
    for (int restart = 0; !stationary && restart < 4 && n_eval < GLMM_MAX_EVAL; ++restart) {
      const double prev = best;
      best = glmm_nelder_mead<P, Q>(prm, m, restart == 0 ? 0.05 : 0.01, 1e-10, 1e-7, lane, n, ncl, n_pad, keep, n_eval);
      if (!(prev - best > 1e-8)) break;
    }
    double cur = glmm_f<P, Q>(prm, m, x0, lane, n, ncl, n_pad, keep, n_eval);
    for (uint32_t k = 0; k < NTH; ++k) {
      if (!glmm_is_diag<Q>(k)) continue;
      const double th_keep = x0[k];
      if (th_keep != 0.0 && th_keep <= 1e-4) {  // snap onto the boundary when that costs no deviance
        gq_sync();
        if (lane == 0u) x0[k] = 0.0;
        gq_sync();
        const double f0 = glmm_f<P, Q>(prm, m, x0, lane, n, ncl, n_pad, keep, n_eval);
        if (f0 <= cur + 1e-8) cur = f0;
        else {
          gq_sync();
          if (lane == 0u) x0[k] = th_keep;
          gq_sync();
        }
      }
    }
    cur = glmm_f<P, Q>(prm, m, x0, lane, n, ncl, n_pad, keep, n_eval);  // modes and weight sums of the final point
    double th[NTH];
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
    bool ok = cur < __builtin_inf() && n_eval < GLMM_MAX_EVAL;
    // ---- RX'RX = X'WX - sum_c C_c' M_c^-1 C_c, C_c = A_c'W X_c (Q x P), M_c = Lm Lm': lane = cluster, wave sums ----
    double A[P * (P + 1) / 2];
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) A[i] = 0.0;
    for (uint32_t c = lane; c < ncl; c += QW) {
      const uint32_t first = prm.cluster_fixed_clusters ? c : c * cv;
      const uint32_t step = prm.cluster_fixed_clusters ? cv : 1u;
      const uint32_t last = prm.cluster_fixed_clusters ? n : ((c + 1u) * cv < n ? (c + 1u) * cv : n);
      double C[Q][P], loc[P * (P + 1) / 2], uc[Q];
#pragma unroll
      for (int b = 0; b < Q; ++b) {
        uc[b] = m.ucl[(size_t)c * Q + b];
#pragma unroll
        for (int i = 0; i < P; ++i) C[b][i] = 0.0;
      }
#pragma unroll
      for (int i = 0; i < P * (P + 1) / 2; ++i) loc[i] = 0.0;
      for (uint32_t i = first; i < last; i += step) {
        double xd[P];
        xd[0] = 1.0;
#pragma unroll
        for (int r = 1; r < P; ++r) xd[r] = (double)m.design[(size_t)(r - 1) * n_pad + i];
        double z[Q], a[Q], au = 0.0;
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
          au += sacc * uc[b];
        }
        double w, wz, dv;
        fam_eval(prm.glm_family, m.eta0[i] + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
        int idx = 0;
#pragma unroll
        for (int r = 0; r < P; ++r) {
#pragma unroll
          for (int b = 0; b < Q; ++b) C[b][r] += w * a[b] * xd[r];
#pragma unroll
          for (int s2 = 0; s2 <= r; ++s2) loc[idx++] += w * xd[r] * xd[s2];
        }
      }
      double Lm[Q][Q];
#pragma unroll
      for (int b = 0; b < Q; ++b)
#pragma unroll
        for (int dd = 0; dd < Q; ++dd) Lm[b][dd] = m.lmcl[((size_t)c * Q + b) * Q + dd];
#pragma unroll
      for (int r = 0; r < P; ++r) {  // D = Lm^-1 C, column r
        double t[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) {
          double v = C[b][r];
#pragma unroll
          for (int k2 = 0; k2 < Q; ++k2)
            if (k2 < b) v -= Lm[b][k2] * t[k2];
          t[b] = v / Lm[b][b];
        }
#pragma unroll
        for (int b = 0; b < Q; ++b) C[b][r] = t[b];
      }
      int idx = 0;
#pragma unroll
      for (int r = 0; r < P; ++r)
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) {
          double v = 0.0;
#pragma unroll
          for (int b = 0; b < Q; ++b) v += C[b][r] * C[b][s2];
          A[idx] += loc[idx] - v;
          ++idx;
        }
    }
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; ++i) A[i] = gq_wave_sum(A[i]);
    double Lx[P][P];
    // lower Cholesky factor of a packed precision matrix over the kept columns; a dropped column is an identity row /
    // column: the kept block factors as the gathered system
    auto factor = [&](const double (&Ap)[P * (P + 1) / 2], double (&L)[P][P]) -> bool {
      bool fok = true;
      int idx = 0;
      double Af[P][P];
#pragma unroll
      for (int r = 0; r < P; ++r)
#pragma unroll
        for (int s = 0; s <= r; ++s) {
          const double v = Ap[idx++];
          Af[r][s] = (((keep >> r) & 1u) && ((keep >> s) & 1u)) ? v : (r == s ? 1.0 : 0.0);
        }
#pragma unroll
      for (int j = 0; j < P; ++j) {
        double dd = Af[j][j];
#pragma unroll
        for (int t = 0; t < j; ++t) dd -= L[j][t] * L[j][t];
        fok = fok && (dd > 0.0) && (dd < __builtin_inf());
        L[j][j] = __builtin_sqrt(dd);
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
          double v = Af[i][j];
#pragma unroll
          for (int t = 0; t < j; ++t) v -= L[i][t] * L[j][t];
          L[i][j] = v / L[j][j];
        }
      }
      return fok;
    };
    ok = factor(A, Lx) && ok;
    if (prm.glmm_hessian && ok) {
      // Wald SE mode "hessian" (mcp_glmm_hessian.hpp) over the FREE covariance parameters and beta: a pinned diagonal is
      // an even point of the deviance and drops out, the coupling with the other parameters (what "rx" ignores) stays;
      // a block that does not factor keeps "rx"
      uint32_t th_free = 0u;
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k) th_free |= (glmm_is_diag<Q>(k) && th[k] < GLMM_THETA_ZERO) ? 0u : (1u << k);
      auto fdev = [&](gdd* pt) -> double {
        uint32_t scratch = 0;
        return glmm_f<P, Q>(prm, m, pt, lane, n, ncl, n_pad, keep, scratch);
      };
      const uint32_t nth_h = NTH, d_h = nth_h + (uint32_t)__popc(keep);
      if (glmm_fd_hessian_schur(fdev, m.simplex, (uint32_t)GLMM_ROW, nth_h, d_h, cur, lane, th_free)) {
        double Ah[P * (P + 1) / 2], Lh[P][P];
        const gdd* Hs = m.simplex + GLMM_ROW;
        int idx = 0;
        uint32_t kr = 0;
#pragma unroll
        for (int r = 0; r < P; ++r) {
          uint32_t kc = 0;
#pragma unroll
          for (int s = 0; s <= r; ++s) {
            const bool both = ((keep >> r) & 1u) && ((keep >> s) & 1u);
            Ah[idx++] = both ? Hs[(size_t)(nth_h + kr) * GLMM_ROW + nth_h + kc] : 0.0;
            kc += ((keep >> s) & 1u) ? 1u : 0u;
          }
          kr += ((keep >> r) & 1u) ? 1u : 0u;
        }
        if (factor(Ah, Lh)) {
#pragma unroll
          for (int r = 0; r < P; ++r)
#pragma unroll
            for (int s = 0; s <= r; ++s) Lx[r][s] = Lh[r][s];
        }
      }
    }
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
#pragma unroll
      for (uint32_t k = 0; k < NTH; ++k) carry[k] = th[k];
      has_prev = true;
    }
    // ---- Wald z^2 (dispersion 1), joint Wald chi2 over the targets vs chi2_ppf(1 - alpha, k), decisions ----
    double beta[P];
    {
      uint32_t k = NTH;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const bool kept = (keep >> c) & 1u;
        beta[c] = kept ? x0[k] : 0.0;
        k += kept ? 1u : 0u;
      }
    }
    const double* crow = prm.crit + (size_t)g * prm.crit_stride;
    double tsq[MAX_T_WAVE];
    double U[MAX_T_WAVE][P];
#pragma unroll
    for (int t = 0; t < MAX_T_WAVE; ++t) {
      tsq[t] = __builtin_nan("");
#pragma unroll
      for (int i = 0; i < P; ++i) U[t][i] = 0.0;
      if ((uint32_t)t < nt && conv) {
        const double* c = prm.contrast + t * MAX_P;
        double est = 0.0, nq = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
          est = __builtin_fma(c[i], beta[i], est);
          double acc = c[i];
#pragma unroll
          for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lx[i][q], U[t][q], acc);
          U[t][i] = acc / Lx[i][i];
          nq = __builtin_fma(U[t][i], U[t][i], nq);
        }
        if (nq > GLMM_NEAR_ZERO && nq < __builtin_inf()) tsq[t] = est * est / nq;
      }
    }
    bool joint_sig = false;
    if (conv && nt != 0u) {
      double Lc[MAX_T_WAVE][MAX_T_WAVE], cj[MAX_T_WAVE];
      bool jok = true;
      double qsum = 0.0;
#pragma unroll
      for (int j = 0; j < MAX_T_WAVE; ++j) {
        if ((uint32_t)j < nt) {
          double dd = 0.0;
#pragma unroll
          for (int i = 0; i < P; ++i) dd = __builtin_fma(U[j][i], U[j][i], dd);
#pragma unroll
          for (int t = 0; t < j; ++t) dd = __builtin_fma(-Lc[j][t], Lc[j][t], dd);
          jok = jok && (dd > 0.0);
          Lc[j][j] = __builtin_sqrt(dd);
#pragma unroll
          for (int i2 = j + 1; i2 < MAX_T_WAVE; ++i2) {
            if ((uint32_t)i2 < nt) {
              double vv = 0.0;
#pragma unroll
              for (int i = 0; i < P; ++i) vv = __builtin_fma(U[i2][i], U[j][i], vv);
#pragma unroll
              for (int t = 0; t < j; ++t) vv = __builtin_fma(-Lc[i2][t], Lc[j][t], vv);
              Lc[i2][j] = vv / Lc[j][j];
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MAX_T_WAVE; ++i) {
        if ((uint32_t)i < nt) {
          const double* c = prm.contrast + i * MAX_P;
          double vv = 0.0;
#pragma unroll
          for (int q = 0; q < P; ++q) vv = __builtin_fma(c[q], beta[q], vv);
#pragma unroll
          for (int t = 0; t < i; ++t) vv = __builtin_fma(-Lc[i][t], cj[t], vv);
          cj[i] = vv / Lc[i][i];
          qsum = __builtin_fma(cj[i], cj[i], qsum);
        }
      }
      joint_sig = jok && (qsum < __builtin_inf()) && (qsum > prm.joint_crit_k[nt]);
    }
    uint32_t unc_bits = 0, cor_bits = 0;
    decide_tests<MAX_T_WAVE>(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
    if (!conv) {
      unc_bits = 0;
      cor_bits = 0;
    }
    if (lane == 0u) {
      const size_t cell = (size_t)(prm.out_base + lsim) * prm.n_grid + g;
#pragma unroll
      for (int t = 0; t < MAX_T_WAVE; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
      prm.conv[cell] = conv ? 1 : 0;
      prm.boundary[cell] = (uint8_t)bh;
      prm.joint[cell] = joint_sig ? 1 : 0;
      if (prm.pinned) prm.pinned[cell] = (uint16_t)pinned;
      if (prm.tau_hat) prm.tau_hat[cell] = conv ? th[0] * th[0] : __builtin_nan("");
      if (prm.overall) prm.overall[cell] = 0;  // batch.rs:1750-1753
      if (prm.stat && g == 0u) {
#pragma unroll
        for (int t = 0; t < MAX_T_WAVE; ++t)
          if ((uint32_t)t < nt) prm.stat[(size_t)(prm.out_base + lsim) * nt + t] = tsq[t];
      }
    }
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
  GQ(2, 2) GQ(3, 2) GQ(3, 3) GQ(4, 2) GQ(4, 3) GQ(5, 2) GQ(5, 3) GQ(6, 2) GQ(6, 3)
#undef GQ
  return hipErrorInvalidValue;
}

}  // namespace mcp
