// glmm_general_kernels.hip -- gfx950 kernel of the clustered GLM with EXTRA GROUPINGS (crossed with / nested in the
// primary one, random slopes on any of them): the structure glmm_kernels.hip / glmm_slopes_kernels.hip cannot split
// by cluster.  Same objective (lme4's glmer, Laplace), same simplex; the conditional modes are found jointly:
//   d(theta, beta) = sum_i devres_i + |u~|^2 + log|I + A'WA|,   u <- (I + A'WA)^-1 A'W (z - eta0),   A = Z Lambda_theta,
// over all R = sum_t q_t L_t random effects, following oracle/fit_glmm_general.c operation for operation.
//
// One wavefront per replicate, everything in LDS (R <= ~115: the R x R system is the big tenant).  One Fisher
// step: lane = row evaluates the family kernel (weight, working residual, deviance); then I + A'WA is assembled with every
// lane busy -- 2^k lanes share a level for its q x q diagonal block and right-hand side (contiguous chunks of its rows, the
// partial sums added in lane order: a fixed association), one lane per (level, level) pair for the strips between two
// terms (it walks the smaller level's rows in ascending order: every strip entry sums its rows in row order) -- no two
// lanes ever touch the same entry; a
// right-looking dense Cholesky with lane = row (entries update in ascending pivot order: the oracle's dot-product
// order) skips exact zeros, which is most of the primary block-diagonal part; two triangular solves give the step.
// Wald covariance "rx": X'WX - (A'WX)' (I + A'WA)^-1 (A'WX) from the last factor.
// (clustered branch of the Glm arm, crates/engine-core/src/batch.rs:1575-1830; structure: mixed_workspace.rs:112-198)
#ifndef MCP_GLMM_WAVES
#define MCP_GLMM_WAVES 2
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
constexpr uint32_t GLMM_MAX_EVAL = 30000;
constexpr double GLMM_THETA_MAX = 32.0;
constexpr double GLMM_THETA_ZERO = 1e-6;
constexpr double GLMM_NEAR_ZERO = 1e-30;
constexpr int GLMM_ROW = 20;  // simplex row stride (d = n_theta + P <= 12 + 8)
constexpr uint32_t GG_MAX_TERMS = 3;

typedef __attribute__((address_space(3))) double gdd;
#include "mcp_stationary.hpp"  // MCP_STAT_H / MCP_STAT_TOL; the probes run as states of the machine below
typedef __attribute__((address_space(3))) float gdf;
typedef __attribute__((address_space(3))) uint16_t gdu;

struct GlmmLds {
  gdd *eta0, *wrow, *res, *M, *u, *ut, *r, *un, *C, *T, *av, *simplex, *fv, *tail;
  uint32_t ao[GG_MAX_TERMS];  // plane of term t's first entry in av (planes of n_pad rows)
  gdf *design, *draws;
  gdu* lvl;          // level of a row in extra term t at lvl[(t - 1) * n_pad + i]
  gdu* rows;         // rows of extra term t grouped by level (ascending within a level) at rows[(t - 1) * n_pad + ..]
  gdu* lstart;       // first slot of level l of extra term t at lstart[so[t] + l] (levels + 1 entries)
  uint32_t so[GG_MAX_TERMS];
  float* st;
  uint32_t R, RS;    // random effects in total, row stride of M (odd: lanes on different rows spread over the banks)
  uint32_t R0, q0;   // the leading, block-diagonal part of the index layout: the columns of term prm.lmg_elim, its block width
  uint32_t off[GG_MAX_TERMS], doff[GG_MAX_TERMS], diag_mask;
};

#include "mcp_glmm_tail.hpp"

__device__ __forceinline__ void gq_sync() { __syncthreads(); }

// entry t of a 3-entry table held in registers: constant indices only, so the table never lands in scratch
__device__ __forceinline__ uint32_t pick3(const uint32_t (&a)[3], uint32_t t) { return t == 0u ? a[0] : (t == 1u ? a[1] : a[2]); }

__device__ __forceinline__ double gq_wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < QW; m <<= 1) v += __shfl_xor(v, m, QW);
  return v;
}

// extra_level_of_row (generation.rs:402-427)
__device__ __forceinline__ uint32_t gg_level_of_row(const SimParams& prm, uint32_t t, uint32_t i) {
  const uint32_t v = prm.cluster_v;
  if (t == 0u) return prm.cluster_fixed_clusters ? i % v : i / v;
  const uint32_t bl = prm.lmg_count[t];
  if (prm.cluster_fixed_clusters) {
    uint32_t stride = v;
    for (uint32_t h = 1; h < t; ++h) stride *= prm.lmg_count[h];
    const uint32_t within = (i / stride) % bl;
    return prm.lmg_relation[t] == 0u ? within : (i % v) * bl + within;
  }
  return (i / v) * bl + (i % v) % bl;
}

// a = Lambda_t' z of row i for term t (q entries), computed from the templates in m.T
template <int P>
__device__ __forceinline__ void gg_row_term_compute(const SimParams& prm, const GlmmLds& m, uint32_t t, uint32_t i, uint32_t n_pad,
                                                double (&a)[3]) {
  const uint32_t q = prm.lmg_q[t];
  double z[3];
  z[0] = 1.0;
  z[1] = q > 1u ? (double)m.design[(size_t)(prm.lmg_cols[t][0] - 1u) * n_pad + i] : 0.0;
  z[2] = q > 2u ? (double)m.design[(size_t)(prm.lmg_cols[t][1] - 1u) * n_pad + i] : 0.0;
  const gdd* T = m.T + t * 9u;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    double s = 0.0;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd)
      if (dd >= b) s += ((uint32_t)dd < q && (uint32_t)b < q) ? T[dd * 3 + b] * z[dd] : 0.0;
    a[b] = s;
  }
}

// the same vector from the per-evaluation cache m.av (the templates do not move during the Fisher steps of one
// evaluation); returns the row's level in that term
template <int P>
__device__ __forceinline__ uint32_t gg_row_term(const SimParams& prm, const GlmmLds& m, uint32_t t, uint32_t i, uint32_t n_pad,
                                                double (&a)[3]) {
  const uint32_t q = prm.lmg_q[t];
  const gdd* av = m.av + (size_t)pick3(m.ao, t) * n_pad + i;
  a[0] = av[0];
  a[1] = q > 1u ? av[n_pad] : 0.0;
  a[2] = q > 2u ? av[2u * n_pad] : 0.0;
  return t == 0u ? gg_level_of_row(prm, 0u, i) : (uint32_t)m.lvl[(size_t)(t - 1u) * n_pad + i];
}

// in-place lower Cholesky of the R x R matrix (row stride RS).  The leading R0 rows / columns are the primary
// grouping's q0 x q0 diagonal blocks, everything below them is dense.  Phase 1: lane = primary level factors its block
// and scales the dense rows of its columns (left-looking: each entry subtracts its products in ascending pivot order,
// then divides -- the operation sequence of the plain right-looking loop).  Phase 2: lanes over the (row, column)
// pairs of the dense block subtract the primary columns' products, ascending.  Phase 3: right-looking on the dense
// block, lane = row.  log|.| sums the pivots in index order.
__device__ __forceinline__ bool gg_chol(gdd* M, uint32_t R, uint32_t RS, uint32_t R0, uint32_t q0, uint32_t lane, gdd* tmp,
                                     double& logdet) {
  bool ok = true;
  const uint32_t L0 = R0 / q0, Re = R - R0;
  gq_sync();
  for (uint32_t l = lane; l < L0; l += QW) {
    const uint32_t base = l * q0;
    for (uint32_t jj = 0; jj < q0; ++jj) {
      const uint32_t j = base + jj;
      double dj = M[(size_t)j * RS + j];
      for (uint32_t k = 0; k < jj; ++k) dj -= M[(size_t)j * RS + base + k] * M[(size_t)j * RS + base + k];
      ok = ok && (dj > 0.0) && (dj < __builtin_inf());
      const double ljj = __builtin_sqrt(dj);
      M[(size_t)j * RS + j] = ljj;
      for (uint32_t ii = jj + 1u; ii < q0; ++ii) {
        const uint32_t i = base + ii;
        double v = M[(size_t)i * RS + j];
        for (uint32_t k = 0; k < jj; ++k) v -= M[(size_t)i * RS + base + k] * M[(size_t)j * RS + base + k];
        M[(size_t)i * RS + j] = v / ljj;
      }
      for (uint32_t i = R0; i < R; ++i) {
        double v = M[(size_t)i * RS + j];
        for (uint32_t k = 0; k < jj; ++k) v -= M[(size_t)i * RS + base + k] * M[(size_t)j * RS + base + k];
        M[(size_t)i * RS + j] = v / ljj;
      }
    }
  }
  gq_sync();
  for (uint32_t e = lane; e < Re * (Re + 1u) / 2u; e += QW) {
    uint32_t ii = (uint32_t)((__builtin_sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while (ii * (ii + 1u) / 2u > e) --ii;
    while ((ii + 1u) * (ii + 2u) / 2u <= e) ++ii;
    const uint32_t cc = e - ii * (ii + 1u) / 2u;
    const gdd* Mi = M + (size_t)(R0 + ii) * RS;
    const gdd* Mc = M + (size_t)(R0 + cc) * RS;
    double v = Mi[R0 + cc];
    for (uint32_t j = 0; j < R0; ++j) {
      const double lij = Mi[j];
      if (lij != 0.0) v -= lij * Mc[j];
    }
    M[(size_t)(R0 + ii) * RS + R0 + cc] = v;
  }
  for (uint32_t j = R0; j < R; ++j) {
    gq_sync();
    const double dj = M[(size_t)j * RS + j];
    ok = ok && (dj > 0.0) && (dj < __builtin_inf());
    const double ljj = __builtin_sqrt(dj);
    gq_sync();
    if (lane == 0u) M[(size_t)j * RS + j] = ljj;
    for (uint32_t i = j + 1u + lane; i < R; i += QW) M[(size_t)i * RS + j] = M[(size_t)i * RS + j] / ljj;
    gq_sync();
    for (uint32_t i = j + 1u + lane; i < R; i += QW) {
      const double lij = M[(size_t)i * RS + j];
      if (lij == 0.0) continue;
      gdd* Mi = M + (size_t)i * RS;
      uint32_t c = j + 1u;
      for (; c + 3u <= i; c += 4u) {  // the pivot column is read-only here: loads first, then four independent updates
        const double l0 = M[(size_t)c * RS + j], l1 = M[(size_t)(c + 1u) * RS + j];
        const double l2 = M[(size_t)(c + 2u) * RS + j], l3 = M[(size_t)(c + 3u) * RS + j];
        const double m0 = Mi[c], m1 = Mi[c + 1u], m2 = Mi[c + 2u], m3 = Mi[c + 3u];
        Mi[c] = m0 - lij * l0;
        Mi[c + 1u] = m1 - lij * l1;
        Mi[c + 2u] = m2 - lij * l2;
        Mi[c + 3u] = m3 - lij * l3;
      }
      for (; c <= i; ++c) Mi[c] -= lij * M[(size_t)c * RS + j];
    }
  }
  gq_sync();
  for (uint32_t a = lane; a < R; a += QW) tmp[a] = 2.0 * log(M[(size_t)a * RS + a]);
  ok = __ballot(!ok) == 0ull;
  gq_sync();
  double ld = 0.0;
  for (uint32_t a = 0; a < R; ++a) ld += tmp[a];
  gq_sync();
  logdet = ld;
  return ok;
}

// L x = b in place (b: nrhs columns, row stride bs): primary blocks by their lanes, the dense rows' products with them
// in ascending column order, then the dense block column by column -- every entry subtracts in the dot product's order
__device__ __forceinline__ void gg_forward(const gdd* M, uint32_t R, uint32_t RS, uint32_t R0, uint32_t q0, gdd* b, uint32_t bs,
                                        uint32_t nrhs, uint32_t lane) {
  const uint32_t L0 = R0 / q0;
  gq_sync();
  for (uint32_t l = lane; l < L0; l += QW) {
    const uint32_t base = l * q0;
    for (uint32_t jj = 0; jj < q0; ++jj) {
      const uint32_t j = base + jj;
      for (uint32_t c = 0; c < nrhs; ++c) {
        double v = b[(size_t)j * bs + c];
        for (uint32_t k = 0; k < jj; ++k) v -= M[(size_t)j * RS + base + k] * b[(size_t)(base + k) * bs + c];
        b[(size_t)j * bs + c] = v / M[(size_t)j * RS + j];
      }
    }
  }
  gq_sync();
  for (uint32_t e = lane; e < (R - R0) * nrhs; e += QW) {
    const uint32_t a = R0 + e / nrhs, c = e % nrhs;
    double v = b[(size_t)a * bs + c];
    for (uint32_t j = 0; j < R0; ++j) {
      const double laj = M[(size_t)a * RS + j];
      if (laj != 0.0) v -= laj * b[(size_t)j * bs + c];
    }
    b[(size_t)a * bs + c] = v;
  }
  for (uint32_t k = R0; k < R; ++k) {
    gq_sync();
    const double lkk = M[(size_t)k * RS + k];
    if (lane < nrhs) b[(size_t)k * bs + lane] = b[(size_t)k * bs + lane] / lkk;
    gq_sync();
    for (uint32_t a = k + 1u + lane; a < R; a += QW) {
      const double lak = M[(size_t)a * RS + k];
      if (lak != 0.0)
        for (uint32_t c = 0; c < nrhs; ++c) b[(size_t)a * bs + c] -= lak * b[(size_t)k * bs + c];
    }
  }
  gq_sync();
}

// L' x = t (t in r, consumed; x to un): the dense block from the bottom, then every primary block by its lane
__device__ __forceinline__ void gg_backward(const gdd* M, uint32_t R, uint32_t RS, uint32_t R0, uint32_t q0, gdd* r, gdd* un,
                                         uint32_t lane) {
  for (uint32_t kk = R; kk-- > R0;) {
    gq_sync();
    const double unk = r[kk] / M[(size_t)kk * RS + kk];
    if (lane == 0u) un[kk] = unk;
    gq_sync();
    for (uint32_t a = R0 + lane; a < kk; a += QW) {
      const double lka = M[(size_t)kk * RS + a];
      if (lka != 0.0) r[a] -= lka * unk;
    }
  }
  gq_sync();
  const uint32_t L0 = R0 / q0;
  for (uint32_t l = lane; l < L0; l += QW) {
    const uint32_t base = l * q0;
    for (uint32_t jj = q0; jj-- > 0u;) {
      const uint32_t j = base + jj;
      double v = r[j];
      for (uint32_t e = R; e-- > R0;) {
        const double lej = M[(size_t)e * RS + j];
        if (lej != 0.0) v -= lej * un[e];
      }
      for (uint32_t kk = q0; kk-- > jj + 1u;) v -= M[(size_t)(base + kk) * RS + j] * un[base + kk];
      un[j] = v / M[(size_t)j * RS + j];
    }
  }
  gq_sync();
}

// Laplace deviance at par = (theta vech per term, beta); leaves the modes in m.u and the factor of I + A'WA in m.M
template <int P>
__device__ __forceinline__ double glmm_eval(const SimParams& prm, const GlmmLds& m, const gdd* par, uint32_t lane, uint32_t n,
                                         uint32_t ncl, uint32_t n_pad, uint32_t keep) {
  const uint32_t fam = prm.glm_family, nterms = prm.lmg_n_terms, nth = prm.lmg_n_theta, R = m.R, RS = m.RS;
  const uint32_t cv = prm.cluster_v;
  double beta[P];
  {
    uint32_t k = nth;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const bool kept = (keep >> c) & 1u;
      beta[c] = kept ? par[k] : 0.0;
      k += kept ? 1u : 0u;
    }
  }
  if (lane < 27u) m.T[lane] = 0.0;
  gq_sync();
  if (lane == 0u) {  // lower templates, column-major vech per term (mixed_workspace.rs:112-198)
    uint32_t k = 0;
    for (uint32_t t = 0; t < nterms; ++t)
      for (uint32_t c = 0; c < prm.lmg_q[t]; ++c)
        for (uint32_t r = c; r < prm.lmg_q[t]; ++r) m.T[t * 9u + r * 3u + c] = par[k++];
  }
  for (uint32_t i = lane; i < n; i += QW) {
    double e = beta[0];
#pragma unroll
    for (int c = 1; c < P; ++c) e += (double)m.design[(size_t)(c - 1) * n_pad + i] * beta[c];
    m.eta0[i] = e;
  }
  for (uint32_t a = lane; a < R; a += QW) {
    m.u[a] = 0.0;
    m.ut[a] = 0.0;
  }
  gq_sync();
  for (uint32_t i = lane; i < n; i += QW)
    for (uint32_t t = 0; t < nterms; ++t) {
      double a[3];
      gg_row_term_compute<P>(prm, m, t, i, n_pad, a);
#pragma unroll
      for (int b = 0; b < 3; ++b)  // static indices: a[] stays in registers
        if ((uint32_t)b < prm.lmg_q[t]) m.av[(size_t)(pick3(m.ao, t) + (uint32_t)b) * n_pad + i] = a[b];
    }
  gq_sync();
  // joint conditional modes by Fisher scoring with step halving (oracle/fit_glmm_general.c): m.u, the factor in m.M, dev,
  // uu and logdet belong to the ACCEPTED point, m.ut is the trial; a trial whose penalised deviance is not below the
  // accepted one's is pulled half-way back to it (and costs one row sweep, no factorisation)
  double pd_acc = __builtin_inf(), dev = 0.0, logdet = 0.0, uu = 0.0;
  bool ok = true;
  const int pirls_cap = (int)prm.glmm_pirls_cap;  // GLMM_PIRLS_ITERS unless the diagnostic knob lowered it
  for (int it = 0; it < pirls_cap; ++it) {
    // ---- lane = row: weight, working residual, deviance at the trial modes ----
    double devp = 0.0;
    for (uint32_t i = lane; i < n; i += QW) {
      double au = 0.0;
      for (uint32_t t = 0; t < nterms; ++t) {
        double a[3];
        const uint32_t l = gg_row_term<P>(prm, m, t, i, n_pad, a);
        const gdd* utp = m.ut + pick3(m.off, t) + l * prm.lmg_q[t];
        au += a[0] * utp[0];
        if (prm.lmg_q[t] > 1u) au += a[1] * utp[1];
        if (prm.lmg_q[t] > 2u) au += a[2] * utp[2];
      }
      const double e0 = m.eta0[i];
      double w, wz, dv;
      fam_eval(fam, e0 + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
      m.wrow[i] = w;
      m.res[i] = wz - w * e0;  // w (z - eta0)
      devp += dv;
    }
    const double devt = gq_wave_sum(devp);
    double uup = 0.0;
    for (uint32_t a = lane; a < R; a += QW) uup += m.ut[a] * m.ut[a];
    const double uut = gq_wave_sum(uup);
    const double pd = devt + uut, tol = 1e-13 * (__builtin_fabs(pd) + 1.0);
    if (!(pd <= pd_acc + tol)) {  // worse than the accepted point (or not finite): halve the step
      bool stuck_t = true, stuck_a = true;
      for (uint32_t a = lane; a < R; a += QW) {
        const double tv = m.ut[a], av = m.u[a], half = 0.5 * (tv + av);
        stuck_t = stuck_t && (half == tv);
        stuck_a = stuck_a && (half == av);
        m.ut[a] = half;
      }
      stuck_t = __ballot(!stuck_t) == 0ull;
      stuck_a = __ballot(!stuck_a) == 0ull;
      gq_sync();
      if (stuck_t || stuck_a) break;
      continue;  // m.M still holds the accepted point's factor
    }
    for (uint32_t e = lane; e < R * RS; e += QW) m.M[e] = 0.0;
    gq_sync();
    // ---- (a) per level of every term: its diagonal block of I + A'WA and its entries of A'W(z - eta0).  2^k lanes share a level
    //      (contiguous chunks of its rows; the first lane adds the partial sums of the others in lane order: a fixed
    //      association).  Until round 3 one lane per level walked the level's 30-80 rows AND updated the strips against the
    //      later terms: 6-16 busy lanes, ~10^4 wave-instructions per Fisher step -- nine tenths of an evaluation ----
    const bool fixedc = prm.cluster_fixed_clusters != 0u;
    // rows of level l of term t: slot s of [0, cnt) -> row (ascending); extra terms list rows past this grid point's n last
    auto level_rows = [&](uint32_t t, uint32_t l, uint32_t& first, uint32_t& step, uint32_t& cnt) {
      if (t == 0u) {
        first = fixedc ? l : l * cv;
        step = fixedc ? cv : 1u;
        const uint32_t last = fixedc ? n : ((l + 1u) * cv < n ? (l + 1u) * cv : n);
        cnt = first < last ? (last - first + step - 1u) / step : 0u;
      } else {
        first = (uint32_t)m.lstart[pick3(m.so, t) + l];
        step = 1u;
        cnt = (uint32_t)m.lstart[pick3(m.so, t) + l + 1u] - first;
      }
    };
    for (uint32_t t = 0; t < nterms; ++t) {
      const uint32_t q = prm.lmg_q[t], nl = t == 0u ? ncl : prm.lmg_levels[t];
      uint32_t ns = 1u;
      while (nl * ns * 2u <= (uint32_t)QW && ns < 16u) ns *= 2u;
      const uint32_t per_pass = (uint32_t)QW / ns;
      for (uint32_t lb = 0; lb < nl; lb += per_pass) {
        const uint32_t l = lb + lane / ns, sub = lane % ns;
        double rl[3] = {0.0, 0.0, 0.0}, bl[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if (l < nl) {
          uint32_t first, step, cnt;
          level_rows(t, l, first, step, cnt);
          const uint32_t chunk = (cnt + ns - 1u) / ns;
          const uint32_t lo = sub * chunk, hi = lo + chunk < cnt ? lo + chunk : cnt;
          for (uint32_t sl = lo; sl < hi; ++sl) {
            const uint32_t i = t == 0u ? first + sl * step : (uint32_t)m.rows[(size_t)(t - 1u) * n_pad + first + sl];
            if (i >= n) break;  // ascending within a level: the rest lies past this grid point's sample
            double a[3];
            gg_row_term<P>(prm, m, t, i, n_pad, a);
            const double w = m.wrow[i], rs = m.res[i];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              rl[b] += rs * a[b];
#pragma unroll
              for (int dd = 0; dd < 3; ++dd)
                if (dd <= b) bl[b][dd] += w * a[b] * a[dd];
            }
          }
        }
        for (uint32_t s2 = 1; s2 < ns; ++s2) {  // the level's first lane adds the partial sums of the others in lane order
          const int src = (int)((lane & ~(ns - 1u)) + s2);
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const double v = __shfl(rl[b], src);
            if (sub == 0u) rl[b] += v;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd)
              if (dd <= b) {
                const double v2 = __shfl(bl[b][dd], src);
                if (sub == 0u) bl[b][dd] += v2;
              }
          }
        }
        if (sub == 0u && l < nl) {
          const uint32_t colbase = pick3(m.off, t) + l * q;
#pragma unroll
          for (int b = 0; b < 3; ++b)
            if ((uint32_t)b < q) {
              m.r[colbase + (uint32_t)b] = rl[b];
#pragma unroll
              for (int dd = 0; dd < 3; ++dd)
                if (dd <= b) m.M[(size_t)(colbase + (uint32_t)b) * RS + colbase + (uint32_t)dd] = bl[b][dd] + (b == dd ? 1.0 : 0.0);
            }
        }
      }
    }
    // ---- (b) the strips between two terms: lane = one (level of the earlier term, level of the later term) pair.  It walks the
    //      rows of the earlier term's level in ascending order and adds those that sit in its level of the later term -- the order
    //      in which the level-owning lane used to add them, so every entry keeps its bits.  A grouping nested in the primary
    //      one only pairs a child with its own parent ----
    for (uint32_t t = 0; t + 1u < nterms; ++t)
      for (uint32_t t2 = t + 1u; t2 < nterms; ++t2) {
        const uint32_t q = prm.lmg_q[t], q2 = prm.lmg_q[t2];
        const uint32_t nl = t == 0u ? ncl : prm.lmg_levels[t], nl2 = prm.lmg_levels[t2];
        const bool child = t == 0u && prm.lmg_relation[t2] == 1u;
        const uint32_t cnt2 = prm.lmg_count[t2];
        const uint32_t npairs = child ? nl2 : nl * nl2;
        // ... unless the pairs outnumber the lanes by more than the walk gets shorter (many levels of few rows each, e.g. 24
        // clusters x 6 items): then the earlier term's level walks its rows once and updates its own strip entries, as before
        const uint32_t big = nl > nl2 ? nl : nl2;
        const uint32_t steps_pairs = ((npairs + (uint32_t)QW - 1u) / (uint32_t)QW) * (n / (big ? big : 1u));
        const uint32_t steps_level = ((nl + (uint32_t)QW - 1u) / (uint32_t)QW) * (n / (nl ? nl : 1u));
        if (steps_pairs > steps_level) {
          for (uint32_t l = lane; l < nl; l += QW) {
            uint32_t first, step, cnt;
            level_rows(t, l, first, step, cnt);
            const uint32_t colbase = pick3(m.off, t) + l * q;
            for (uint32_t sl = 0; sl < cnt; ++sl) {
              const uint32_t i = t == 0u ? first + sl * step : (uint32_t)m.rows[(size_t)(t - 1u) * n_pad + first + sl];
              if (i >= n) break;
              double a[3], a2[3];
              gg_row_term<P>(prm, m, t, i, n_pad, a);
              const uint32_t l2 = gg_row_term<P>(prm, m, t2, i, n_pad, a2);
              const uint32_t rowbase = pick3(m.off, t2) + l2 * q2;
              const bool mine_below = colbase > rowbase;
              const double w = m.wrow[i];
#pragma unroll
              for (int b2 = 0; b2 < 3; ++b2)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                  if ((uint32_t)b2 < q2 && (uint32_t)b < q) {
                    const uint32_t ra = rowbase + (uint32_t)b2, ca = colbase + (uint32_t)b;
                    m.M[mine_below ? (size_t)ca * RS + ra : (size_t)ra * RS + ca] += w * a2[b2] * a[b];
                  }
            }
          }
          continue;
        }
        for (uint32_t pe = lane; pe < npairs; pe += QW) {
          const uint32_t l = child ? pe / cnt2 : pe / nl2, l2 = child ? pe : pe % nl2;
          if (l >= nl) continue;  // children of clusters past this grid point's sample
          double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
          // walk the rows of whichever of the two levels is the smaller one (the term with more levels), ascending either way
          const bool walk2 = nl2 > nl;
          const uint32_t tw = walk2 ? t2 : t, lw = walk2 ? l2 : l, to = walk2 ? t : t2, lo2 = walk2 ? l : l2;
          uint32_t first, step, cnt;
          level_rows(tw, lw, first, step, cnt);
          for (uint32_t sl = 0; sl < cnt; ++sl) {
            const uint32_t i = tw == 0u ? first + sl * step : (uint32_t)m.rows[(size_t)(tw - 1u) * n_pad + first + sl];
            if (i >= n) break;
            const uint32_t lother = to == 0u ? gg_level_of_row(prm, 0u, i) : (uint32_t)m.lvl[(size_t)(to - 1u) * n_pad + i];
            if (lother != lo2) continue;
            double a[3], a2[3];
            gg_row_term<P>(prm, m, t, i, n_pad, a);
            gg_row_term<P>(prm, m, t2, i, n_pad, a2);
            const double w = m.wrow[i];
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2)
#pragma unroll
              for (int b = 0; b < 3; ++b) acc[b2][b] += w * a2[b2] * a[b];
          }
          // the lower triangle: the entry lives in the row of whichever term sits later in the index layout
          const uint32_t colbase = pick3(m.off, t) + l * q, rowbase = pick3(m.off, t2) + l2 * q2;
          const bool mine_below = colbase > rowbase;
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              if ((uint32_t)b2 < q2 && (uint32_t)b < q) {
                const uint32_t ra = rowbase + (uint32_t)b2, ca = colbase + (uint32_t)b;
                m.M[mine_below ? (size_t)ca * RS + ra : (size_t)ra * RS + ca] = acc[b2][b];
              }
        }
      }
    gq_sync();
    // levels past this grid point's sample (FixedSize sizing) hold no rows: identity rows, zero modes
    for (uint32_t a = lane; a < R; a += QW) {
      if (m.M[(size_t)a * RS + a] == 0.0) {
        m.M[(size_t)a * RS + a] = 1.0;
        m.r[a] = 0.0;
      }
    }
    gq_sync();
    ok = gg_chol(m.M, R, RS, m.R0, m.q0, lane, m.un, logdet);
    if (!ok) break;
    const bool conv = __builtin_fabs(pd_acc - pd) <= tol;  // pd_acc = inf on the first pass: never
    for (uint32_t a = lane; a < R; a += QW) m.u[a] = m.ut[a];  // accept
    dev = devt;
    uu = uut;
    pd_acc = pd;
    gq_sync();
    if (conv) break;
    gg_forward(m.M, R, RS, m.R0, m.q0, m.r, 1u, 1u, lane);
    gg_backward(m.M, R, RS, m.R0, m.q0, m.r, m.un, lane);
    bool same = true;
    for (uint32_t a = lane; a < R; a += QW) same = same && (m.un[a] == m.u[a]);
    same = __ballot(!same) == 0ull;
    if (same) break;
    for (uint32_t a = lane; a < R; a += QW) m.ut[a] = m.un[a];
    gq_sync();
  }
  if (!(pd_acc < __builtin_inf())) ok = false;  // not even the prior modes evaluate
  gq_sync();
  const double total = dev + uu + logdet;
  return (ok && total < __builtin_inf() && total > -__builtin_inf()) ? total : __builtin_inf();
}

// SCEN: general data path (scenario design records, non-normal marginals, per-study effects) plus the LME knobs --
// random-effect shape and per-replicate tau^2 of every grouping (data_gen.rs:146-161, 1069-1092)
//
// The fit of one grid point is ONE loop around ONE inlined evaluation of the Laplace deviance: the bounded Nelder-Mead
// runs with their restarts, the boundary snaps, the final evaluation, the "rx" covariance and the finite-difference
// Hessian stencil are states of a wave-uniform machine that hands the next point to that single site.  No function is
// called: nothing is spilled around calls, the register allocator sees the whole kernel (one wave per workgroup: up to
// 512 VGPRs) and the kernel runs without scratch.  (The call-structured version spent 1 072 B / lane of scratch and
// wrote 36 GB per 4 000 fits of it; worse, hipcc placed a spill of live constants inside an `if (lane == 0)` region in
// front of a call -- stored under EXEC = lane 0, reloaded under full EXEC: DESIGN.md section 5.)  The sequence of
// evaluated points and every floating-point operation are those of oracle/fit_glmm_general.c.
template <int P, bool SCEN>
__global__ __launch_bounds__(QW) void glmm_general_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ double lds_q[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = replicate_key(prm, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t max_n = prm.max_n, n_pad = (max_n + 3u) & ~3u;
  const uint32_t ncl_max = prm.n_clusters_max;
  const uint32_t cv = prm.cluster_v;
  const uint32_t nterms = prm.lmg_n_terms, nth = prm.lmg_n_theta;
  const uint32_t dmax = nth + (uint32_t)P;
  GlmmLds m;
  {
    // index layout of the random effects: the term with the most columns (prm.lmg_elim, host_lowering.cpp) leads -- its levels
    // are mutually exclusive, so that part of I + A'WA is block diagonal and gg_chol factors it one block per lane -- the
    // others follow in declaration order (glmm_nested: 16 children lead, the dense block is the 8 parents)
    const uint32_t E = prm.lmg_elim;
    m.q0 = prm.lmg_q[E];
    m.R0 = prm.lmg_levels[E] * m.q0;
    m.R = prm.lmg_levels[0] * prm.lmg_q[0] + prm.lmg_R;
    m.RS = m.R | 1u;
    uint32_t nd = 0, dm = 0, k = 0;
#pragma unroll
    for (uint32_t t = 0; t < GG_MAX_TERMS; ++t) {
      m.doff[t] = nd;
      if (t < nterms) {
        nd += prm.lmg_levels[t] * prm.lmg_q[t];
        for (uint32_t c = 0; c < prm.lmg_q[t]; ++c)
          for (uint32_t r = c; r < prm.lmg_q[t]; ++r) dm |= (r == c ? 1u : 0u) << k++;
      }
    }
    m.diag_mask = dm;
    {
      uint32_t o = m.R0;
#pragma unroll
      for (uint32_t t = 0; t < GG_MAX_TERMS; ++t) {
        m.off[t] = t == E ? 0u : o;
        if (t != E && t < nterms) o += prm.lmg_levels[t] * prm.lmg_q[t];
      }
    }
    gdd* q = (gdd*)lds_q;
    m.eta0 = q; q += n_pad;
    m.wrow = q; q += n_pad;
    m.res = q; q += n_pad;
    m.M = q; q += (size_t)m.R * m.RS;
    m.u = q; q += m.R;
    m.ut = q; q += m.R;
    m.r = q; q += m.R;
    m.un = q; q += m.R;
    m.C = q; q += (size_t)m.R * P;
    m.T = q; q += 28;
    {
      uint32_t na = 0;
#pragma unroll
      for (uint32_t t = 0; t < GG_MAX_TERMS; ++t) {
        m.ao[t] = na;
        if (t < nterms) na += prm.lmg_q[t];
      }
      // the per-evaluation cache of the transformed random-effect rows shares its LDS with the generator's random
      // effect draws and staging area, which die when the last row exists (the footprint bounds the occupancy of this
      // latency-bound kernel: 33 -> 30 KB on the glmm_crossed shape, 4 -> 5 workgroups per CU)
      m.av = q;
      const size_t av_d = (size_t)na * n_pad, gen_d = ((size_t)((nd + 1u) & ~1u) + (size_t)4 * (P + 1) * QW + 1u) / 2u;
      q += av_d > gen_d ? av_d : gen_d;
    }
    m.simplex = q; q += (size_t)(dmax + 4u) * GLMM_ROW;
    m.fv = q; q += (dmax + 2u) & ~1u;
    m.tail = q; q += 8 + 64 + 64 + 64 + 8 + 8;  // the once-per-fit tail block of mcp_glmm_tail.hpp
    gdf* f = (gdf*)q;
    m.design = f; f += (size_t)P * n_pad;
    m.draws = (gdf*)m.av;
    m.lvl = (gdu*)f; f += (size_t)(nterms - 1u) * n_pad / 2u + 2u;
    m.rows = (gdu*)f; f += (size_t)(nterms - 1u) * n_pad / 2u + 2u;
    m.lstart = (gdu*)f;
    {
      uint32_t so = 0;
#pragma unroll
      for (uint32_t t = 0; t < GG_MAX_TERMS; ++t) {
        m.so[t] = so;
        if (t >= 1u && t < nterms) so += prm.lmg_levels[t] + 2u;
      }
      f += (so + 3u) / 2u + 1u;
    }
    m.st = (float*)(m.draws + ((nd + 1u) & ~1u)) + lane;
  }
  // ---- 1. random effects on the sequential stream (data_gen.rs:146-284): per grouping its intercepts, then its
  //         correlated slopes through chol(diag(tau) R diag(tau)); draw order: primary intercepts, primary slopes, every
  //         extra grouping's intercepts, every extra grouping's slopes (the lowering's lmg_seq offsets) ----
  auto re_draw = [&](uint32_t ordinal) -> double {
    if (SCEN) return draw_cluster_effect(prm, k0, k1, ordinal);
    return (double)norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, (uint64_t)ordinal)));
  };
  for (uint32_t t = 0; t < nterms; ++t) {
    const uint32_t q = prm.lmg_q[t], nl = prm.lmg_levels[t];
    gdf* ux = m.draws + pick3(m.doff, t);
    double tau_t = prm.lmg_tau[t];
    if (SCEN && prm.scen_tau_sq) {  // realised tau per grouping (ICC jitter, data_gen.rs:1069-1092)
      const double t2 = prm.scen_tau_sq[(size_t)t * prm.scen_stride + lsim];
      tau_t = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
    }
    for (uint32_t c = lane; c < nl; c += QW) ux[(size_t)c * q] = (float)(re_draw(prm.lmg_seq[2u * t] + c) * tau_t);
    gq_sync();
    if (q > 1u) {
      // 3 x 3 frame with constant indices (entries beyond q stay 0): the factor lives in registers, not in scratch
      double lt[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) lt[i][j] = 0.0;
      if (!SCEN) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if ((uint32_t)i < q && (uint32_t)j < q) lt[i][j] = prm.lmg_chol[t][(uint32_t)i * q + (uint32_t)j];
      } else {  // chol_lower (data_gen.rs:55-75) of the replicate's own D
        const double tv[3] = {tau_t, prm.lmg_slope_sd[t][0], prm.lmg_slope_sd[t][1]};
        double dm[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            dm[i][j] = ((uint32_t)i < q && (uint32_t)j < q) ? tv[i] * prm.lmg_recorr[t][(uint32_t)i * q + (uint32_t)j] * tv[j] : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if ((uint32_t)j < q) {
            double diag = dm[j][j];
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (k < j) diag -= lt[j][k] * lt[j][k];
            const double ljj = __builtin_sqrt(diag > 0.0 ? diag : 0.0);
            lt[j][j] = ljj;
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (i > j && (uint32_t)i < q && ljj > 0.0) {
                double sacc = dm[i][j];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  if (k < j) sacc -= lt[i][k] * lt[j][k];
                lt[i][j] = sacc / ljj;
              }
          }
      }
      for (uint32_t c = lane; c < nl; c += QW) {
        double z[3];
        z[0] = tau_t > 0.0 ? (double)ux[(size_t)c * q] / tau_t : 0.0;
        z[1] = q > 1u ? re_draw(prm.lmg_seq[2u * t + 1u] + c * (q - 1u)) : 0.0;
        z[2] = q > 2u ? re_draw(prm.lmg_seq[2u * t + 1u] + c * (q - 1u) + 1u) : 0.0;
#pragma unroll
        for (int k = 1; k < 3; ++k)
          if ((uint32_t)k < q) {
            double uk = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (j <= k) uk += lt[k][j] * z[j];
            ux[(size_t)c * q + (uint32_t)k] = (float)uk;
          }
      }
    }
    gq_sync();
  }
  for (uint32_t t = 1; t < nterms; ++t)
    for (uint32_t i = lane; i < max_n; i += QW) m.lvl[(size_t)(t - 1u) * n_pad + i] = (uint16_t)gg_level_of_row(prm, t, i);
  gq_sync();
  for (uint32_t t = 1; t < nterms; ++t) {  // counting sort of the rows by level, once per replicate
    const uint32_t nl = prm.lmg_levels[t];
    const gdu* lv = m.lvl + (size_t)(t - 1u) * n_pad;
    gdu* ls = m.lstart + pick3(m.so, t);
    for (uint32_t l = lane; l < nl; l += QW) {
      uint32_t cnt = 0;
      for (uint32_t i = 0; i < max_n; ++i) cnt += (uint32_t)lv[i] == l ? 1u : 0u;
      ls[l + 1u] = (uint16_t)cnt;
    }
    gq_sync();
    if (lane == 0u) {
      ls[0] = 0;
      for (uint32_t l = 0; l < nl; ++l) ls[l + 1u] = (uint16_t)((uint32_t)ls[l] + (uint32_t)ls[l + 1u]);
    }
    gq_sync();
    for (uint32_t l = lane; l < nl; l += QW) {
      uint32_t slot = ls[l];
      for (uint32_t i = 0; i < max_n; ++i)
        if ((uint32_t)lv[i] == l) m.rows[(size_t)(t - 1u) * n_pad + slot++] = (uint16_t)i;
    }
  }
  gq_sync();
  // latent-scale random effect of a row in the reference's order: primary intercept, extras (intercept, slopes),
  // primary slopes (data_gen.rs:812-842)
  auto row_effect = [&](uint32_t rr, const float (&x)[P]) -> double {
    uint32_t cid = gg_level_of_row(prm, 0u, rr);
    cid = cid < ncl_max ? cid : 0u;
    const uint32_t q0 = prm.lmg_q[0];
    double u_re = (double)m.draws[(size_t)cid * q0];
    for (uint32_t t = 1; t < nterms; ++t) {
      const uint32_t q = prm.lmg_q[t], lv = (uint32_t)m.lvl[(size_t)(t - 1u) * n_pad + rr];
      const gdf* ux = m.draws + pick3(m.doff, t) + (size_t)lv * q;
      u_re += (double)ux[0];
      for (uint32_t s2 = 1; s2 < q; ++s2) {
        float xv = 0.0f;
#pragma unroll
        for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[t][s2 - 1u] ? x[c] : xv;
        u_re += (double)ux[s2] * (double)xv;
      }
    }
    for (uint32_t s2 = 1; s2 < q0; ++s2) {
      float xv = 0.0f;
#pragma unroll
      for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[0][s2 - 1u] ? x[c] : xv;
      u_re += (double)m.draws[(size_t)cid * q0 + s2] * (double)xv;
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
  double carry[12];
#pragma unroll
  for (uint32_t k = 0; k < 12u; ++k) carry[k] = 0.0;
  const uint32_t nt = prm.n_tests;
  // fitted columns: all of them, or the test_formula subset (no factor exclusion on clustered specs, batch.rs:1583)
  const uint32_t keep = prm.glmm_keep;
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    const uint32_t ncl = prm.cluster_fixed_clusters ? cv : (n + cv - 1u) / cv;
    gdd* x0 = m.simplex;
    if (lane == 0u) {
#pragma unroll
      for (uint32_t k = 0; k < 12u; ++k)
        if (k < nth) {  // carried theta-hat / truth / blind THETA0 (identity templates)
          double v = has_prev ? carry[k] : (prm.truth_start ? prm.lmg_theta_truth[k] : (((m.diag_mask >> k) & 1u) ? 1.0 : 0.0));
          if ((has_prev || prm.truth_start) && v < 0.01) v = 0.01;
          x0[k] = v;
        }
      uint32_t k = nth;
      for (uint32_t c = 0; c < (uint32_t)P; ++c)
        if ((keep >> c) & 1u) x0[k++] = prm.beta[c];                      // beta starts at the truth, unconditionally
    }
    gq_sync();
    // ---- the fit: a wave-uniform state machine around the single evaluation site ----
    enum : int { S_NM_BEGIN, S_NM_F0, S_NM_INIT_SETUP, S_NM_INIT_F, S_NM_LOOP, S_NM_FR, S_NM_FE, S_NM_FC, S_NM_SHRINK_SETUP,
                 S_NM_SHRINK_F, S_NM_END, S_STAT_SETUP, S_STAT_FP, S_STAT_FM, S_CUR, S_CUR_F, S_SNAP_SETUP, S_SNAP_F, S_FINAL, S_FINAL_F, S_HESS_A,
                 S_HESS_DIAG_SETUP, S_HESS_DIAG_F, S_HESS_OFF_B, S_HESS_OFF_SETUP, S_HESS_OFF_F, S_HESS_SOLVE, S_DECIDE };
    const uint32_t d = nth + (uint32_t)__popc(keep);
    gdd* sx = m.simplex;  // vertices 0..d, then centroid, xr, xe (= the stencil point of the Hessian); rows 1..d hold H later
    gdd* fv = m.fv;
    gdd* cen = sx + (size_t)(d + 1u) * GLMM_ROW;
    gdd* xr = cen + GLMM_ROW;
    gdd* xe = xr + GLMM_ROW;
    gdd* Hm = sx + GLMM_ROW;  // H[a][b] at Hm[a * GLMM_ROW + b]
    const double inf = __builtin_inf(), dl = 1e-4;
    int st = S_NM_BEGIN, run = 0, restart = 0;
    uint32_t n_eval = 0, vi = 0, lo = 0, hi = 0, snap_k = 0, ha = 0, hb = 0, th_free = 0u, stat_k = 0;
    int hq = 0;
    double nm_step = 0.25, ftol = 1e-10, xtol = 1e-7, best = 0.0, prev = 0.0, cur = 0.0, f = 0.0;
    double stat_h = 0.0, stat_fp = 0.0;
    bool stat_one = false;
    // a rejected stationarity check (mcp_stationary.hpp: the same probes, here as three states) takes the restart loop
    auto stat_reject = [&]() {
      run = 1; restart = 0;
      if (n_eval < GLMM_MAX_EVAL) {
        prev = best;
        nm_step = 0.05; ftol = 1e-10; xtol = 1e-7;
        st = S_NM_BEGIN;
      } else {
        st = S_CUR;
      }
    };
    double flo = 0.0, fhi = 0.0, fnh = 0.0, fr = 0.0, th_keep = 0.0, fadd = 0.0, fsub = 0.0, hacc = 0.0;
    bool outside = false, ok = false, count = true;
    gdd* pt = sx;
    for (;;) {
      // ---- advance: consume f (the deviance at pt) and set up the next point, until a state asks for an evaluation ----
      bool need = false;
      while (!need && st != S_DECIDE) {
        switch (st) {
          case S_NM_BEGIN:  // one bounded Nelder-Mead run from simplex row 0 with initial edge nm_step
            pt = sx; count = true; need = true; st = S_NM_F0;
            break;
          case S_NM_F0:
            if (lane == 0u) fv[0] = f;
            vi = 0; st = S_NM_INIT_SETUP;
            break;
          case S_NM_INIT_SETUP:
            if (vi < d) {
              gdd* v = sx + (size_t)(vi + 1u) * GLMM_ROW;
              if (lane < d) v[lane] = sx[lane] + (lane == vi ? nm_step : 0.0);
              gq_sync();
              pt = v; count = true; need = true; st = S_NM_INIT_F;
            } else {
              gq_sync();
              st = S_NM_LOOP;
            }
            break;
          case S_NM_INIT_F:
            if (lane == 0u) fv[vi + 1u] = f;
            ++vi; st = S_NM_INIT_SETUP;
            break;
          case S_NM_LOOP: {
            if (!(n_eval < GLMM_MAX_EVAL)) { st = S_NM_END; break; }
            lo = 0; hi = 0;
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
            if (fv[hi] < inf && fv[hi] - fv[lo] <= ftol && size <= xtol) { st = S_NM_END; break; }
            flo = fv[lo]; fhi = fv[hi]; fnh = fv[nh];
            gq_sync();
            if (lane < d) {
              double ssum = 0.0;
              for (uint32_t i = 0; i <= d; ++i)
                if (i != hi) ssum += sx[(size_t)i * GLMM_ROW + lane];
              const double c = ssum / (double)d;
              cen[lane] = c;
              xr[lane] = c + (c - sx[(size_t)hi * GLMM_ROW + lane]);
            }
            gq_sync();
            pt = xr; count = true; need = true; st = S_NM_FR;
          } break;
          case S_NM_FR: {
            fr = f;
            gdd* worst = sx + (size_t)hi * GLMM_ROW;
            if (fr < flo) {
              if (lane < d) xe[lane] = cen[lane] + 2.0 * (xr[lane] - cen[lane]);
              gq_sync();
              pt = xe; count = true; need = true; st = S_NM_FE;
            } else if (fr < fnh) {
              if (lane < d) worst[lane] = xr[lane];
              if (lane == 0u) fv[hi] = fr;
              gq_sync();
              st = S_NM_LOOP;
            } else {
              outside = fr < fhi;
              if (lane < d) xe[lane] = cen[lane] + 0.5 * ((outside ? xr[lane] : worst[lane]) - cen[lane]);
              gq_sync();
              pt = xe; count = true; need = true; st = S_NM_FC;
            }
          } break;
          case S_NM_FE: {
            gdd* worst = sx + (size_t)hi * GLMM_ROW;
            if (lane < d) worst[lane] = f < fr ? xe[lane] : xr[lane];
            if (lane == 0u) fv[hi] = f < fr ? f : fr;
            gq_sync();
            st = S_NM_LOOP;
          } break;
          case S_NM_FC: {
            gdd* worst = sx + (size_t)hi * GLMM_ROW;
            if (f < (outside ? fr : fhi)) {
              if (lane < d) worst[lane] = xe[lane];
              if (lane == 0u) fv[hi] = f;
              gq_sync();
              st = S_NM_LOOP;
            } else {
              vi = 0; st = S_NM_SHRINK_SETUP;
            }
          } break;
          case S_NM_SHRINK_SETUP:
            if (vi <= d && vi == lo) ++vi;
            if (vi <= d) {
              gdd* v = sx + (size_t)vi * GLMM_ROW;
              if (lane < d) v[lane] = sx[(size_t)lo * GLMM_ROW + lane] + 0.5 * (v[lane] - sx[(size_t)lo * GLMM_ROW + lane]);
              gq_sync();
              pt = v; count = true; need = true; st = S_NM_SHRINK_F;
            } else {
              gq_sync();
              st = S_NM_LOOP;
            }
            break;
          case S_NM_SHRINK_F:
            if (lane == 0u) fv[vi] = f;
            gq_sync();
            ++vi; st = S_NM_SHRINK_SETUP;
            break;
          case S_NM_END: {  // the best vertex to row 0; then the restart schedule of the caller
            uint32_t bl = 0;
            for (uint32_t i = 1; i <= d; ++i)
              if (fv[i] < fv[bl]) bl = i;
            best = fv[bl];
            gq_sync();
            double best_k = 0.0;
            if (lane < d) best_k = sx[(size_t)bl * GLMM_ROW + lane];
            gq_sync();
            if (lane < d) sx[lane] = best_k;
            gq_sync();
            if (run == 0) {  // first run: restarts only for a point the stationarity check rejects
              stat_k = 0; st = S_STAT_SETUP;
              break;
            }
            bool again;
            if (!(prev - best > 1e-8)) {
              again = false;
            } else {
              ++restart; again = true;
            }
            if (again && restart < 4 && n_eval < GLMM_MAX_EVAL) {
              prev = best;
              nm_step = restart == 0 ? 0.05 : 0.01; ftol = 1e-10; xtol = 1e-7;
              st = S_NM_BEGIN;
            } else {
              st = S_CUR;
            }
          } break;
          case S_STAT_SETUP: {  // probe +h along coordinate stat_k (the single probe of a parameter pinned at 0)
            if (!(best < inf && best > -inf)) { stat_reject(); break; }
            if (stat_k >= d) { st = S_CUR; break; }
            gq_sync();
            const double pk = sx[stat_k];
            const bool bounded = stat_k < nth && ((m.diag_mask >> stat_k) & 1u);
            stat_h = MCP_STAT_H;
            stat_one = bounded && pk == 0.0;
            if (!stat_one && bounded && pk < stat_h) stat_h = pk;
            if (lane < d) xe[lane] = lane == stat_k ? (stat_one ? stat_h : sx[lane] + stat_h) : sx[lane];
            gq_sync();
            pt = xe; count = true; need = true; st = S_STAT_FP;
          } break;
          case S_STAT_FP:
            stat_fp = f;
            if (stat_one) {
              if (!(f >= best - MCP_STAT_TOL)) stat_reject();
              else { ++stat_k; st = S_STAT_SETUP; }
              break;
            }
            gq_sync();
            if (lane < d) xe[lane] = lane == stat_k ? sx[lane] - stat_h : sx[lane];
            gq_sync();
            pt = xe; count = true; need = true; st = S_STAT_FM;
            break;
          case S_STAT_FM: {
            const double fp = stat_fp, fm = f;
            bool rej = !(fp < inf && fp > -inf && fm < inf && fm > -inf) || fp < best - MCP_STAT_TOL || fm < best - MCP_STAT_TOL;
            if (!rej) {
              const double gk = (fp - fm) / (2.0 * stat_h), Hk = (fp - 2.0 * best + fm) / (stat_h * stat_h);
              rej = Hk > 0.0 ? !(gk * gk <= 2.0 * MCP_STAT_TOL * Hk) : !(__builtin_fabs(fp - fm) <= 2.0 * MCP_STAT_TOL);
            }
            if (rej) stat_reject();
            else { ++stat_k; st = S_STAT_SETUP; }
          } break;
          case S_CUR:
            pt = sx; count = true; need = true; st = S_CUR_F;
            break;
          case S_CUR_F:
            cur = f; snap_k = 0; st = S_SNAP_SETUP;
            break;
          case S_SNAP_SETUP: {  // snap a near-zero diagonal onto the boundary when that costs no deviance
            bool found = false;
            while (snap_k < nth) {
              if ((m.diag_mask >> snap_k) & 1u) {
                th_keep = sx[snap_k];
                if (th_keep != 0.0 && th_keep <= 1e-4) { found = true; break; }
              }
              ++snap_k;
            }
            if (found) {
              gq_sync();
              if (lane == 0u) sx[snap_k] = 0.0;
              gq_sync();
              pt = sx; count = true; need = true; st = S_SNAP_F;
            } else {
              st = S_FINAL;
            }
          } break;
          case S_SNAP_F:
            if (f <= cur + 1e-8) cur = f;
            else {
              gq_sync();
              if (lane == 0u) sx[snap_k] = th_keep;
              gq_sync();
            }
            ++snap_k; st = S_SNAP_SETUP;
            break;
          case S_FINAL:  // modes and the factor of the final point
            pt = sx; count = true; need = true; st = S_FINAL_F;
            break;
          case S_FINAL_F: {
            cur = f;
            ok = cur < inf && n_eval < GLMM_MAX_EVAL;
            // ---- RX'RX = X'WX - C' (I + A'WA)^-1 C, C = A'WX (R x P): weights at the final modes, C folded by its owners ----
            double A[P * (P + 1) / 2];
#pragma unroll
            for (int i = 0; i < P * (P + 1) / 2; ++i) A[i] = 0.0;
            for (uint32_t i = lane; i < n; i += QW) {
              // weights at the FINAL modes, as oracle/fit_glmm_general.c: m.u is the accepted point of the step-halving loop
              // and the factor in m.M belongs to it on every exit
              double au = 0.0;
              for (uint32_t t = 0; t < nterms; ++t) {
                double a[3];
                const uint32_t l = gg_row_term<P>(prm, m, t, i, n_pad, a);
                const gdd* ut = m.u + pick3(m.off, t) + l * prm.lmg_q[t];
                au += a[0] * ut[0];
                if (prm.lmg_q[t] > 1u) au += a[1] * ut[1];
                if (prm.lmg_q[t] > 2u) au += a[2] * ut[2];
              }
              double w, wz, dv;
              fam_eval(prm.glm_family, m.eta0[i] + au, (double)m.design[(size_t)(P - 1) * n_pad + i], w, wz, dv);
              m.wrow[i] = w;
              double xd[P];
              xd[0] = 1.0;
#pragma unroll
              for (int r = 1; r < P; ++r) xd[r] = (double)m.design[(size_t)(r - 1) * n_pad + i];
              int idx = 0;
#pragma unroll
              for (int r = 0; r < P; ++r)
#pragma unroll
                for (int s2 = 0; s2 <= r; ++s2) A[idx++] += w * xd[r] * xd[s2];
            }
            for (uint32_t e = lane; e < m.R * (uint32_t)P; e += QW) m.C[e] = 0.0;
            gq_sync();
            for (uint32_t t = 0; t < nterms; ++t) {
              const uint32_t q = prm.lmg_q[t], nl = t == 0u ? ncl : prm.lmg_levels[t];
              for (uint32_t l = lane; l < nl; l += QW) {
                double cl[3][P];
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                  for (int r = 0; r < P; ++r) cl[b][r] = 0.0;
                // rows of the level: an arithmetic progression for the primary grouping, the level's slice of the row list else
                const uint32_t first = t == 0u ? (prm.cluster_fixed_clusters ? l : l * cv) : (uint32_t)m.lstart[pick3(m.so, t) + l];
                const uint32_t step = t == 0u ? (prm.cluster_fixed_clusters ? cv : 1u) : 1u;
                const uint32_t last = t == 0u ? (prm.cluster_fixed_clusters ? n : ((l + 1u) * cv < n ? (l + 1u) * cv : n))
                                              : (uint32_t)m.lstart[pick3(m.so, t) + l + 1u];
                for (uint32_t s2 = first; s2 < last; s2 += step) {
                  const uint32_t i = t == 0u ? s2 : (uint32_t)m.rows[(size_t)(t - 1u) * n_pad + s2];
                  if (i >= n) break;  // ascending within a level: the rest lies past this grid point's sample
                  double a[3];
                  gg_row_term<P>(prm, m, t, i, n_pad, a);
                  const double w = m.wrow[i];
#pragma unroll
                  for (int r = 0; r < P; ++r) {
                    const double xv = r == 0 ? 1.0 : (double)m.design[(size_t)(r - 1) * n_pad + i];
#pragma unroll
                    for (int b = 0; b < 3; ++b) cl[b][r] += w * a[b] * xv;
                  }
                }
                const uint32_t base = pick3(m.off, t) + l * q;
#pragma unroll
                for (int b = 0; b < 3; ++b)
                  if ((uint32_t)b < q)
#pragma unroll
                    for (int r = 0; r < P; ++r) m.C[(size_t)(base + b) * P + r] = cl[b][r];
              }
            }
            gq_sync();
            gg_forward(m.M, m.R, m.RS, m.R0, m.q0, m.C, (uint32_t)P, (uint32_t)P, lane);  // D = L^-1 C
            {
              double dd[P * (P + 1) / 2];
#pragma unroll
              for (int i = 0; i < P * (P + 1) / 2; ++i) dd[i] = 0.0;
              for (uint32_t a = lane; a < m.R; a += QW) {
                int idx = 0;
#pragma unroll
                for (int r = 0; r < P; ++r)
#pragma unroll
                  for (int s2 = 0; s2 <= r; ++s2) dd[idx++] += m.C[(size_t)a * P + r] * m.C[(size_t)a * P + s2];
              }
#pragma unroll
              for (int i = 0; i < P * (P + 1) / 2; ++i) A[i] = gq_wave_sum(A[i]) - gq_wave_sum(dd[i]);
            }
            // the factor of the "rx" precision, in the LDS tail (mcp_glmm_tail.hpp): nothing of size P^2 stays in registers
            if (lane == 0u) {
              int idx = 0;
#pragma unroll
              for (int r = 0; r < P; ++r)
#pragma unroll
                for (int s2 = 0; s2 <= r; ++s2) m.tail[GT_LX + r * 8 + s2] = A[idx++];
              m.tail[GT_SPARE] = glmm_tail_factor(m.tail + GT_LX, m.tail + GT_LX, (uint32_t)P, keep) ? 1.0 : 0.0;
            }
            gq_sync();
            ok = (m.tail[GT_SPARE] != 0.0) && ok;
            if (prm.glmm_hessian && ok) {
              // Wald SE mode "hessian": lme4's deriv12 stencil (delta = 1e-4) over the FREE covariance parameters and beta
              // (a pinned diagonal is an even point of the deviance and drops out); a block that does not factor keeps "rx"
              th_free = 0u;
              for (uint32_t k = 0; k < nth; ++k) th_free |= (((m.diag_mask >> k) & 1u) && sx[k] < GLMM_THETA_ZERO) ? 0u : (1u << k);
              ha = 0; st = S_HESS_A;
            } else {
              st = S_DECIDE;
            }
          } break;
          case S_HESS_A:
            if (ha >= d) { st = S_HESS_SOLVE; break; }
            if (ha < nth && !((th_free >> ha) & 1u)) {  // left out: identity row / column
              gq_sync();
              if (lane < d) {
                Hm[(size_t)ha * GLMM_ROW + lane] = lane == ha ? 1.0 : 0.0;
                Hm[(size_t)lane * GLMM_ROW + ha] = lane == ha ? 1.0 : 0.0;
              }
              ++ha;
              break;
            }
            hq = 0; st = S_HESS_DIAG_SETUP;
            break;
          case S_HESS_DIAG_SETUP:
            gq_sync();
            if (lane < d) xe[lane] = lane == ha ? sx[lane] + (hq ? -dl : dl) : sx[lane];
            gq_sync();
            pt = xe; count = false; need = true; st = S_HESS_DIAG_F;
            break;
          case S_HESS_DIAG_F:
            if (hq == 0) fadd = f;
            else fsub = f;
            if (++hq < 2) { st = S_HESS_DIAG_SETUP; break; }
            if (lane == 0u) Hm[(size_t)ha * GLMM_ROW + ha] = fadd / (dl * dl) - 2.0 * cur / (dl * dl) + fsub / (dl * dl);
            hb = 0; st = S_HESS_OFF_B;
            break;
          case S_HESS_OFF_B:
            while (hb < ha && hb < nth && !((th_free >> hb) & 1u)) ++hb;  // that row / column is already identity
            if (hb >= ha) { ++ha; st = S_HESS_A; break; }
            hq = 0; hacc = 0.0; st = S_HESS_OFF_SETUP;
            break;
          case S_HESS_OFF_SETUP:
            gq_sync();
            if (lane < d) {
              double v = sx[lane];
              if (lane == hb) v = v + ((hq & 2) ? -dl : dl);
              if (lane == ha) v = v + ((hq & 1) ? -dl : dl);
              xe[lane] = v;
            }
            gq_sync();
            pt = xe; count = false; need = true; st = S_HESS_OFF_F;
            break;
          case S_HESS_OFF_F:
            hacc = hq == 0 ? f : (hq == 3 ? hacc + f : hacc - f);  // f++ - f+- - f-+ + f--, summed in that order
            if (++hq < 4) { st = S_HESS_OFF_SETUP; break; }
            if (lane == 0u) {
              const double h = hacc / (4.0 * dl * dl);
              Hm[(size_t)ha * GLMM_ROW + hb] = h;
              Hm[(size_t)hb * GLMM_ROW + ha] = h;
            }
            ++hb; st = S_HESS_OFF_B;
            break;
          case S_HESS_SOLVE: {  // S / 2, S = H_bb - H_bt H_tt^-1 H_tb, by lane 0; then its factor replaces the "rx" one
            gq_sync();
            if (lane == 0u) {
              bool okh = true;
              const uint32_t pf = d - nth;
              for (uint32_t a = 0; a < d; ++a)
                for (uint32_t b = 0; b < d; ++b) {
                  const double v = Hm[(size_t)a * GLMM_ROW + b];
                  okh = okh && (v < __builtin_inf()) && (v > -__builtin_inf());
                }
              for (uint32_t j = 0; j < nth; ++j) {  // lower Cholesky of H_tt, in place
                double dj = Hm[(size_t)j * GLMM_ROW + j];
                for (uint32_t k = 0; k < j; ++k) dj -= Hm[(size_t)j * GLMM_ROW + k] * Hm[(size_t)j * GLMM_ROW + k];
                okh = okh && (dj > 0.0);
                const double ljj = __builtin_sqrt(dj);
                Hm[(size_t)j * GLMM_ROW + j] = ljj;
                for (uint32_t i = j + 1u; i < nth; ++i) {
                  double v = Hm[(size_t)i * GLMM_ROW + j];
                  for (uint32_t k = 0; k < j; ++k) v -= Hm[(size_t)i * GLMM_ROW + k] * Hm[(size_t)j * GLMM_ROW + k];
                  Hm[(size_t)i * GLMM_ROW + j] = v / ljj;
                }
              }
              for (uint32_t r = 0; r < pf; ++r)  // Y = Lt^-1 H_tb, over H_tb
                for (uint32_t i = 0; i < nth; ++i) {
                  double v = Hm[(size_t)i * GLMM_ROW + nth + r];
                  for (uint32_t k = 0; k < i; ++k) v -= Hm[(size_t)i * GLMM_ROW + k] * Hm[(size_t)k * GLMM_ROW + nth + r];
                  Hm[(size_t)i * GLMM_ROW + nth + r] = v / Hm[(size_t)i * GLMM_ROW + i];
                }
              for (uint32_t r = 0; r < pf; ++r)
                for (uint32_t c = 0; c <= r; ++c) {
                  double v = Hm[(size_t)(nth + r) * GLMM_ROW + nth + c];
                  for (uint32_t k = 0; k < nth; ++k) v -= Hm[(size_t)k * GLMM_ROW + nth + r] * Hm[(size_t)k * GLMM_ROW + nth + c];
                  Hm[(size_t)(nth + r) * GLMM_ROW + nth + c] = 0.5 * v;
                }
              xe[0] = okh ? 1.0 : 0.0;
            }
            gq_sync();
            const bool okh = xe[0] != 0.0;
            gq_sync();
            if (okh && lane == 0u) glmm_tail_take_hessian(m.tail, Hm, (uint32_t)GLMM_ROW, nth, (uint32_t)P, keep);
            gq_sync();
            st = S_DECIDE;
          } break;
          default: st = S_DECIDE; break;
        }
      }
      if (st == S_DECIDE && !need) break;
      // ---- the ONE evaluation site: clamp the point onto the bounds, then the Laplace deviance ----
      if (lane < d) {
        double v = pt[lane];
        if (lane < nth) {
          if ((m.diag_mask >> lane) & 1u) v = v < 0.0 ? 0.0 : (v > GLMM_THETA_MAX ? GLMM_THETA_MAX : v);
          else v = v < -GLMM_THETA_MAX ? -GLMM_THETA_MAX : (v > GLMM_THETA_MAX ? GLMM_THETA_MAX : v);
        } else {
          v = v < -50.0 ? -50.0 : (v > 50.0 ? 50.0 : v);
        }
        pt[lane] = v;
      }
      gq_sync();
      n_eval += count ? 1u : 0u;
      f = glmm_eval<P>(prm, m, pt, lane, n, ncl, n_pad, keep);
    }
    bool conv = ok;
    bool at_upper = false;
    uint32_t pinned = 0u;
    {
      uint32_t comp = 0;
      for (uint32_t k = 0; k < nth; ++k)
        if ((m.diag_mask >> k) & 1u) {
          const double thk = x0[k];
          if (conv && thk < GLMM_THETA_ZERO) pinned |= 1u << comp;
          at_upper = at_upper || thk > GLMM_THETA_MAX - 1e-3;
          ++comp;
        }
    }
    const uint32_t bh = !conv ? 0u : (at_upper ? 2u : (pinned ? 1u : 0u));
    if (at_upper) conv = false;
    const double th0 = x0[0];
    if (conv) {
#pragma unroll
      for (uint32_t k = 0; k < 12u; ++k) carry[k] = k < nth ? x0[k] : 0.0;
      has_prev = true;
    }
    // ---- Wald z^2 (dispersion 1), joint Wald chi2 over the targets vs chi2_ppf(1 - alpha, k), decisions: ONE lane over
    //      the LDS tail (mcp_glmm_tail.hpp) ----
    if (lane == 0u) {
      uint32_t k = nth;
      for (uint32_t c = 0; c < (uint32_t)P; ++c) {
        const bool kept = (keep >> c) & 1u;
        m.tail[GT_BETA + c] = kept ? x0[k] : 0.0;
        k += kept ? 1u : 0u;
      }
      glmm_tail_decide(prm, m.tail, (uint32_t)P, nt, conv, bh, pinned, th0 * th0, prm.crit + (size_t)g * prm.crit_stride,
                       (size_t)(prm.out_base + lsim) * prm.n_grid + g, (size_t)(prm.out_base + lsim), g == 0u);
    }
    gq_sync();
  }
}

}  // namespace

size_t glmm_general_lds_bytes(const SimParams& prm) {
  const size_t P = prm.P, n_pad = ((size_t)prm.max_n + 3u) & ~(size_t)3u;
  const size_t R = (size_t)prm.lmg_levels[0] * prm.lmg_q[0] + prm.lmg_R, RS = R | 1u;
  const size_t dmax = prm.lmg_n_theta + P;
  size_t nd = 0;
  for (uint32_t t = 0; t < prm.lmg_n_terms; ++t) nd += (size_t)prm.lmg_levels[t] * prm.lmg_q[t];
  size_t na = 0;
  for (uint32_t t = 0; t < prm.lmg_n_terms; ++t) na += prm.lmg_q[t];
  // the av cache (na n_pad doubles) and the generator's draws + staging share one region
  const size_t av_d = na * n_pad, gen_d = (((nd + 1) & ~(size_t)1) + (size_t)4 * (P + 1) * QW + 1) / 2;
  const size_t dbl = 3 * n_pad + (av_d > gen_d ? av_d : gen_d) + R * RS + 4 * R + R * P + 28 + (dmax + 4) * GLMM_ROW +
                     ((dmax + 2) & ~(size_t)1) + 216;
  size_t so = 0;
  for (uint32_t t = 1; t < prm.lmg_n_terms; ++t) so += (size_t)prm.lmg_levels[t] + 2;
  const size_t flt = P * n_pad + 2 * ((prm.lmg_n_terms - 1) * n_pad / 2 + 2) + (so + 3) / 2 + 1;
  return dbl * 8 + flt * 4 + 64;
}

hipError_t launch_glmm_general(int P, const SimParams* d_prm, uint32_t n_sims, size_t lds, bool scen, hipStream_t st) {
  const dim3 block(QW), grid(n_sims);
#define GQ(PP)                                                                                                         \
  case PP: {                                                                                                           \
    if (lds > 48 * 1024) {                                                                                             \
      hipError_t e = hipFuncSetAttribute(scen ? (const void*)glmm_general_kernel<PP, true>                             \
                                              : (const void*)glmm_general_kernel<PP, false>,                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
      if (e != hipSuccess) return e;                                                                                   \
    }                                                                                                                  \
    if (scen) hipLaunchKernelGGL((glmm_general_kernel<PP, true>), grid, block, lds, st, d_prm);                        \
    else hipLaunchKernelGGL((glmm_general_kernel<PP, false>), grid, block, lds, st, d_prm);                            \
  } break;
  switch (P) {
    GQ(2) GQ(3) GQ(4) GQ(5) GQ(6) GQ(7) GQ(8)
    default: return hipErrorInvalidValue;
  }
#undef GQ
  return hipGetLastError();
}

}  // namespace mcp
