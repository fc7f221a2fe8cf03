// glm_impl.hpp -- the IRLS simulate kernel template (see glm_kernels.hip for the mapping); instantiated for P = 2..8 in
// glm_kernels.hip and for the wide designs P = 9..16 (the reference bench's glm_wide, throughput.rs:300-303) in
// glm_wide_kernels.hip -- separate translation units so that the wide instantiations compile in parallel.
#pragma once
#ifndef MCP_GLM_WAVES
#define MCP_GLM_WAVES 1
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

// NT threads (NT / 64 waves) share one replicate when its design is too large for many single-wave blocks per CU: each
// wave reduces as above, the wave totals cross through LDS and every lane adds them in wave order.  NT = 64 is the plain
// wave reduction (same instructions as before the parameter existed).
template <int NT>
__device__ __forceinline__ double block_allreduce_sum(double v, double* red, uint32_t lane) {
  v = wave_allreduce_sum(v);
  if (NT == WAVE) return v;
  if ((lane & 63u) == 0u) red[lane >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < NT / WAVE; ++w) s += red[w];
  __syncthreads();
  return s;
}

// IRLS on the first n rows of the LDS-resident design, all lanes returning identical values.
// MASKED: `keep` selects the fitted columns; dropped columns become identity rows/columns of X'WX and zeros of
// X'Wz, so their coefficients stay exactly 0 and the kept block is solved by the same operations as the gathered
// p_red x p_red system of the reference's reduced round (batch.rs:1893-1925).
// LOGIT: the family is known at compile time (the probit / Poisson arms and their ocml expansions stay out of the
// kernel's register allocation: 180 -> fewer VGPRs, one more wave per SIMD).
// On convergence the squared Wald statistic of every test is formed HERE, from the factor of the converged pass, and
// the factor dies with the pass: nothing of size P^2 is live across passes or after the fit, and the Cholesky runs in
// place on the packed lower triangle of X'WX -- the live set of a pass is one triangle (P(P+1)/2 doubles: 136 at
// P = 16) plus O(P), which is what lets the wide instantiations (P = 13..16) stay in the 512-entry register file.
template <int P, bool MASKED, bool LOGIT, int MT, int NT = WAVE>
__device__ __forceinline__ void irls_fit(const SimParams& prm, const float* design, uint32_t n_pad, uint32_t n, uint32_t lane,
                                         uint32_t keep, uint32_t fam_rt, uint32_t nt, double (&beta)[P], double (&tsq)[MT],
                                         double& dev, bool& conv, double* red = nullptr) {
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
    for (uint32_t i = lane; i < n; i += NT) {
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
    if (NT > WAVE) {  // the wave totals of all P(P+1)/2 + P + 1 sums cross through LDS in one exchange
      constexpr int TRI = P * (P + 1) / 2, NV = TRI + P + 1, NW = NT / WAVE;
      double* tot = red + (size_t)NW * NV;
      if ((lane & 63u) == 0u) {
        double* mine = red + (size_t)(lane >> 6) * NV;
#pragma unroll
        for (int i = 0; i < TRI; ++i) mine[i] = xtwx[i];
#pragma unroll
        for (int i = 0; i < P; ++i) mine[TRI + i] = xtwz[i];
        mine[TRI + P] = dev;
      }
      __syncthreads();
      if (lane < (uint32_t)NV) {  // lane i adds sum i over the waves, in wave order
        double a = red[lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) a += red[(size_t)w * NV + lane];
        tot[lane] = a;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TRI; ++i) xtwx[i] = tot[i];
#pragma unroll
      for (int i = 0; i < P; ++i) xtwz[i] = tot[TRI + i];
      dev = tot[TRI + P];
      __syncthreads();
    }
    if (!(dev < __builtin_inf())) break;  // NaN / inf deviance
    const bool done = it > 0 && __builtin_fabs(dev - dev_prev) / (__builtin_fabs(dev) + 0.1) < IRLS_EPS;
    // Cholesky of X'WX in place on its packed lower triangle, and the weighted least-squares update (all lanes, identical
    // values); L(i, j) = MCP_TRI(i, j) for j < i, and the diagonal slot keeps 1 / L(j, j) -- the only form the solves use
    bool ok = true;
#define MCP_TRI(i, j) xtwx[(i) * ((i) + 1) / 2 + (j)]
    if (MASKED) {
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b)
          MCP_TRI(a, b) = (((keep >> a) & 1u) && ((keep >> b) & 1u)) ? MCP_TRI(a, b) : (a == b ? 1.0 : 0.0);
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      double dd = MCP_TRI(j, j);
#pragma unroll
      for (int t = 0; t < j; ++t) dd = __builtin_fma(-MCP_TRI(j, t), MCP_TRI(j, t), dd);
      ok = ok && (dd > 0.0) && (dd < __builtin_inf());
      const double inv_ljj = 1.0 / __builtin_sqrt(dd);
      MCP_TRI(j, j) = inv_ljj;
#pragma unroll
      for (int i = j + 1; i < P; ++i) {
        double v = MCP_TRI(i, j);
#pragma unroll
        for (int t = 0; t < j; ++t) v = __builtin_fma(-MCP_TRI(i, t), MCP_TRI(j, t), v);
        MCP_TRI(i, j) = v * inv_ljj;
      }
    }
    if (!ok) break;
    if (done) {  // converged: Wald z^2 per test, cov = (X'WX)^-1 = L^-T L^-1 of this pass
      conv = true;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        tsq[t] = __builtin_nan("");
        if ((uint32_t)t < nt && (!MASKED || ((keep >> prm.test_pos[t]) & 1u))) {
          const double* c = prm.contrast + t * MAX_P;
          double est = 0.0, nq = 0.0, u[P];
#pragma unroll
          for (int i = 0; i < P; ++i) {
            est = __builtin_fma(c[i], beta[i], est);
            double acc = c[i];
#pragma unroll
            for (int q = 0; q < i; ++q) acc = __builtin_fma(-MCP_TRI(i, q), u[q], acc);
            u[i] = acc * MCP_TRI(i, i);
            nq = __builtin_fma(u[i], u[i], nq);
          }
          if (nq > GLM_NEAR_ZERO && nq < __builtin_inf()) tsq[t] = est * est / nq;
        }
      }
      break;
    }
    double zf[P], nb[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double v = (!MASKED || ((keep >> i) & 1u)) ? xtwz[i] : 0.0;
#pragma unroll
      for (int t = 0; t < i; ++t) v = __builtin_fma(-MCP_TRI(i, t), zf[t], v);
      zf[i] = v * MCP_TRI(i, i);
    }
    bool capped = false;
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
      double v = zf[i];
#pragma unroll
      for (int t = i + 1; t < P; ++t) v = __builtin_fma(-MCP_TRI(t, i), nb[t], v);
      nb[i] = v * MCP_TRI(i, i);
      capped = capped || !(__builtin_fabs(nb[i]) <= BETA_CAP);
    }
    if (capped) break;
#pragma unroll
    for (int i = 0; i < P; ++i) beta[i] = nb[i];
    dev_prev = dev;
#undef MCP_TRI
  }
  if (!conv) {
#pragma unroll
    for (int t = 0; t < MT; ++t) tsq[t] = __builtin_nan("");
  }
}

// ---- wide designs, P = 9..16 (the reference bench's glm_wide is P = 16) ----
// The row-owning scheme above needs P(P+1)/2 + P accumulators per lane (152 doubles at P = 16): past P = 12 they leave
// the architectural registers for AGPR copies and scratch, and the 152 butterfly all-reduces of a pass outweigh the
// ~3 rows a lane owns at n = 200.  Here lanes own ENTRIES of the normal equations instead:
//   1. lane per row: eta, IRLS weight w and w z of the row into LDS (wk), deviance by one wave all-reduce;
//   2. lane per entry: entry e of [X'WX lower triangle | X'Wz] (<= 152 entries, <= 3 per lane) summed over ALL rows
//      straight from the LDS design, 4 rows per step on 16-byte LDS reads -- no cross-lane reduction at all;
//   3. the Cholesky factor is formed in LDS (am), right-looking: column j scaled by lanes, the trailing entries updated
//      by their owners; per entry the subtractions run in the same order t = 0, 1, ... as in the left-looking register
//      version above, so the factor is the one that version forms from the same matrix;
//   4. the two triangular solves run with lane i owning component i (the finished component is handed on by readlane;
//      beta returns to wave-uniform registers the same way), and on convergence lane t forms the Wald z^2 of test t.
// Live registers: beta and <= 3 accumulators -- nothing of size P^2 is ever in registers.
template <int P, bool MASKED, bool LOGIT, int MT>
__device__ __forceinline__ void irls_fit_wide(const SimParams& prm, const float* design, double* wk, double* am, uint32_t n_pad,
                                              uint32_t n, uint32_t lane, uint32_t keep, uint32_t fam_rt, uint32_t nt,
                                              double (&beta)[P], double (&tsq)[MT], double& dev, bool& conv) {
  constexpr int TRI = P * (P + 1) / 2, RUN = 3;
  const uint32_t fam = LOGIT ? 0u : fam_rt;
  // The runs this wave's lanes own: row a of the triangle cut into runs of RUN consecutive columns b0 .. b0 + len - 1
  // (the products (w x_a) x_b share w x_a), then the right-hand side as one more row, (w z) x_b over all columns b.
  // P = 16: 51 + 6 = 57 runs, one per lane.  Column 0 is the LDS column of ones in front of the design.
  const float* col0 = design - n_pad;
  uint32_t ra = 0, rb0 = 0, rlen = 0;
  bool rrhs = false;
  {
    uint32_t g = lane;
    bool found = false;
#pragma unroll
    for (int r = 0; r <= P; ++r) {
      const uint32_t rowlen = r < P ? (uint32_t)(r + 1) : (uint32_t)P, ng = (rowlen + RUN - 1u) / RUN;
      if (!found && g < ng) {
        found = true;
        ra = r < P ? (uint32_t)r : 0u;
        rrhs = r == P;
        rb0 = RUN * g;
        rlen = rowlen - rb0 < (uint32_t)RUN ? rowlen - rb0 : (uint32_t)RUN;
      }
      g -= found ? 0u : ng;
    }
  }
  static_assert((P * (P + 1) / 2 + P + 2 * (P + 1)) / 3 <= WAVE, "one run per lane");
  dev = __builtin_nan("");
  double dev_prev = __builtin_inf();
  conv = false;
#define MCP_TRI(i, j) am[(i) * ((i) + 1) / 2 + (j)]
  for (int it = 0; it < MAX_IRLS_ITERS; ++it) {
    // ---- 1. rows: weights and working responses ----
    double d = 0.0;
    for (uint32_t i = lane; i < n; i += WAVE) {
      double eta = 0.0;
      eta = __builtin_fma(1.0, beta[0], eta);
#pragma unroll
      for (int c = 1; c < P; ++c) eta = __builtin_fma((double)design[(size_t)(c - 1) * n_pad + i], beta[c], eta);
      const double yi = (double)design[(size_t)(P - 1) * n_pad + i];
      double w, wz, di;
      fam_eval(fam, eta, yi, w, wz, di);
      wk[i] = w;
      wk[n_pad + i] = wz;
      d += di;
    }
    if (lane < 3u && n + lane < ((n + 3u) & ~3u)) {  // rows n .. n4 - 1 of the last group of 4: weight 0 (their x stay finite)
      wk[n + lane] = 0.0;
      wk[n_pad + n + lane] = 0.0;
    }
    dev = wave_allreduce_sum(d);
    if (!(dev < __builtin_inf())) break;  // NaN / inf deviance
    const bool done = it > 0 && __builtin_fabs(dev - dev_prev) / (__builtin_fabs(dev) + 0.1) < IRLS_EPS;
    __syncthreads();
    // ---- 2. runs: X'WX and X'Wz over all rows; rows past n inside the last group of 4 enter with weight 0 ----
    double acc[RUN];
#pragma unroll
    for (int k = 0; k < RUN; ++k) acc[k] = 0.0;
    {
      const double* src = wk + (rrhs ? n_pad : 0u);
      const float* ca = col0 + (size_t)ra * n_pad;
      const float* cb = col0 + (size_t)rb0 * n_pad;
      // columns past the run's end alias its first column (their sums are dropped), so every read stays inside the design
      const uint32_t st1 = rlen > 1u ? n_pad : 0u, st2 = rlen > 2u ? 2u * n_pad : 0u;
      for (uint32_t i = 0; i < n; i += 4u) {
        const double2 s01 = *(const double2*)(src + i), s23 = *(const double2*)(src + i + 2u);
        const float4 fa = *(const float4*)(ca + i);
        const double wx[4] = {s01.x * (double)fa.x, s01.y * (double)fa.y, s23.x * (double)fa.z, s23.y * (double)fa.w};
        const float4 f0 = *(const float4*)(cb + i), f1 = *(const float4*)(cb + st1 + i), f2 = *(const float4*)(cb + st2 + i);
        acc[0] = __builtin_fma(wx[0], (double)f0.x, acc[0]);
        acc[1] = __builtin_fma(wx[0], (double)f1.x, acc[1]);
        acc[2] = __builtin_fma(wx[0], (double)f2.x, acc[2]);
        acc[0] = __builtin_fma(wx[1], (double)f0.y, acc[0]);
        acc[1] = __builtin_fma(wx[1], (double)f1.y, acc[1]);
        acc[2] = __builtin_fma(wx[1], (double)f2.y, acc[2]);
        acc[0] = __builtin_fma(wx[2], (double)f0.z, acc[0]);
        acc[1] = __builtin_fma(wx[2], (double)f1.z, acc[1]);
        acc[2] = __builtin_fma(wx[2], (double)f2.z, acc[2]);
        acc[0] = __builtin_fma(wx[3], (double)f0.w, acc[0]);
        acc[1] = __builtin_fma(wx[3], (double)f1.w, acc[1]);
        acc[2] = __builtin_fma(wx[3], (double)f2.w, acc[2]);
      }
    }
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      if ((uint32_t)k < rlen) {
        const uint32_t bcol = rb0 + (uint32_t)k;
        double v = acc[k];
        if (MASKED) {  // dropped columns: identity rows / columns of X'WX, zeros of X'Wz
          const bool ka = (keep >> ra) & 1u, kb = (keep >> bcol) & 1u;
          v = rrhs ? (kb ? v : 0.0) : ((ka && kb) ? v : (ra == bcol ? 1.0 : 0.0));
        }
        am[rrhs ? (uint32_t)TRI + bcol : ra * (ra + 1u) / 2u + bcol] = v;
      }
    }
    __syncthreads();
    // ---- 3. Cholesky in LDS; off-diagonal slots end as L(i, j), the diagonal slot as 1 / L(j, j) ----
    bool ok = true;
#pragma unroll 1
    for (uint32_t j = 0; j < (uint32_t)P; ++j) {
      const double dd = MCP_TRI(j, j);
      ok = ok && (dd > 0.0) && (dd < __builtin_inf());
      const double inv_ljj = 1.0 / __builtin_sqrt(dd);
      __syncthreads();
      if (lane > j && lane < (uint32_t)P) MCP_TRI(lane, j) = MCP_TRI(lane, j) * inv_ljj;
      if (lane == j) MCP_TRI(j, j) = inv_ljj;
      __syncthreads();
      if (!rrhs && ra > j) {
        const double laj = MCP_TRI(ra, j);
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
          const uint32_t bcol = rb0 + (uint32_t)k;
          if ((uint32_t)k < rlen && bcol > j) MCP_TRI(ra, bcol) = __builtin_fma(-laj, MCP_TRI(bcol, j), MCP_TRI(ra, bcol));
        }
      }
      __syncthreads();
    }
    if (!ok) break;
    if (done) {  // converged: lane t forms the Wald z^2 of test t, cov = (X'WX)^-1 = L^-T L^-1 of this pass
      conv = true;
      double* bl = am + TRI + P + MT;  // beta, then u = L^-1 c of each test: u_i of test t at ul[i * MT + t]
      double* ul = bl + P;
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < P; ++i) bl[i] = beta[i];
      }
      __syncthreads();
      if (lane < (uint32_t)MT) {
        const bool act = lane < nt && (!MASKED || ((keep >> prm.test_pos[lane < nt ? lane : 0u]) & 1u));
        const double* c = prm.contrast + (lane < nt ? lane : 0u) * MAX_P;
        double est = 0.0, nq = 0.0;
#pragma unroll 1
        for (uint32_t i = 0; i < (uint32_t)P; ++i) {
          const double ci = c[i];
          est = __builtin_fma(ci, bl[i], est);
          double a = ci;
#pragma unroll 4
          for (uint32_t q = 0; q < i; ++q) a = __builtin_fma(-MCP_TRI(i, q), ul[q * MT + lane], a);
          const double ui = a * MCP_TRI(i, i);
          ul[i * MT + lane] = ui;
          nq = __builtin_fma(ui, ui, nq);
        }
        am[TRI + P + lane] = (act && nq > GLM_NEAR_ZERO && nq < __builtin_inf()) ? est * est / nq : __builtin_nan("");
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < MT; ++t) tsq[t] = am[TRI + P + t];
      break;
    }
    // ---- 4. the two triangular solves, lane i owning component i: the finished component travels through scalar
    //         registers (readlane), the others subtract their multiple of it -- forward in the order t = 0, 1, ... ----
    double r = lane < (uint32_t)P ? am[TRI + lane] : 0.0, zo = 0.0;
#pragma unroll 1
    for (uint32_t t = 0; t < (uint32_t)P; ++t) {
      const double zt = readlane_f64(r, (int)t) * MCP_TRI(t, t);
      zo = lane == t ? zt : zo;
      if (lane > t && lane < (uint32_t)P) r = __builtin_fma(-MCP_TRI(lane, t), zt, r);
    }
    r = zo;
    double bo = 0.0;
#pragma unroll 1
    for (int t = P - 1; t >= 0; --t) {
      const double bt = readlane_f64(r, t) * MCP_TRI(t, t);
      bo = lane == (uint32_t)t ? bt : bo;
      if (lane < (uint32_t)t) r = __builtin_fma(-MCP_TRI(t, lane), bt, r);
    }
    if (__builtin_amdgcn_ballot_w64(lane < (uint32_t)P && !(__builtin_fabs(bo) <= BETA_CAP)) != 0ull) break;
#pragma unroll
    for (int i = 0; i < P; ++i) beta[i] = readlane_f64(bo, i);
    dev_prev = dev;
    __syncthreads();  // the next pass overwrites wk and am
  }
#undef MCP_TRI
  if (!conv) {
#pragma unroll
    for (int t = 0; t < MT; ++t) tsq[t] = __builtin_nan("");
  }
}

// LDS layout per block (one wave): design[c][row] for columns x1..x_{P-1} then y[row] (stride n_pad),
// followed by the 4-row staging area [slot][lane] of stage_group.
// RED: sparse-factor exclusion recounted from rows [0, N) at every grid point (batch.rs:1836-1853), reduced
// fits, and the separation fallback -- when attempt 0 does not converge, drop the included factor with the
// smallest minimum level count (code 2) and refit once from a cold start (:1976-2030).
// SCEN: general data path -- per-replicate correlation factor / marginals from the scenario design records and
// per-study effects (data_gen.rs:709-751, 990-1093); the fit still starts from the spec's effects.
// NT: threads sharing one replicate (64, or 256 / 512 for designs whose LDS footprint leaves a CU with fewer than two
// single-wave blocks per SIMD: the reference bench's glm_large_n, n = 5 000, is 80 KB -- ONE wave per CU at NT = 64);
// rows spread over all NT lanes, sums cross the waves through LDS.  The scenario and wide variants stay at 64.
template <int P, bool FACTORS, bool RED, bool SCEN, bool LOGIT, int NT = WAVE>
__global__ __launch_bounds__(NT, MCP_GLM_WAVES) void glm_sim_kernel(const SimParams* __restrict__ pp) {
  static_assert(NT == WAVE || (!SCEN && P <= MAX_P_WAVE), "several waves per replicate: narrow fits on the plain data path only");
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
  // wide designs (P > 8) keep a column of ones in front of the design (irls_fit_wide reads the intercept as column 0),
  // the fit's row weights w | w z (2 n_pad doubles) after the staging area, and the normal equations (triangle |
  // right-hand side | z^2 | beta | test solves) over the staging area, which is dead once the rows exist
  constexpr bool WIDE = P > MAX_P_WAVE;
  float* design = lds + (WIDE ? n_pad : 0u);  // P columns x n_pad: x1..x_{P-1}, y
  float* st = design + (size_t)P * n_pad + lane;  // staging, stride NT
  double* fit_am = (double*)(design + (size_t)P * n_pad);
  double* fit_wk = (double*)(design + (size_t)P * n_pad + (size_t)4 * (P + 1) * NT);  // NT > 64: the cross-wave exchange area
  double* red = fit_wk;
  if (WIDE) {
    for (uint32_t i = lane; i < n_pad; i += NT) lds[i] = 1.0f;
    // rows max_n .. n_pad - 1 are never generated; the 4-row steps of the fit read them (with weight 0): keep them finite
    for (uint32_t i = lane; i < (n_pad - max_n) * (uint32_t)P; i += NT) design[(size_t)(i / (n_pad - max_n)) * n_pad + max_n + i % (n_pad - max_n)] = 0.0f;
  }

  // ---- generate the replicate into LDS: lane takes Philox blocks lane, lane + 64, ... ----
  RowCtx<P> cx;
  row_ctx_init<P, FACTORS>(prm, cx);
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  for (uint32_t ctr = lane; 4u * ctr < max_n; ctr += NT) {
    stage_group<P, NT, 1, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      if (rr < max_n) {
        float x[P], y;
        build_row<P, FACTORS, NT, 1, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, 0.0, &sl);
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
    for (uint32_t i = lane; i < n; i += NT) sum_y += (double)design[(size_t)(P - 1) * n_pad + i];
    sum_y = block_allreduce_sum<NT>(sum_y, red, lane);
    const double nd = (double)n;
    const double pbar = sum_y / nd;
    double dev_null = (pbar > 0.0 && pbar < 1.0) ? -2.0 * (sum_y * log(pbar) + (nd - sum_y) * log(1.0 - pbar)) : 0.0;
    if (!LOGIT && prm.glm_family == 2u) {  // Poisson: the intercept-only fit has mu = mean(y)
      double dn = 0.0;
      for (uint32_t i = lane; i < n; i += NT) {
        const double yi = (double)design[(size_t)(P - 1) * n_pad + i];
        dn += 2.0 * (((yi > 0.0 && pbar > 0.0) ? yi * log(yi / pbar) : 0.0) - (yi - pbar));
      }
      dev_null = block_allreduce_sum<NT>(dn, red, lane);
    }

    constexpr int MT = P > MAX_P_WAVE ? MAX_T : MAX_T_WAVE;  // tests per fit: <= P - 1 marginal targets
    double beta[P], tsq[MT];  // tsq: written by irls_fit on every exit (NaN unless the fit converged and the test is kept)
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
              for (uint32_t i = lane; i < n; i += NT) {
                uint32_t lv = 0;
#pragma unroll
                for (int c = 1; c < P; ++c)
                  if (prm.col_factor[c] == (uint32_t)f && design[(size_t)(c - 1) * n_pad + i] == 1.0f) lv = prm.col_level[c];
                cnt += lv == l ? 1u : 0u;
              }
              if (NT > WAVE) {
                cnt = (uint32_t)block_allreduce_sum<NT>((double)cnt, red, lane);  // per-lane counts -> block total (exact in f64)
              } else {
#pragma unroll
                for (int m = 1; m < WAVE; m <<= 1) cnt += __shfl_xor(cnt, m, WAVE);
              }
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
        if (P > MAX_P_WAVE)
          irls_fit_wide<P, true, LOGIT, MT>(prm, design, fit_wk, fit_am, n_pad, n, lane, keep, prm.glm_family, nt, beta, tsq, dev, conv);
        else
          irls_fit<P, true, LOGIT, MT, NT>(prm, design, n_pad, n, lane, keep, prm.glm_family, nt, beta, tsq, dev, conv, red);
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
      if (P > MAX_P_WAVE)
        irls_fit_wide<P, false, LOGIT, MT>(prm, design, fit_wk, fit_am, n_pad, n, lane, keep, prm.glm_family, nt, beta, tsq, dev, conv);
      else
        irls_fit<P, false, LOGIT, MT, NT>(prm, design, n_pad, n, lane, keep, prm.glm_family, nt, beta, tsq, dev, conv, red);
    }

    // ---- decisions on the Wald z^2 the converged pass left in tsq (NaN otherwise), LRT overall ----
    const double* crow = prm.crit + (size_t)g * prm.crit_stride;
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
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
      prm.conv[cell] = conv ? 1 : 0;
      if (prm.overall) prm.overall[cell] = ov ? 1 : 0;
      if (prm.stat && g == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
          if ((uint32_t)t < nt) prm.stat[(size_t)out_row * nt + t] = tsq[t];
      }
      if (RED && prm.factor_excluded)
        for (uint32_t f = 0; f < prm.n_factors; ++f)
          prm.factor_excluded[cell * prm.n_factors + f] = (uint8_t)((codes >> (2 * f)) & 3u);
    }
  }
}

}  // namespace mcp
