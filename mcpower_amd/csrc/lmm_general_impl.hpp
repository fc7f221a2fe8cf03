// lmm_general_impl.hpp -- gfx950 kernel of the GENERAL linear mixed model (REML), included by the two translation
// units that instantiate it (lmm_general_kernels.hip: plain data path; lmm_general_scen_kernels.hip: general path): random slopes on the
// primary grouping, crossed / nested extra groupings, slopes on those.
//
// Reference: crates/engine-core/src/batch.rs:1214-1560 (Mle arm), data_gen.rs:146-284 (draw order of the random
// effects on the sequential stream: primary intercepts, primary slopes, all extra intercepts, all extra slopes),
// :812-842 (latent-scale contribution), engine-contract/src/generation.rs:297-427 (row -> level maps),
// mixed_workspace.rs:57-198 (model terms, theta layout, truth start).  The solver behind `fit_on` (glmm 0.1.2 +
// bobyqa 0.1.1) is out of tree; this is lme4's penalised least squares (Bates et al. 2015, eqs. 39-41) minimised
// by a bounded Nelder-Mead simplex -- the same algorithm, constants and operation order as oracle/fit_lmm_general.c.
//
// ONE WAVEFRONT PER REPLICATE, everything resident in LDS:
//   1. the random effects are drawn (lane = level), the n x p design and the outcome are generated (lane = row);
//   2. rows are folded into the theta-free sufficient statistics in row order (lanes = the <= 15 x 15 products of
//      one sparse row, so the summation order -- hence every bit -- is fixed):  per primary level c the q0 x q0
//      block Z0'Z0 and the q0 x W strip Z0'[Zr | X | y], plus the W x W Gram matrix of [Zr | X | y]
//      (Zr = columns of the extra groupings, W = R + p + 1);
//   3. per grid N the criterion is minimised.  One evaluation = block Cholesky in the order [primary blocks | rest |
//      X | y]: the primary blocks are eliminated one level at a time (only the columns that level actually touches:
//      a nested child touches one parent, so its strip is short), then a left-looking dense Cholesky (lane = row)
//      runs over the rest and X columns with y riding along as the right-hand side.
// Wave-uniform control flow throughout: all 64 lanes carry the same simplex and take the same branches.
#pragma once
#ifndef MCP_LMG_WAVES
#define MCP_LMG_WAVES 3  // waves per SIMD the register allocator leaves room for: 3 is +20% on slope designs (profiles/r01_j_occupancy.txt)
#endif
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_scenario.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"

#pragma clang fp contract(off)

namespace mcp {
namespace {

constexpr int GW = 64;
constexpr double LMG_THETA_ZERO = 1e-6;
constexpr double LMG_THETA_MAX = 32.0;
constexpr uint32_t LMG_MAX_EVAL = 6000;
constexpr double LMG_NEAR_ZERO = 1e-30;
constexpr int LMG_MAX_P = 12;  // design columns: 9..12 with random intercepts only (one block width, see the launchers)
constexpr int LMG_MAX_NZ = 2 * LMG_MAX_SLOPES + 2 + LMG_MAX_P + 1;  // sparse row of [Zr | X | y]
// row stride of the p x p factor and of the per-test solves in the tail block: 8 up to 8 design columns, 16 beyond
__device__ __forceinline__ uint32_t lmg_tail_stride(uint32_t p) { return p > 8u ? 16u : 8u; }

__device__ __forceinline__ void lmg_sync() { __syncthreads(); }  // one wave per block

// extra_level_of_row (generation.rs:402-427)
__device__ __forceinline__ uint32_t lmg_level_of_row(const SimParams& prm, uint32_t t, uint32_t i) {
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

// LDS-qualified pointers: 32-bit, ds_* instructions, and provably disjoint from private memory -- with generic
// pointers every access went through flat instructions and the struct of pointers was re-read from scratch
// after each store (measured: 10x slower)
typedef __attribute__((address_space(3))) double ldd;
typedef __attribute__((address_space(3))) float ldf;
typedef __attribute__((address_space(3))) uint32_t ldu;

// packed lower triangle, by rows: element (i, j <= i); lmg_tri(W, 0) = W (W + 1) / 2 is the size of a W x W triangle
__device__ __forceinline__ size_t lmg_tri(uint32_t i, uint32_t j) { return (size_t)i * (i + 1u) / 2u + j; }

// LDS map of one replicate (doubles first, then floats)
struct LmgLds {
  ldd *gzz, *gzw, *gww, *S, *B, *ld, *simplex, *fv, *tm, *misc, *tail, *carry;
  ldu* ridx;
  ldd* rval;
  ldf *design, *u0, *us, *ux;
  float* st;  // staging of the row generator (generic: shared with mcp_rows.hpp)
};

// fit view (mcp_params.hpp): the term that owns column i < fR of the dense rest block
__device__ __forceinline__ uint32_t lmg_rest_term(const SimParams& prm, uint32_t i) {
  const uint32_t t0 = prm.lmg_frest[0], t1 = prm.lmg_frest[1];  // at most two terms besides the eliminated one
  return (prm.lmg_n_rest > 1u && i >= prm.lmg_froff[t1]) ? t1 : t0;
}

// (i, j <= i) of entry e of a packed lower triangle (lmg_tri(i, j) = e): the loops over the triangle hand one ENTRY to a lane --
// row by row with lanes over the columns, a W x W triangle took W serial trips of at most W / 64 busy waves each
__device__ __forceinline__ void lmg_tri_decode(uint32_t e, uint32_t& i, uint32_t& j) {
  uint32_t r = (uint32_t)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  while (r * (r + 1u) / 2u > e) --r;
  while ((r + 1u) * (r + 2u) / 2u <= e) ++r;
  i = r;
  j = e - r * (r + 1u) / 2u;
}

// compact position of column i of [Zr | X | y] inside an eliminated level's strip (mcp_params.hpp, lmg_compact): the columns of
// a grouped term are counted within the level's own primary level -- the caller only pairs a level with columns it meets
// (`compact` is the caller's register copy of prm.lmg_compact: read through prm at every call it was a scalar load inside the
// per-row fold and the Schur loops -- 11 % of lmm_multislope at two waves per SIMD)
__device__ __forceinline__ uint32_t lmg_cpos(const SimParams& prm, uint32_t compact, uint32_t i) {
  if (!compact) return i;
  if (i >= prm.lmg_fR) return prm.lmg_coff_xy + (i - prm.lmg_fR);
  const uint32_t t = lmg_rest_term(prm, i), o = i - prm.lmg_froff[t];
  return prm.lmg_coff[t] + (prm.lmg_frel[t] == 1u ? o % (prm.lmg_fcnt[t] * prm.lmg_q[t]) : o);
}

// template entries T_t[r][c] of every term from theta (column-major vech per term, mixed_workspace.rs:189-196)
__device__ __forceinline__ void lmg_templates(const SimParams& prm, const ldd* th, ldd* tm, uint32_t lane) {
  if (lane < 27u) {
    const uint32_t t = lane / 9u, r = (lane % 9u) / 3u, c = lane % 3u;
    double v = 0.0;
    if (t < prm.lmg_n_terms && r < prm.lmg_q[t] && c <= r) {
      const uint32_t q = prm.lmg_q[t];
      // column-major lower vech: entries of columns 0..c-1 come first
      uint32_t k = prm.lmg_theta_off[t];
      for (uint32_t cc = 0; cc < c; ++cc) k += q - cc;
      v = th[k + (r - c)];
    }
    tm[lane] = v;
  }
}

// one evaluation of the profiled REML criterion at theta `th` (LDS, n_theta doubles).  Returns the deviance (inf
// when a pivot fails); leaves sigma^2 in misc[0], beta in tail[0..p), the lower factor of RX'RX in tail[ts + a*ts + b], ts = lmg_tail_stride(p).
template <int Q0>
__device__ __forceinline__ double lmg_eval(const SimParams& prm, const LmgLds m, const ldd* th, uint32_t lane, double n_rows) {
  const uint32_t W = prm.lmg_fW, R = prm.lmg_fR, p = prm.P, E = prm.lmg_elim, L0 = prm.lmg_levels[E], cpe = prm.lmg_cpe;
  const uint32_t Wc = prm.lmg_Wc, compact = prm.lmg_compact;  // stride of the strips (gzw, B): W, or the columns a level meets when compact
  const double inf = __builtin_inf();
  lmg_templates(prm, th, m.tm, lane);
  lmg_sync();
  // ---- S (lower triangle) = Lambda~' Gww Lambda~ (+ I on the rest block), Lambda~ = diag(Lambda_rest, I): lane = entry ----
  for (uint32_t en = lane; en < (uint32_t)lmg_tri(W, 0u); en += GW) {
    uint32_t i, j;
    lmg_tri_decode(en, i, j);
    uint32_t ti = 0, bi = 0, qi = 1, si = i;  // block of the row index
    if (i < R) {
      ti = lmg_rest_term(prm, i);
      qi = prm.lmg_q[ti];
      const uint32_t rel = i - prm.lmg_froff[ti];
      bi = rel % qi;
      si = i - bi;
    }
    uint32_t tj = 0, bj = 0, qj = 1, sj = j;  // ... and of the column index
    if (j < R) {
      tj = lmg_rest_term(prm, j);
      qj = prm.lmg_q[tj];
      const uint32_t rel = j - prm.lmg_froff[tj];
      bj = rel % qj;
      sj = j - bj;
    }
    // the oracle applies Lambda' along the row index first, then along the column index: keep that association
    double acc = 0.0;
    for (uint32_t e = bj; e < qj; ++e) {
      double inner = 0.0;
      for (uint32_t d = bi; d < qi; ++d) {
        const uint32_t a = si + d, b = sj + e;
        const double g = a >= b ? m.gww[lmg_tri(a, b)] : m.gww[lmg_tri(b, a)];
        inner = i < R ? inner + m.tm[ti * 9u + d * 3u + bi] * g : g;
      }
      acc = j < R ? acc + m.tm[tj * 9u + e * 3u + bj] * inner : inner;
    }
    if (i == j && i < R) acc += 1.0;
    m.S[en] = acc;
  }
  lmg_sync();
  // ---- eliminate the primary blocks: lane = level ----
  // Each lane factors its level's q0 x q0 block and forward-solves the level's strip B_c = Lc^-1 T0' [Gzw Lambda~]
  // (q0 x W, zeros where the level has no rows) into LDS; then S -= sum_k B[k]' B[k] runs as a small matrix product
  // with lanes over the (i, j) pairs and k = (level, a) ascending -- the oracle's summation order.
  double logdet = 0.0, pd = 1.0;
  uint32_t npd = 0;
  bool ok = true;
  double T0[Q0][Q0];
#pragma unroll
  for (int r = 0; r < Q0; ++r)
#pragma unroll
    for (int c = 0; c < Q0; ++c) T0[r][c] = m.tm[E * 9u + r * 3 + c];
  for (uint32_t c = lane; c < L0; c += GW) {
    const ldd* G = m.gzz + (size_t)c * Q0 * Q0;
    double A[Q0][Q0], Lc[Q0][Q0], pdc = 1.0;
#pragma unroll
    for (int a = 0; a < Q0; ++a)
#pragma unroll
      for (int b = 0; b < Q0; ++b) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < Q0; ++r)
#pragma unroll
          for (int cc = 0; cc < Q0; ++cc)
            if (r >= a && cc >= b) s += T0[r][a] * G[r * Q0 + cc] * T0[cc][b];
        A[a][b] = s + (a == b ? 1.0 : 0.0);
      }
#pragma unroll
    for (int j = 0; j < Q0; ++j) {
      double d = A[j][j];
#pragma unroll
      for (int k = 0; k < Q0; ++k)
        if (k < j) d -= Lc[j][k] * Lc[j][k];
      ok = ok && (d > 0.0);
      Lc[j][j] = __builtin_sqrt(d);
      pdc *= d;
#pragma unroll
      for (int i = 0; i < Q0; ++i)
        if (i > j) {
          double v = A[i][j];
#pragma unroll
          for (int k = 0; k < Q0; ++k)
            if (k < j) v -= Lc[i][k] * Lc[j][k];
          Lc[i][j] = v / Lc[j][j];
        }
    }
    m.ld[c] = log(pdc);
    // columns this level can meet: every crossed column, the columns of its own primary level (nested children; the
    // parent itself when a nested grouping is the eliminated one), X and y
    for (uint32_t rj = 0; rj <= prm.lmg_n_rest; ++rj) {
      const bool xy = rj == prm.lmg_n_rest;
      const uint32_t tj = xy ? 0u : prm.lmg_frest[rj];
      const uint32_t qj = xy ? 1u : prm.lmg_q[tj];
      uint32_t jlo, jhi;
      if (xy) { jlo = R; jhi = W; }
      else if (prm.lmg_frel[tj] == 1u) { jlo = prm.lmg_froff[tj] + (cpe == 1u ? c : c / cpe) * prm.lmg_fcnt[tj] * qj; jhi = jlo + prm.lmg_fcnt[tj] * qj; }
      else { jlo = prm.lmg_froff[tj]; jhi = jlo + prm.lmg_levels[tj] * qj; }
      // compact strips: the columns [jlo, jhi) sit at [coff, ..) of the level's strip
      const uint32_t cshift = compact ? jlo - (xy ? prm.lmg_coff_xy : prm.lmg_coff[tj]) : 0u;
      for (uint32_t sj = jlo; sj < jhi; sj += qj) {
        for (uint32_t bj = 0; bj < qj; ++bj) {
          const uint32_t j = sj + bj;
          double zw[Q0];
#pragma unroll
          for (int r = 0; r < Q0; ++r) {
            const ldd* row = m.gzw + ((size_t)c * Q0 + r) * Wc - cshift;
            double sacc = 0.0;
            if (!xy) {
              for (uint32_t d = bj; d < qj; ++d) sacc += m.tm[tj * 9u + d * 3u + bj] * row[sj + d];
            } else {
              sacc = row[j];
            }
            zw[r] = sacc;
          }
          double bcol[Q0];
#pragma unroll
          for (int a = 0; a < Q0; ++a) {
            double sacc = 0.0;
#pragma unroll
            for (int r = 0; r < Q0; ++r)
              if (r >= a) sacc += T0[r][a] * zw[r];
#pragma unroll
            for (int k = 0; k < Q0; ++k)
              if (k < a) sacc -= Lc[a][k] * bcol[k];
            bcol[a] = sacc / Lc[a][a];
            m.B[((size_t)c * Q0 + a) * Wc + (j - cshift)] = bcol[a];
          }
        }
      }
    }
  }
  ok = __ballot(!ok) == 0ull;
  lmg_sync();
  for (uint32_t c = 0; c < L0; ++c) logdet += m.ld[c];  // level order, as the oracle sums
  {
    const uint32_t KT = L0 * Q0;
    // S -= sum_k B[k]' B[k], k = (level, a) ascending -- the oracle's summation order per entry; lane = entry
    for (uint32_t en = lane; en < (uint32_t)lmg_tri(W, 0u); en += GW) {
      uint32_t i, j;
      lmg_tri_decode(en, i, j);
      // a column tied to one primary level only meets the rows of B of that level's eliminated blocks
      uint32_t pi = 0xffffffffu, pj = 0xffffffffu;
      if (i < R) {
        const uint32_t ti = lmg_rest_term(prm, i);
        if (prm.lmg_frel[ti] == 1u) pi = ((i - prm.lmg_froff[ti]) / prm.lmg_q[ti]) / prm.lmg_fcnt[ti];
      }
      if (j < R) {
        const uint32_t tj = lmg_rest_term(prm, j);
        if (prm.lmg_frel[tj] == 1u) pj = ((j - prm.lmg_froff[tj]) / prm.lmg_q[tj]) / prm.lmg_fcnt[tj];
      }
      uint32_t klo = 0, khi = KT;
      if (pi != 0xffffffffu) { klo = pi * cpe * Q0; khi = klo + cpe * Q0; }
      if (pj != 0xffffffffu) {
        if (pi != 0xffffffffu && pi != pj) khi = klo;  // columns of different primary levels never share a row
        else { klo = pj * cpe * Q0; khi = klo + cpe * Q0; }
      }
      khi = khi < KT ? khi : KT;
      const uint32_t ci2 = lmg_cpos(prm, compact, i), cj2 = lmg_cpos(prm, compact, j);
      double sacc = m.S[en];
      for (uint32_t k = klo; k < khi; ++k) sacc -= m.B[(size_t)k * Wc + ci2] * m.B[(size_t)k * Wc + cj2];
      m.S[en] = sacc;
    }
  }
  lmg_sync();
  // ---- Cholesky over [rest | X], y is the last row: right-looking, one ENTRY of the trailing triangle per lane.  Every entry
  //      still loses its pivot products in ascending pivot order and is divided last -- the operation sequence of the left-looking
  //      form (lane = row, a k-loop of length j per column: W^2 / 2 dependent steps) with W (W + 1) / 128 trips per column ----
  const uint32_t M = R + p;
  for (uint32_t j = 0; j < M; ++j) {
    const double d = m.S[lmg_tri(j, j)];  // finished: every earlier column has been subtracted
    ok = ok && (d > 0.0) && (d < inf);
    const double ljj = __builtin_sqrt(d);
    pd *= d;
    if (++npd == 4u) { logdet += log(pd); pd = 1.0; npd = 0; }
    lmg_sync();  // every lane holds the pivot
    for (uint32_t i = j + lane; i < W; i += GW) m.S[lmg_tri(i, j)] = i == j ? ljj : m.S[lmg_tri(i, j)] / ljj;
    lmg_sync();
    const uint32_t T = W - j - 1u;  // rows j + 1 .. W - 1
    for (uint32_t en = lane; en < (uint32_t)lmg_tri(T, 0u); en += GW) {
      uint32_t a, b;
      lmg_tri_decode(en, a, b);
      const uint32_t i = j + 1u + a, c = j + 1u + b;
      m.S[lmg_tri(i, c)] -= m.S[lmg_tri(i, j)] * m.S[lmg_tri(c, j)];
    }
    lmg_sync();
  }
  logdet += log(pd);
  const double pwrss = m.S[lmg_tri(M, M)];  // y'y minus the squares of the y row, in column order
  const double dfr = n_rows - (double)p;
  double lmin = inf, lmax = 0.0;
  for (uint32_t a = 0; a < p; ++a) {
    const double v = __builtin_fabs(m.S[lmg_tri((R + a), R + a)]);
    lmin = v < lmin ? v : lmin;
    lmax = v > lmax ? v : lmax;
  }
  ok = ok && (pwrss > 0.0) && (dfr > 0.0) && (lmin / lmax >= 1e-12);
  if (!ok) return inf;
  const double dev = logdet + dfr * (1.0 + log(2.0 * 3.14159265358979323846 * pwrss / dfr));
  const uint32_t ts = lmg_tail_stride(p);
  if (lane == 0u) {
    m.misc[0] = pwrss / dfr;
    for (uint32_t a = 0; a < p; ++a)
      for (uint32_t b = 0; b < p; ++b) m.tail[ts + a * ts + b] = b <= a ? m.S[lmg_tri((R + a), R + b)] : 0.0;
    for (int a = (int)p - 1; a >= 0; --a) {
      double v = m.S[lmg_tri(M, R + (uint32_t)a)];
      for (uint32_t k = (uint32_t)a + 1u; k < p; ++k) v -= m.tail[ts + k * ts + (uint32_t)a] * m.tail[k];
      m.tail[a] = v / m.tail[ts + (uint32_t)a * ts + (uint32_t)a];
    }
  }
  lmg_sync();
  return (dev < inf && dev > -inf) ? dev : inf;
}


// A reduced round (batch.rs:1343-1400): some design columns are out of the fit (sparse-factor exclusion, a test_formula
// subset) and every grouping keeps its intercept only.  The fitted model's statistics are sub-blocks of the full
// structure's: the intercept row of each primary strip, the intercept column of each extra-grouping level, the kept X
// columns.  theta has one entry per grouping; a dropped X column is an identity row / column of the dense block.
struct LmgRed {
  uint32_t keep;   // kept design columns
  uint32_t p_fit;  // their count
};

// view column (reduced rest | X | y) -> column of the full layout
__device__ __forceinline__ uint32_t lmg_red_col(const SimParams& prm, uint32_t i, uint32_t Rr, uint32_t& term) {
  term = 0u;
  if (i >= Rr) return prm.lmg_fR + (i - Rr);
  uint32_t r = 0u, off = 0u;
  while (r + 1u < prm.lmg_n_rest && i >= off + prm.lmg_levels[prm.lmg_frest[r]]) { off += prm.lmg_levels[prm.lmg_frest[r]]; ++r; }
  const uint32_t t = prm.lmg_frest[r];
  term = t;
  return prm.lmg_froff[t] + (i - off) * prm.lmg_q[t];
}

template <int Q0F>
__device__ __forceinline__ double lmg_eval_red(const SimParams& prm, const LmgLds m, const ldd* th, uint32_t lane, double n_rows,
                                            const LmgRed red) {
  const uint32_t Wf = prm.lmg_fW, p = prm.P, E = prm.lmg_elim, L0 = prm.lmg_levels[E];
  uint32_t Rr = 0;
  for (uint32_t r = 0; r < prm.lmg_n_rest; ++r) Rr += prm.lmg_levels[prm.lmg_frest[r]];
  const uint32_t W = Rr + p + 1u, M = Rr + p;
  const double inf = __builtin_inf();
  auto theta_of = [&](uint32_t term) { return th[term]; };
  auto dropped = [&](uint32_t i) { return i >= Rr && i < M && !((red.keep >> (i - Rr)) & 1u); };
  // ---- S (lower triangle): theta_tj (theta_ti G) on the rest block + I, pass-through elsewhere, identity for dropped X ----
  for (uint32_t en = lane; en < (uint32_t)lmg_tri(W, 0u); en += GW) {  // lane = entry (lmg_tri_decode)
    uint32_t i, j, ti, tj;
    lmg_tri_decode(en, i, j);
    const uint32_t fi = lmg_red_col(prm, i, Rr, ti);
    const uint32_t fj = lmg_red_col(prm, j, Rr, tj);
    double acc;
    if (dropped(i) || dropped(j)) {
      acc = i == j ? 1.0 : 0.0;
    } else {
      const double g = fi >= fj ? m.gww[lmg_tri(fi, fj)] : m.gww[lmg_tri(fj, fi)];
      const double inner = i < Rr ? 0.0 + theta_of(ti) * g : g;
      acc = j < Rr ? 0.0 + theta_of(tj) * inner : inner;
      if (i == j && i < Rr) acc += 1.0;
    }
    m.S[en] = acc;
  }
  lmg_sync();
  // ---- primary blocks (scalars): lane = level ----
  double logdet = 0.0, pd = 1.0;
  uint32_t npd = 0;
  bool ok = true;
  const double t0 = th[E];
  for (uint32_t c = lane; c < L0; c += GW) {
    const double g = m.gzz[(size_t)c * Q0F * Q0F];
    const double dd = 0.0 + t0 * g * t0 + 1.0;
    ok = ok && (dd > 0.0);
    const double lc = __builtin_sqrt(dd);
    m.ld[c] = log(dd);
    const ldd* row = m.gzw + (size_t)c * Q0F * Wf;  // intercept row of the level's strip
    for (uint32_t j = 0; j < W; ++j) {
      uint32_t tj;
      const uint32_t fj = lmg_red_col(prm, j, Rr, tj);
      const double zw = j < Rr ? 0.0 + theta_of(tj) * row[fj] : row[fj];
      const double tv = 0.0 + t0 * zw;
      m.B[(size_t)c * W + j] = dropped(j) ? 0.0 : tv / lc;
    }
  }
  ok = __ballot(!ok) == 0ull;
  lmg_sync();
  for (uint32_t c = 0; c < L0; ++c) logdet += m.ld[c];
  for (uint32_t en = lane; en < (uint32_t)lmg_tri(W, 0u); en += GW) {
    uint32_t i, j;
    lmg_tri_decode(en, i, j);
    double sacc = m.S[en];
    for (uint32_t k = 0; k < L0; ++k) sacc -= m.B[(size_t)k * W + i] * m.B[(size_t)k * W + j];
    m.S[en] = sacc;
  }
  lmg_sync();
  // ---- dense Cholesky over [rest | X] (right-looking, lane = entry, as lmg_eval); a dropped column is an identity: its
  //      products are exact zeros and its pivot stays out of the determinant ----
  for (uint32_t j = 0; j < M; ++j) {
    const double dj = m.S[lmg_tri(j, j)];
    ok = ok && (dj > 0.0) && (dj < inf);
    const double ljj = __builtin_sqrt(dj);
    if (!dropped(j)) {
      pd *= dj;
      if (++npd == 4u) { logdet += log(pd); pd = 1.0; npd = 0; }
    }
    lmg_sync();
    for (uint32_t i = j + lane; i < W; i += GW) m.S[lmg_tri(i, j)] = i == j ? ljj : m.S[lmg_tri(i, j)] / ljj;
    lmg_sync();
    const uint32_t T = W - j - 1u;
    for (uint32_t en = lane; en < (uint32_t)lmg_tri(T, 0u); en += GW) {
      uint32_t a, b;
      lmg_tri_decode(en, a, b);
      const uint32_t i = j + 1u + a, c = j + 1u + b;
      m.S[lmg_tri(i, c)] -= m.S[lmg_tri(i, j)] * m.S[lmg_tri(c, j)];
    }
    lmg_sync();
  }
  logdet += log(pd);
  const double pwrss = m.S[lmg_tri(M, M)];
  const double dfr = n_rows - (double)red.p_fit;
  double lmin = inf, lmax = 0.0;
  for (uint32_t a = 0; a < p; ++a) {
    if (!((red.keep >> a) & 1u)) continue;
    const double v = __builtin_fabs(m.S[lmg_tri((Rr + a), Rr + a)]);
    lmin = v < lmin ? v : lmin;
    lmax = v > lmax ? v : lmax;
  }
  ok = ok && (pwrss > 0.0) && (dfr > 0.0) && (lmin / lmax >= 1e-12);
  if (!ok) return inf;
  const double dev = logdet + dfr * (1.0 + log(2.0 * 3.14159265358979323846 * pwrss / dfr));
  const uint32_t ts = lmg_tail_stride(p);
  if (lane == 0u) {
    m.misc[0] = pwrss / dfr;
    for (uint32_t a = 0; a < p; ++a)
      for (uint32_t b = 0; b < p; ++b) m.tail[ts + a * ts + b] = b <= a ? m.S[lmg_tri((Rr + a), Rr + b)] : 0.0;
    for (int a = (int)p - 1; a >= 0; --a) {
      double v = m.S[lmg_tri(M, Rr + (uint32_t)a)];
      for (uint32_t k = (uint32_t)a + 1u; k < p; ++k) v -= m.tail[ts + k * ts + (uint32_t)a] * m.tail[k];
      m.tail[a] = v / m.tail[ts + (uint32_t)a * ts + (uint32_t)a];
    }
  }
  lmg_sync();
  return (dev < inf && dev > -inf) ? dev : inf;
}

__device__ __forceinline__ bool lmg_is_diag(const SimParams& prm, uint32_t k) {
  for (uint32_t t = 0; t < prm.lmg_n_terms; ++t) {
    uint32_t o = prm.lmg_theta_off[t];
    const uint32_t q = prm.lmg_q[t];
    for (uint32_t c = 0; c < q; ++c) {
      if (k == o) return true;
      o += q - c;
    }
  }
  return false;
}

template <bool RED>
__device__ __forceinline__ void lmg_clamp(const SimParams& prm, ldd* x, uint32_t lane) {
  const uint32_t d = RED ? prm.lmg_n_terms : prm.lmg_n_theta;
  if (lane < d) {
    double v = x[lane];
    if (RED || lmg_is_diag(prm, lane)) v = v < 0.0 ? 0.0 : (v > LMG_THETA_MAX ? LMG_THETA_MAX : v);
    else v = v < -LMG_THETA_MAX ? -LMG_THETA_MAX : (v > LMG_THETA_MAX ? LMG_THETA_MAX : v);
    x[lane] = v;
  }
  lmg_sync();
}

#include "mcp_stationary.hpp"  // MCP_STAT_H / MCP_STAT_TOL
#include "mcp_fit_fsm.hpp"

// SCEN: general data path (scenario design records, non-normal marginals / residuals, per-study effects,
// heteroskedastic residuals) plus the LME knobs -- random-effect shape, one ICC jitter per grouping
template <int P, int Q0, bool SCEN>
// (three random effects on the eliminated grouping, nine or more design columns and the general data path from seven columns on get 256
// registers instead of 168, or the block algebra spills -- 20-108 B; the two widest general-path variants take all 512.  Those designs
// are LDS-bound below 3 waves per SIMD anyway)
__global__ __launch_bounds__(GW, ((SCEN && P >= 11) ? 1 : (Q0 >= 3 || P >= 9 || (SCEN && P >= 7)) ? 2 : MCP_LMG_WAVES)) void lmm_general_kernel(const SimParams* __restrict__ pp) {
  extern __shared__ double lds_d[];
  const SimParams& prm = *pp;
  const uint32_t lane = threadIdx.x;
  const uint32_t lsim = blockIdx.x;
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t max_n = prm.max_n, n_pad = (max_n + 3u) & ~3u;
  const uint32_t W = prm.lmg_fW, R = prm.lmg_fR, L0 = prm.lmg_levels[0], d = prm.lmg_n_theta, nterms = prm.lmg_n_terms;
  const uint32_t E = prm.lmg_elim, LE = prm.lmg_levels[E];  // fit view (mcp_params.hpp): the grouping eliminated block by block
  const uint32_t Wc = prm.lmg_Wc, compact = prm.lmg_compact;  // strips of the eliminated levels: stride, compact positions
  LmgLds m;
  {
    // persistent: the sufficient statistics, the sparse-row scratch of their accumulation, the theta carry, the design
    ldd* q = (ldd*)lds_d;
    m.gzz = q; q += (size_t)LE * Q0 * Q0;
    m.gzw = q; q += (size_t)LE * Q0 * Wc;
    m.gww = q; q += lmg_tri(W, 0u);  // lower triangles, packed by rows
    m.rval = q; q += LMG_MAX_NZ;
    m.carry = q; q += LMG_MAX_THETA;
    ldu* u = (ldu*)q;
    m.ridx = u; u += (LMG_MAX_NZ + 1) & ~1;
    ldf* f = (ldf*)u;
    m.design = f; f += (size_t)P * (prm.lmg_stream ? 256u : n_pad);
    // one region, two lives: the generator's random effects and staging area die when the last row exists; the
    // evaluation's matrices and simplex are first written by the first evaluation (the LDS footprint bounds the
    // occupancy of this latency-bound kernel: 24 -> 18 KB on the lmm_slope shape)
    ldd* e = (ldd*)f;
    m.S = e; e += lmg_tri(W, 0u);
    m.B = e; e += (size_t)LE * Q0 * Wc;
    m.ld = e; e += LE;
    m.simplex = e; e += (size_t)(d + 4u) * LMG_MAX_THETA;
    m.fv = e; e += d + 1u;
    m.tm = e; e += 27;
    m.misc = e; e += 8;
    constexpr uint32_t TS = P > 8 ? 16u : 8u;
    constexpr uint32_t MTL = P > 8 ? 12u : 8u, TCL = P > 8 ? 16u : 8u;  // tests per fit, stride of their joint factor
    m.tail = e; e += TS + TS * TS + MTL * TS + TCL * TCL + TCL + 8;  // beta | L | U | Lc | cj | spare
    m.u0 = f; f += L0;
    m.us = f; f += (size_t)L0 * LMG_MAX_SLOPES;
    m.ux = f;
    for (uint32_t t = 1; t < nterms; ++t) f += (size_t)prm.lmg_levels[t] * prm.lmg_q[t];
    m.st = (float*)f + lane;
  }
  // ---- 1. random effects on the sequential stream (data_gen.rs:146-284): draw d is the d-th block of
  //         re_draws_per_cluster words, shaped by the scenario's random_effect_dist on the general path ----
  auto re_draw = [&](uint32_t ordinal) -> double {
    if (SCEN) return draw_cluster_effect(prm, k0, k1, ordinal);
    return (double)norm_inv_cdf_f32(u32_to_unit_f32(sequential_word(k0, k1, (uint64_t)ordinal)));
  };
  // realised tau per grouping (ICC jitter, data_gen.rs:1069-1092) and the factor of its D = diag(tau) R diag(tau)
  auto term_tau = [&](uint32_t t) -> double {
    if (SCEN && prm.scen_tau_sq) {
      const double t2 = prm.scen_tau_sq[(size_t)t * prm.scen_stride + lsim];
      return __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
    }
    return prm.lmg_tau[t];
  };
  // lower factor of D_t = diag(tau) R diag(tau) (row-major q x q) into LDS: lane 0 writes l = lc[0..9) (dm = lc[9..18) is its
  // work area), every lane reads it after the barrier.  Kept out of private arrays: indexed at run time they became
  // 72-208 B of scratch per lane.  lc = m.rval .. m.carry: free until the rows are folded / the first fit has converged.
  ldd* lc = m.rval;
  auto term_chol = [&](uint32_t t, double tau0) {
    const uint32_t q = prm.lmg_q[t];
    lmg_sync();
    if (lane == 0u) {
      if (!SCEN) {
        for (uint32_t i = 0; i < 9u; ++i) lc[i] = prm.lmg_chol[t][i];
      } else {
        ldd* dm = lc + 9;
        for (uint32_t i = 0; i < 9u; ++i) lc[i] = 0.0;
        for (uint32_t i = 0; i < q; ++i)
          for (uint32_t j = 0; j < q; ++j) {
            const double ti = i == 0u ? tau0 : prm.lmg_slope_sd[t][i - 1u], tj = j == 0u ? tau0 : prm.lmg_slope_sd[t][j - 1u];
            dm[i * q + j] = ti * prm.lmg_recorr[t][i * q + j] * tj;
          }
        for (uint32_t j = 0; j < q; ++j) {  // chol_lower (data_gen.rs:55-75)
          double diag = dm[j * q + j];
          for (uint32_t k = 0; k < j; ++k) diag -= lc[j * q + k] * lc[j * q + k];
          const double ljj = __builtin_sqrt(diag > 0.0 ? diag : 0.0);
          lc[j * q + j] = ljj;
          for (uint32_t i = j + 1u; i < q; ++i)
            if (ljj > 0.0) {
              double sacc = dm[i * q + j];
              for (uint32_t k = 0; k < j; ++k) sacc -= lc[i * q + k] * lc[j * q + k];
              lc[i * q + j] = sacc / ljj;
            }
        }
      }
    }
    lmg_sync();
  };
  const double tau_p = term_tau(0u);
  for (uint32_t c = lane; c < L0; c += GW) m.u0[c] = (float)(re_draw(prm.lmg_seq[0] + c) * tau_p);
  lmg_sync();
  if (Q0 > 1) {
    term_chol(0u, tau_p);
    for (uint32_t c = lane; c < L0; c += GW) {
      double z[3];
      z[0] = tau_p > 0.0 ? (double)m.u0[c] / tau_p : 0.0;
#pragma unroll
      for (int j = 1; j < Q0; ++j) z[j] = re_draw(prm.lmg_seq[1] + c * (uint32_t)(Q0 - 1) + (uint32_t)(j - 1));
#pragma unroll
      for (int k = 1; k < Q0; ++k) {
        double uk = 0.0;
#pragma unroll
        for (int j = 0; j < Q0; ++j)
          if (j <= k) uk += lc[k * Q0 + j] * z[j];
        m.us[(size_t)c * (Q0 - 1) + (k - 1)] = (float)uk;
      }
    }
  }
  {
    ldf* ux = m.ux;
    for (uint32_t t = 1; t < nterms; ++t) {
      const uint32_t q = prm.lmg_q[t], nl = prm.lmg_levels[t];
      const double tau_t = term_tau(t);
      for (uint32_t c = lane; c < nl; c += GW) ux[(size_t)c * q] = (float)(re_draw(prm.lmg_seq[2u * t] + c) * tau_t);
      lmg_sync();
      if (q > 1u) {
        term_chol(t, tau_t);
        for (uint32_t c = lane; c < nl; c += GW) {
          double z0 = tau_t > 0.0 ? (double)ux[(size_t)c * q] / tau_t : 0.0, z1 = 0.0, z2 = 0.0;
          z1 = re_draw(prm.lmg_seq[2u * t + 1u] + c * (q - 1u));
          if (q > 2u) z2 = re_draw(prm.lmg_seq[2u * t + 1u] + c * (q - 1u) + 1u);
          // u_k = sum_{j <= k} l[k][j] z_j, j ascending
          ux[(size_t)c * q + 1u] = (float)(0.0 + lc[q] * z0 + lc[q + 1u] * z1);
          if (q > 2u) ux[(size_t)c * q + 2u] = (float)(0.0 + lc[2u * q] * z0 + lc[2u * q + 1u] * z1 + lc[2u * q + 2u] * z2);
        }
      }
      ux += (size_t)nl * q;
    }
  }
  lmg_sync();

  // ---- 2. the replicate's rows (lane = row) ----
  RowCtx<P> cx;
  row_ctx_init<P, true>(prm, cx);  // factor columns cost nothing here: generation is a sliver of this kernel
  ScenLane<P> sl;
  if (SCEN) scen_lane_load<P>(prm, lsim, sl);
  // latent-scale random effect of a row in the reference's order: primary intercept, extras (intercept, slopes),
  // primary slopes (data_gen.rs:812-842)
  auto row_effect = [&](uint32_t rr, const float (&x)[P]) -> double {
    const uint32_t cid = lmg_level_of_row(prm, 0u, rr);
    double u_re = (double)m.u0[cid];
    const ldf* ux = m.ux;
    for (uint32_t t = 1; t < nterms; ++t) {
      const uint32_t q = prm.lmg_q[t], lvl = lmg_level_of_row(prm, t, rr);
      u_re += (double)ux[(size_t)lvl * q];
      for (uint32_t s = 1; s < q; ++s) {
        float xv = 0.0f;
#pragma unroll
        for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[t][s - 1u] ? x[c] : xv;
        u_re += (double)ux[(size_t)lvl * q + s] * (double)xv;
      }
      ux += (size_t)prm.lmg_levels[t] * q;
    }
#pragma unroll
    for (int s = 1; s < Q0; ++s) {
      float xv = 0.0f;
#pragma unroll
      for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[0][s - 1] ? x[c] : xv;
      u_re += (double)m.us[(size_t)cid * (Q0 - 1) + (s - 1)] * (double)xv;
    }
    return u_re;
  };
  const uint32_t bmask = prm.between_mask;
  // Streaming (prm.lmg_stream: one sample size, no cluster-level columns, no factors): the rows exist only to be folded
  // into the sufficient statistics, in row order -- generate 256 of them (one Philox quad per lane), fold, reuse the
  // window.  dstride / dbase: row stride of a design column in LDS and the first row the window holds.
  const bool stream = prm.lmg_stream != 0u;
  const uint32_t dstride = stream ? 256u : n_pad;
  auto gen_quad = [&](uint32_t ctr, uint32_t dbase) {
    stage_group<P, GW, 0, SCEN>(prm, k0, k1, ctr, m.st, SCEN ? sl.meta : 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rr = 4u * ctr + r;
      if (rr < max_n) {
        float x[P], y;
        const float e = build_x<P, true, GW, SCEN>(prm, cx, k0, k1, m.st, r, rr, x, &sl);
        if (bmask == 0u) finish_row<P, 0, SCEN>(prm, cx, x, e, y, row_effect(rr, x), &sl);
        else y = e;  // the residual waits in the outcome slot until the cluster-level columns are broadcast
#pragma unroll
        for (int c = 1; c < P; ++c) m.design[(size_t)(c - 1) * dstride + (rr - dbase)] = x[c];
        m.design[(size_t)(P - 1) * dstride + (rr - dbase)] = y;
      }
    }
  };
  if (!stream)
    for (uint32_t ctr = lane; 4u * ctr < max_n; ctr += GW) gen_quad(ctr, 0u);
  if (bmask != 0u) {
    // cluster-level predictors (data_gen.rs:606-636): every row takes the marked columns from its primary cluster's
    // representative row, then interactions, the random-effect contribution and the outcome follow from the
    // broadcast values; the marked columns themselves are rewritten last (see lmm_kernels.hip)
    lmg_sync();
    const uint32_t cvb = prm.cluster_v;
    for (uint32_t rr = lane; rr < max_n; rr += GW) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cvb : rr - (rr % cvb);
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
      finish_row<P, 0, SCEN>(prm, cx, x, e, y, row_effect(rr, x), &sl);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 1; c < P; ++c)
        if (((bmask >> c) & 1u) == 0u && ((cx.inter_any >> c) & 1u)) m.design[(size_t)(c - 1) * n_pad + rr] = x[c];
      m.design[(size_t)(P - 1) * n_pad + rr] = y;
    }
    lmg_sync();
    for (uint32_t rr = lane; rr < max_n; rr += GW) {
      const uint32_t rep = prm.cluster_fixed_clusters ? rr % cvb : rr - (rr % cvb);
#pragma unroll
      for (int c = 1; c < P; ++c)
        if ((bmask >> c) & 1u) {
          const float v = m.design[(size_t)(c - 1) * n_pad + rep];
          __builtin_amdgcn_wave_barrier();
          m.design[(size_t)(c - 1) * n_pad + rr] = v;
        }
    }
  }
  // ---- 3. sufficient statistics, zeroed once and grown over the grid ----
  for (size_t i = lane; i < (size_t)LE * Q0 * Q0 + (size_t)LE * Q0 * Wc + lmg_tri(W, 0u); i += GW) m.gzz[i] = 0.0;  // gzz | gzw | gww are contiguous
  lmg_sync();

  uint32_t nzr = 0;  // non-zeros of the rest part of a row
  for (uint32_t r = 0; r < prm.lmg_n_rest; ++r) nzr += prm.lmg_q[prm.lmg_frest[r]];
  const uint32_t nz = nzr + (uint32_t)P + 1u;
  bool has_prev = false;
  uint32_t prev_n = 0;
  const uint32_t nt = prm.n_tests;
  // fold rows [lo, hi) into the statistics, in row order; the window holds them from row dbase on
  auto fold_rows = [&](uint32_t lo, uint32_t hi, uint32_t dbase) {
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t di = i - dbase;
      if (lane < nz) {
        uint32_t idx;
        double val;
        if (lane < nzr) {
          uint32_t r = 0, e = lane;
          while (e >= prm.lmg_q[prm.lmg_frest[r]]) { e -= prm.lmg_q[prm.lmg_frest[r]]; ++r; }
          const uint32_t t = prm.lmg_frest[r];
          idx = prm.lmg_froff[t] + lmg_level_of_row(prm, t, i) * prm.lmg_q[t] + e;
          val = e == 0u ? 1.0 : (double)m.design[(size_t)(prm.lmg_cols[t][e - 1u] - 1u) * dstride + di];
        } else {
          const uint32_t a = lane - nzr;  // 0..P-1 design column, P = outcome
          idx = R + a;
          val = a == 0u ? 1.0 : (double)m.design[(size_t)(a - 1u) * dstride + di];
        }
        m.ridx[lane] = idx;
        m.rval[lane] = val;
      }
      lmg_sync();
      const uint32_t c = lmg_level_of_row(prm, E, i);
      for (uint32_t pr = lane; pr < nz * nz; pr += GW) {
        const uint32_t a = pr / nz, b = pr % nz;
        const uint32_t ia = m.ridx[a], ib = m.ridx[b];
        if (ib <= ia) m.gww[lmg_tri(ia, ib)] = __builtin_fma(m.rval[a], m.rval[b], m.gww[lmg_tri(ia, ib)]);
      }
      for (uint32_t pr = lane; pr < (uint32_t)Q0 * nz; pr += GW) {
        const uint32_t a = pr / nz, b = pr % nz;
        const double za = a == 0u ? 1.0 : (double)m.design[(size_t)(prm.lmg_cols[E][a - 1u] - 1u) * dstride + di];
        ldd* cell = m.gzw + ((size_t)c * Q0 + a) * Wc + lmg_cpos(prm, compact, m.ridx[b]);
        *cell = __builtin_fma(za, m.rval[b], *cell);
      }
      if (lane < (uint32_t)(Q0 * Q0)) {
        const uint32_t a = lane / Q0, b = lane % Q0;
        const double za = a == 0u ? 1.0 : (double)m.design[(size_t)(prm.lmg_cols[E][a - 1u] - 1u) * dstride + di];
        const double zb = b == 0u ? 1.0 : (double)m.design[(size_t)(prm.lmg_cols[E][b - 1u] - 1u) * dstride + di];
        ldd* cell = m.gzz + (size_t)c * Q0 * Q0 + lane;
        *cell = __builtin_fma(za, zb, *cell);
      }
      lmg_sync();
    }
  };
  if (stream) {  // generate and fold window by window (n_grid = 1: every row belongs to the one fit)
    const uint32_t n_only = prm.ns[0];
    for (uint32_t base = 0; base < max_n; base += 256u) {
      if (4u * (base / 4u + lane) < max_n) gen_quad(base / 4u + lane, base);
      lmg_sync();
      const uint32_t hi = base + 256u < n_only ? base + 256u : n_only;
      if (base < hi) fold_rows(base, hi, base);
      lmg_sync();
    }
    prev_n = n_only;
  }
  for (uint32_t g = 0; g < prm.n_grid; ++g) {
    const uint32_t n = prm.ns[g];
    if (prev_n < n) fold_rows(prev_n, n, 0u);
    prev_n = n;
    // ---- which model this round fits: sparse-factor flags recounted from rows [0, n) (batch.rs:1236-1250), then the
    //      kept columns of that flag combination (test_formula subsets included) ----
    uint32_t flag_bits = 0;
    const uint32_t full_mask = (1u << P) - 1u;
    if (prm.red_active && prm.min_level_count) {
#pragma unroll
      for (int f = 0; f < MAX_FACTORS; ++f) {
        if ((uint32_t)f < prm.n_factors) {
          uint32_t mn = 0xffffffffu;
          for (uint32_t l = 0; l < prm.f_levels[f]; ++l) {
            uint32_t cnt = 0;
            for (uint32_t i = lane; i < n; i += GW) {
              uint32_t lv = 0;
#pragma unroll
              for (int c = 1; c < P; ++c)
                if (prm.col_factor[c] == (uint32_t)f && m.design[(size_t)(c - 1) * n_pad + i] == 1.0f) lv = prm.col_level[c];
              cnt += lv == l ? 1u : 0u;
            }
#pragma unroll
            for (int mm = 1; mm < GW; mm <<= 1) cnt += __shfl_xor(cnt, mm, GW);
            mn = cnt < mn ? cnt : mn;
          }
          if (mn < prm.min_level_count) flag_bits |= 1u << f;
        }
      }
    }
    const uint32_t keep = prm.red_active ? prm.keep_mask[flag_bits] : full_mask;
    const bool reduced_round = keep != full_mask;
    LmgRed red;
    red.keep = keep;
    red.p_fit = (uint32_t)__popc(keep);
    const double n_rows = (double)n;
    ldd* x0 = m.simplex;
    uint32_t n_eval = 0, pinned = 0;
    bool conv = false, at_upper = false;
    if (!reduced_round) {
      // ---- minimise: truth start / carried theta-hat / blind start (batch.rs:1316-1332) ----
      if (lane < d) {
        double v;
        if (has_prev) v = m.carry[lane];  // theta-hat of the last converged grid point
        else if (prm.truth_start) {
          uint32_t li = lane;
          asm volatile("" : "+v"(li));  // opaque per grid point: hoisted out of the grid loop, the table address sat in two VGPRs for the whole kernel (12 B of scratch at the 168-register cap)
          v = prm.lmg_theta_truth[li];
        }
        else v = lmg_is_diag(prm, lane) ? 1.0 : 0.0;
        if ((has_prev || prm.truth_start) && v < 0.01) v = 0.01;  // THETA_TRUTH_FLOOR on every component of a given start
        x0[lane] = v;
      }
      lmg_sync();
      // the fit: one wave-uniform state machine around the single inlined evaluation (mcp_fit_fsm.hpp)
      uint32_t lower0 = 0u;
      for (uint32_t k = 0; k < d; ++k) lower0 |= lmg_is_diag(prm, k) ? (1u << k) : 0u;
      const FitFsmOut fit = simplex_fit_fsm(
          m.simplex, m.fv, d, (uint32_t)LMG_MAX_THETA, lower0, d, LMG_MAX_EVAL, lane,
          [&](ldd* x) -> double {
            lmg_clamp<false>(prm, x, lane);
            return lmg_eval<Q0>(prm, m, x, lane, n_rows);
          },
          [&](double, uint32_t) -> uint32_t { return 0u; }, true);
      const double fin = fit.cur;
      n_eval = fit.n_eval;
      conv = fin < __builtin_inf() && n_eval < LMG_MAX_EVAL;
      uint32_t comp = 0;
      for (uint32_t k = 0; k < d; ++k) {
        if (!lmg_is_diag(prm, k)) continue;
        if (x0[k] < LMG_THETA_ZERO) pinned |= 1u << comp;
        if (x0[k] > LMG_THETA_MAX - 1e-3) at_upper = true;
        ++comp;
      }
    } else {
      // ---- reduced round (batch.rs:1343-1400): intercept-only groupings on the kept columns, started at the truth
      //      taus (or blind); it neither reads nor feeds the theta carry (:1414-1428) ----
      if (lane < nterms) {
        double v = prm.truth_start ? prm.lmg_tau[lane] : 1.0;
        if (prm.truth_start && v < 0.01) v = 0.01;
        x0[lane] = v;
      }
      lmg_sync();
      const FitFsmOut fit = simplex_fit_fsm(
          m.simplex, m.fv, nterms, (uint32_t)LMG_MAX_THETA, (1u << nterms) - 1u, nterms, LMG_MAX_EVAL, lane,
          [&](ldd* x) -> double {
            lmg_clamp<true>(prm, x, lane);
            return lmg_eval_red<Q0>(prm, m, x, lane, n_rows, red);
          },
          [&](double, uint32_t) -> uint32_t { return 0u; }, true);
      const double fin = fit.cur;
      n_eval = fit.n_eval;
      conv = fin < __builtin_inf() && n_eval < LMG_MAX_EVAL;
      for (uint32_t k = 0; k < nterms; ++k) {
        if (x0[k] < LMG_THETA_ZERO) pinned |= 1u << k;
        if (x0[k] > LMG_THETA_MAX - 1e-3) at_upper = true;
      }
    }
    const uint32_t bh = !conv ? 0u : (at_upper ? 2u : (pinned ? 1u : 0u));
    if (at_upper) conv = false;
    if (conv && !reduced_round) {  // grid-sequential warm start (batch.rs:1414-1428): a diverged theta-hat never replaces the carry
      if (lane < d) m.carry[lane] = x0[lane];
      has_prev = true;
    }
    // ---- Wald z^2, joint Wald chi2, decisions: scalar work of one lane ----
    if (lane == 0u) {
      const uint32_t p = (uint32_t)P;
      constexpr uint32_t TS = P > 8 ? 16u : 8u;
      const ldd* beta = m.tail;
      const ldd* Lx = m.tail + TS;
      ldd* U = m.tail + TS + TS * TS;
      constexpr int MT = P > 8 ? 12 : MAX_T_WAVE;      // tests per fit: the wide designs test up to 12
      constexpr uint32_t TC = P > 8 ? 16u : 8u;        // row stride of the joint-test factor
      ldd* Lc = U + (uint32_t)MT * TS;
      ldd* cj = Lc + TC * TC;
      const double sigma_sq = m.misc[0];
      const double* crow = prm.crit + (size_t)g * prm.crit_stride;
      double tsq[MT];
      uint32_t tkeep = 0;  // targets whose column is in the fitted model
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt && ((keep >> prm.test_pos[t]) & 1u)) tkeep |= 1u << t;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        tsq[t] = __builtin_nan("");
        for (uint32_t i = 0; i < p; ++i) U[t * TS + i] = 0.0;
        if (((tkeep >> t) & 1u) && conv) {
          const double* c = prm.contrast + t * MAX_P;
          double est = 0.0, nq = 0.0;
          for (uint32_t i = 0; i < p; ++i) {
            est = __builtin_fma(c[i], beta[i], est);
            double acc = c[i];
            for (uint32_t q = 0; q < i; ++q) acc = __builtin_fma(-Lx[i * TS + q], U[t * TS + q], acc);
            U[t * TS + i] = acc / Lx[i * TS + i];
            nq = __builtin_fma(U[t * TS + i], U[t * TS + i], nq);
          }
          const double var = sigma_sq * nq;
          if (var > LMG_NEAR_ZERO && var < __builtin_inf()) tsq[t] = est * est / var;
        }
      }
      bool joint_sig = false;
      if (conv && tkeep != 0u) {
        bool jok = true;
        double qsum = 0.0;
        for (uint32_t j = 0; j < nt; ++j) {
          if (!((tkeep >> j) & 1u)) {  // untestable target: identity row, no contribution
            for (uint32_t i2 = j; i2 < nt; ++i2) Lc[i2 * TC + j] = i2 == j ? 1.0 : 0.0;
            continue;
          }
          double dd = 0.0;
          for (uint32_t i = 0; i < p; ++i) dd = __builtin_fma(U[j * TS + i], U[j * TS + i], dd);
          dd *= sigma_sq;
          for (uint32_t t = 0; t < j; ++t) dd = __builtin_fma(-Lc[j * TC + t], Lc[j * TC + t], dd);
          jok = jok && (dd > 0.0);
          Lc[j * TC + j] = __builtin_sqrt(dd);
          for (uint32_t i2 = j + 1u; i2 < nt; ++i2) {
            double vv = 0.0;
            for (uint32_t i = 0; i < p; ++i) vv = __builtin_fma(U[i2 * TS + i], U[j * TS + i], vv);
            vv *= sigma_sq;
            for (uint32_t t = 0; t < j; ++t) vv = __builtin_fma(-Lc[i2 * TC + t], Lc[j * TC + t], vv);
            Lc[i2 * TC + j] = vv / Lc[j * TC + j];
          }
        }
        for (uint32_t i = 0; i < nt; ++i) {
          const double* c = prm.contrast + i * MAX_P;
          double vv = 0.0;
          for (uint32_t q = 0; q < p; ++q) vv = __builtin_fma(((tkeep >> i) & 1u) ? c[q] : 0.0, beta[q], vv);
          for (uint32_t t = 0; t < i; ++t) vv = __builtin_fma(-Lc[i * TC + t], cj[t], vv);
          cj[i] = vv / Lc[i * TC + i];
          qsum = __builtin_fma(cj[i], cj[i], qsum);
        }
        // the table row holds chi2(n_targets); a reduced round tests its kept targets (batch.rs:1456-1468)
        const double joint_crit = reduced_round ? prm.joint_crit_k[__popc(tkeep)] : crow[2 + nt];
        joint_sig = jok && (qsum < __builtin_inf()) && (qsum > joint_crit);
      }
      uint32_t unc_bits = 0, cor_bits = 0;
      decide_tests<MT>(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
      if (!conv) {
        unc_bits = 0;
        cor_bits = 0;
      }
      const size_t cell = (size_t)(prm.out_base + lsim) * prm.n_grid + g;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
      prm.conv[cell] = conv ? 1 : 0;
      prm.boundary[cell] = (uint8_t)bh;
      prm.joint[cell] = joint_sig ? 1 : 0;
      if (prm.pinned) prm.pinned[cell] = (uint16_t)(conv ? pinned : 0u);
      if (prm.factor_excluded)
        for (uint32_t f = 0; f < prm.n_factors; ++f) prm.factor_excluded[cell * prm.n_factors + f] = (uint8_t)((flag_bits >> f) & 1u);
      if (prm.overall) prm.overall[cell] = 0;  // batch.rs:1449-1451
      if (prm.stat && g == 0u) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
          if ((uint32_t)t < nt) prm.stat[(size_t)(prm.out_base + lsim) * nt + t] = tsq[t];
      }
    }
    lmg_sync();
  }
}

}  // namespace
}  // namespace mcp
