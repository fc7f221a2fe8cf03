// mcp_rows.hpp -- per-replicate data generation shared by the OLS (lane-per-replicate) and GLM / LMM
// (wave-per-replicate) kernels: 4-row Philox blocks staged in LDS with the deferred inverse-CDF tail
// fix-up, and the row builder (L.z mix, factor dummies, interactions, outcome).
// Reference: crates/engine-core/src/data_gen.rs:87-925 (line ranges at each function).
#pragma once
#include "mcp_device.hpp"
#include "mcp_params.hpp"

#pragma clang fp contract(off)

namespace mcp {

// categorical draw on the sequential stream (rng.rs:112-127)
__device__ __forceinline__ uint32_t categorical(const SimParams& prm, uint32_t f, uint32_t k0, uint32_t k1, uint64_t q) {
  const double u = (double)u32_to_unit_f32(sequential_word(k0, k1, q));
  const uint32_t nl = prm.f_levels[f];
  const uint32_t off = prm.f_prop_off[f];
  double acc = 0.0;
  uint32_t lvl = nl - 1;
  bool found = false;
  for (uint32_t i = 0; i < nl; ++i) {
    acc += prm.f_props[off + i];
    if (!found && u < acc) {
      lvl = i;
      found = true;
    }
  }
  return lvl;
}

// ---- 4-row block staging ------------------------------------------------------------------
// LDS layout: stage[slot * 256 + tid], slot = 4 * column + r, column j < K = continuous predictor j,
// column K = the residual.  One Philox call yields rows 4*ctr..4*ctr+3 of one column (rng.rs:149-178).

template <int STRIDE>
__device__ __forceinline__ void stage_column(float* st, uint32_t col_slot, const u32x4 w, uint32_t& pend) {
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = u32_to_unit_f32(ws[r]) - 0.5f;
    const bool tail = __builtin_fabsf(v) > NORM_INV_BP;
    const float z = icdf_central(v);           // branch-free central polynomial for every lane
    st[(col_slot * 4u + r) * STRIDE] = tail ? v : z;  // tail lanes park v for the fix-up loop
    pend |= tail ? (1u << (col_slot * 4u + r)) : 0u;
  }
}

// rows 4*ctr .. 4*ctr+3 of one replicate: standard normals of every column into LDS (data_gen.rs:316-421)
// OUTCOME: 0 = continuous (normal residual), 1 = binary logit (the residual slot holds the Uniform(0,1) draw,
// data_gen.rs:390-398)
template <int P, int STRIDE, int OUTCOME>
__device__ __forceinline__ void stage_group(const SimParams& prm, uint32_t k0, uint32_t k1, uint32_t ctr, float* st) {
  const uint32_t K = prm.K;
  uint32_t pend = 0;
#pragma unroll
  for (int j = 0; j < P - 1; ++j)
    if ((uint32_t)j < K) stage_column<STRIDE>(st, (uint32_t)j, planar_words(k0, k1, CLASS_XNORM, (uint32_t)j, ctr), pend);
  if (OUTCOME == 0) {
    stage_column<STRIDE>(st, K, planar_words(k0, k1, CLASS_RESID, 0u, ctr), pend);
  } else {
    const u32x4 w = planar_words(k0, k1, CLASS_RESID, 0u, ctr);
    st[(K * 4u + 0u) * STRIDE] = u32_to_unit_f32(w.x);
    st[(K * 4u + 1u) * STRIDE] = u32_to_unit_f32(w.y);
    st[(K * 4u + 2u) * STRIDE] = u32_to_unit_f32(w.z);
    st[(K * 4u + 3u) * STRIDE] = u32_to_unit_f32(w.w);
  }
  // tail fix-up: each pass retires one parked draw per lane; ~3-4 passes per block instead of 4*(K+1)
  while (__ballot(pend != 0u) != 0ull) {
    if (pend != 0u) {
      const uint32_t slot = (uint32_t)__builtin_ctz(pend);
      pend &= pend - 1u;
      st[slot * STRIDE] = icdf_tail(st[slot * STRIDE]);
    }
  }
}

// launch-constant values of the row loop, hoisted into scalar registers once per wave
template <int P>
struct RowCtx {
  uint32_t K;
  uint32_t ident_L;
  uint32_t n_factors;        // 0 on the continuous-only fast path (FACTORS == false)
  uint32_t beta_nz;          // bit c set when beta[c] != 0
  uint32_t inter_any;        // bit c set when column c is an interaction
  uint32_t imask[P];
  double beta[P];
};

template <int P, bool FACTORS>
__device__ __forceinline__ void row_ctx_init(const SimParams& prm, RowCtx<P>& cx) {
  cx.K = prm.K;
  cx.ident_L = prm.ident_L;
  cx.n_factors = FACTORS ? prm.n_factors : 0u;
  cx.beta_nz = 0u;
  cx.inter_any = 0u;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    cx.beta[c] = prm.beta[c];
    cx.imask[c] = prm.inter_mask[c];
    cx.beta_nz |= (cx.beta[c] != 0.0) ? (1u << c) : 0u;
    cx.inter_any |= (cx.imask[c] != 0u) ? (1u << c) : 0u;
  }
}

// one design row + outcome from the staged normals (data_gen.rs:334-363, 509-551, 643-649, 781-857)
template <int P, bool FACTORS, int STRIDE, int OUTCOME>
__device__ __forceinline__ void build_row(const SimParams& prm, const RowCtx<P>& cx, uint32_t k0, uint32_t k1,
                                          const float* st, int r, uint32_t row, float (&x)[P], float& y,
                                          double u_re = 0.0) {
  const uint32_t K = cx.K;
  float z[P - 1];
#pragma unroll
  for (int j = 0; j < P - 1; ++j) z[j] = ((uint32_t)j < K) ? st[(j * 4 + r) * STRIDE] : 0.0f;
  const float e = st[(K * 4u + r) * STRIDE];
  x[0] = 1.0f;
  // factor levels of this row (data_gen.rs:513-551)
  uint32_t lvl[MAX_FACTORS];
  if (FACTORS) {
#pragma unroll
    for (int f = 0; f < MAX_FACTORS; ++f) {
      lvl[f] = 0u;
      if ((uint32_t)f < cx.n_factors) {
        if (prm.f_sampled[f]) {
          const uint64_t q = (uint64_t)prm.n_seq_pre + (uint64_t)row * prm.n_seq_per_row + prm.f_seq_slot[f];
          lvl[f] = categorical(prm, (uint32_t)f, k0, k1, q);
        } else {
          lvl[f] = row < prm.max_n ? (uint32_t)prm.fixed_levels[(size_t)row * prm.n_factors + f] : 0u;
        }
      }
    }
  }
#pragma unroll
  for (int c = 1; c < P; ++c) {
    float v = 0.0f;
    if ((uint32_t)(c - 1) < K) {
      if (cx.ident_L) {
        v = z[c - 1];
      } else {  // x_j = f32( sum_{t<=j} L[j][t] * f64(z_t) ), t ascending, separately rounded (data_gen.rs:334-351)
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < c; ++t) {
          const double l = prm.L[(c - 1) * MAX_K + t];
          if (l != 0.0) acc = acc + l * (double)z[t];
        }
        v = (float)acc;
      }
    } else if ((cx.inter_any >> c) & 1u) {  // raw product in f64, components ascending (data_gen.rs:643-649)
      double prod = 1.0;                    // non-members multiply by exactly 1.0: no branch, same bits
#pragma unroll
      for (int cc = 1; cc < c; ++cc) prod = prod * (((cx.imask[c] >> cc) & 1u) ? (double)x[cc] : 1.0);
      v = (float)prod;
    } else if (FACTORS && prm.col_factor[c] != 0xffffffffu) {
      const uint32_t f = prm.col_factor[c];
      const uint32_t lv = f == 0 ? lvl[0] : (f == 1 ? lvl[1] : (f == 2 ? lvl[2] : lvl[3]));
      v = (lv == prm.col_level[c]) ? 1.0f : 0.0f;
    }
    x[c] = v;
  }
  // y = f32( sum_j f64(x_ij) beta_j + f64(eps_i) + u_re )  (data_gen.rs:785-857); zero betas add exact zeros
  double lp = 0.0;
#pragma unroll
  for (int c = 0; c < P; ++c)
    if ((cx.beta_nz >> c) & 1u) lp = lp + (double)x[c] * cx.beta[c];
  if (OUTCOME == 0) {
    y = (float)((lp + (double)e) + u_re);  // u_re = 0.0 without a cluster (data_gen.rs:814-857)
  } else {  // Bernoulli(sigmoid(eta)): 1{u < sigma(eta)} with the pre-drawn uniform (data_gen.rs:890-911)
    const double eta = lp + u_re;
    const double pr = eta >= 0.0 ? 1.0 / (1.0 + exp(-eta)) : exp(eta) / (1.0 + exp(eta));
    y = ((double)e < pr) ? 1.0f : 0.0f;
  }
}

}  // namespace mcp
