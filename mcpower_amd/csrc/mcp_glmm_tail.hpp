// mcp_glmm_tail.hpp -- the once-per-fit tail of the clustered-GLM kernels (glmm_kernels.hip, glmm_slopes_kernels.hip) in
// LDS: Cholesky factor of the Wald precision over the fitted columns, Wald z^2 per test (dispersion 1), joint Wald
// chi2 over the targets, decisions and the flag stores (batch.rs:1723-1830) -- the scalar work of ONE lane over
// run-time loops, the way lmm_general_impl.hpp does it.  The register-resident form (Lx[P][P], U[8][P], Lc[8][8]
// unrolled) needed > 256 VGPRs at P >= 6 and spilled 150-1000 B per lane once per fit; here nothing of size P^2 lives
// in registers.  Same operations in the same order as the register form (and as oracle/fit_glmm.c).
// Included inside namespace mcp { namespace { after `gdd` and GLMM_NEAR_ZERO.
#pragma once

// tail block (doubles): beta[8] | Lx[64] (row stride 8) | U[64] | Lc[64] | cj[8] | spare[8]
constexpr uint32_t GT_BETA = 0, GT_LX = 8, GT_U = 72, GT_LC = 136, GT_CJ = 200, GT_SPARE = 208;

// ONE lane: lower Cholesky factor of the symmetric matrix whose lower triangle sits at A[r * 8 + s] into L[r * 8 + s]
// (A == L allowed) over the kept columns; a dropped column is an identity row / column, so the kept block factors as the
// gathered system (batch.rs:1588-1640)
__device__ __forceinline__ bool glmm_tail_factor(const gdd* A, gdd* L, uint32_t p, uint32_t keep) {
  bool fok = true;
  for (uint32_t j = 0; j < p; ++j) {
    const bool kj = (keep >> j) & 1u;
    double dd = kj ? A[j * 8u + j] : 1.0;
    for (uint32_t t = 0; t < j; ++t) dd -= L[j * 8u + t] * L[j * 8u + t];
    fok = fok && (dd > 0.0) && (dd < __builtin_inf());
    const double ljj = __builtin_sqrt(dd);
    for (uint32_t i = j + 1u; i < p; ++i) {
      double v = (kj && ((keep >> i) & 1u)) ? A[i * 8u + j] : 0.0;
      for (uint32_t t = 0; t < j; ++t) v -= L[i * 8u + t] * L[j * 8u + t];
      L[i * 8u + j] = v / ljj;
    }
    L[j * 8u + j] = ljj;
  }
  return fok;
}

// ONE lane: Wald statistics from beta = tail[GT_BETA..] and the factor at tail[GT_LX..], decisions, flag stores
__device__ __forceinline__ void glmm_tail_decide(const SimParams& prm, gdd* tail, uint32_t p, uint32_t nt, bool conv, uint32_t bh,
                                                 uint32_t pinned, double tau_sq, const double* crow, size_t cell, size_t stat_row,
                                                 bool first_grid) {
  const gdd* beta = tail + GT_BETA;
  const gdd* Lx = tail + GT_LX;
  gdd* U = tail + GT_U;
  gdd* Lc = tail + GT_LC;
  gdd* cj = tail + GT_CJ;
  double tsq[MAX_T_WAVE];
#pragma unroll
  for (int t = 0; t < MAX_T_WAVE; ++t) {
    tsq[t] = __builtin_nan("");
    for (uint32_t i = 0; i < p; ++i) U[t * 8 + i] = 0.0;
    if ((uint32_t)t < nt && conv) {
      const double* c = prm.contrast + t * MAX_P;
      double est = 0.0, nq = 0.0;
      for (uint32_t i = 0; i < p; ++i) {
        est = __builtin_fma(c[i], beta[i], est);
        double acc = c[i];
        for (uint32_t q = 0; q < i; ++q) acc = __builtin_fma(-Lx[i * 8u + q], U[t * 8 + q], acc);
        const double ui = acc / Lx[i * 8u + i];
        U[t * 8 + i] = ui;
        nq = __builtin_fma(ui, ui, nq);
      }
      if (nq > GLMM_NEAR_ZERO && nq < __builtin_inf()) tsq[t] = est * est / nq;
    }
  }
  bool joint_sig = false;
  if (conv && nt != 0u) {
    bool jok = true;
    double qsum = 0.0;
    for (uint32_t j = 0; j < nt; ++j) {
      double dd = 0.0;
      for (uint32_t i = 0; i < p; ++i) dd = __builtin_fma(U[j * 8u + i], U[j * 8u + i], dd);
      for (uint32_t t = 0; t < j; ++t) dd = __builtin_fma(-Lc[j * 8u + t], Lc[j * 8u + t], dd);
      jok = jok && (dd > 0.0);
      const double ljj = __builtin_sqrt(dd);
      Lc[j * 8u + j] = ljj;
      for (uint32_t i2 = j + 1u; i2 < nt; ++i2) {
        double vv = 0.0;
        for (uint32_t i = 0; i < p; ++i) vv = __builtin_fma(U[i2 * 8u + i], U[j * 8u + i], vv);
        for (uint32_t t = 0; t < j; ++t) vv = __builtin_fma(-Lc[i2 * 8u + t], Lc[j * 8u + t], vv);
        Lc[i2 * 8u + j] = vv / ljj;
      }
    }
    for (uint32_t i = 0; i < nt; ++i) {
      const double* c = prm.contrast + i * MAX_P;
      double vv = 0.0;
      for (uint32_t q = 0; q < p; ++q) vv = __builtin_fma(c[q], beta[q], vv);
      for (uint32_t t = 0; t < i; ++t) vv = __builtin_fma(-Lc[i * 8u + t], cj[t], vv);
      const double ci = vv / Lc[i * 8u + i];
      cj[i] = ci;
      qsum = __builtin_fma(ci, ci, qsum);
    }
    joint_sig = jok && (qsum < __builtin_inf()) && (qsum > prm.joint_crit_k[nt]);
  }
  uint32_t unc_bits = 0, cor_bits = 0;
  decide_tests<MAX_T_WAVE>(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
  if (!conv) {
    unc_bits = 0;
    cor_bits = 0;
  }
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
  if (prm.tau_hat) prm.tau_hat[cell] = conv ? tau_sq : __builtin_nan("");
  if (prm.overall) prm.overall[cell] = 0;  // batch.rs:1750-1753
  if (prm.stat && first_grid) {
#pragma unroll
    for (int t = 0; t < MAX_T_WAVE; ++t)
      if ((uint32_t)t < nt) prm.stat[stat_row * nt + t] = tsq[t];
  }
}

// ONE lane, Wald SE mode "hessian": gather S / 2 from the stencil block (rows 1 .. d of the simplex, mcp_fit_fsm.hpp) over
// the kept columns into the U area, factor it there, and replace the "rx" factor on success (a block that does not
// factor keeps "rx", as vcov.merMod falls back)
__device__ __forceinline__ void glmm_tail_take_hessian(gdd* tail, const gdd* Hm, uint32_t hrow, uint32_t nth, uint32_t p, uint32_t keep) {
  gdd* T = tail + GT_U;
  uint32_t kr = 0;
  for (uint32_t r = 0; r < p; ++r) {
    uint32_t kc = 0;
    for (uint32_t s = 0; s <= r; ++s) {
      const bool both = ((keep >> r) & 1u) && ((keep >> s) & 1u);
      T[r * 8u + s] = both ? Hm[(size_t)(nth + kr) * hrow + nth + kc] : 0.0;
      kc += ((keep >> s) & 1u) ? 1u : 0u;
    }
    kr += ((keep >> r) & 1u) ? 1u : 0u;
  }
  if (glmm_tail_factor(T, T, p, keep)) {
    gdd* Lx = tail + GT_LX;
    for (uint32_t r = 0; r < p; ++r)
      for (uint32_t s = 0; s <= r; ++s) Lx[r * 8u + s] = T[r * 8u + s];
  }
}
