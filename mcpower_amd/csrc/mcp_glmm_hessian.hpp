// mcp_glmm_hessian.hpp -- Wald SE mode "hessian" of the clustered-GLM kernels (engine-contract estimator.rs:39-56,
// lme4 `use.hessian = TRUE`): the beta block of 2 H^-1, H the central finite-difference Hessian of the Laplace
// deviance over (theta, beta) at the optimum with lme4's `deriv12` stencil (delta = 1e-4; the objective clamps a
// step past a bound onto it).  Same stencil, same operation order as the checker restates in oracle/fit_glmm.c.
// Included by glmm_kernels.hip / glmm_slopes_kernels.hip after their `gdd` / `gq_sync` definitions.
#pragma once

// f(point) -> deviance, wave-uniform, may clamp the point in place.  sx: the simplex block (row 0 = the optimum, d
// entries: nth theta then the fitted coefficients); rows 1 .. d hold H, row d + 3 the stencil point.  On success the
// lower triangle of S / 2, S = H_bb - H_bt H_tt^-1 H_tb, sits at H[nth + r][nth + c] and the function returns true;
// false when a block does not factor (the caller keeps the "rx" covariance, as vcov.merMod falls back).
// theta_free: bit k set = covariance parameter k takes part.  A diagonal pinned at 0 is left out: it sits at an even
// point of the deviance (flipping the sign of a column of Lambda only relabels u), so its cross-derivatives vanish and
// the beta block is the same without it -- its row / column become identity (exact zeros in every later sum: the
// free parameters go through the operation sequence of the compacted system the oracle factors).
template <class F>
__device__ __noinline__ bool glmm_fd_hessian_schur(F& f, gdd* sx, uint32_t row, uint32_t nth, uint32_t d, double fx, uint32_t lane,
                                                   uint32_t theta_free) {
  const double dl = 1e-4;
  gdd* pt = sx + (size_t)row * (d + 3u);
  gdd* H = sx + row;  // H[a][b] at H[a * row + b]
  auto is_free = [&](uint32_t k) -> bool { return k >= nth || ((theta_free >> k) & 1u); };
  for (uint32_t a = 0; a < d; ++a) {
    if (!is_free(a)) {
      gq_sync();
      if (lane < d) {
        H[(size_t)a * row + lane] = lane == a ? 1.0 : 0.0;
        H[(size_t)lane * row + a] = lane == a ? 1.0 : 0.0;
      }
      continue;
    }
    double fadd = 0.0, fsub = 0.0;
    for (int q = 0; q < 2; ++q) {
      gq_sync();
      if (lane < d) pt[lane] = lane == a ? sx[lane] + (q ? -dl : dl) : sx[lane];
      gq_sync();
      const double v = f(pt);
      if (q == 0) fadd = v;
      else fsub = v;
    }
    if (lane == 0u) H[(size_t)a * row + a] = fadd / (dl * dl) - 2.0 * fx / (dl * dl) + fsub / (dl * dl);
    for (uint32_t b = 0; b < a; ++b) {
      if (!is_free(b)) continue;  // that row / column is already identity
      double acc = 0.0;  // f++ - f+- - f-+ + f--, summed in that order
      for (int q = 0; q < 4; ++q) {
        gq_sync();
        if (lane < d) {
          double v = sx[lane];
          if (lane == b) v = v + ((q & 2) ? -dl : dl);
          if (lane == a) v = v + ((q & 1) ? -dl : dl);
          pt[lane] = v;
        }
        gq_sync();
        const double v = f(pt);
        acc = q == 0 ? v : ((q == 3) ? acc + v : acc - v);
      }
      if (lane == 0u) {
        const double h = acc / (4.0 * dl * dl);
        H[(size_t)a * row + b] = h;
        H[(size_t)b * row + a] = h;
      }
    }
  }
  gq_sync();
  if (lane == 0u) {
    bool okh = true;
    const uint32_t pf = d - nth;
    for (uint32_t a = 0; a < d; ++a)
      for (uint32_t b = 0; b < d; ++b) {
        const double v = H[(size_t)a * row + b];
        okh = okh && (v < __builtin_inf()) && (v > -__builtin_inf());
      }
    for (uint32_t j = 0; j < nth; ++j) {  // lower Cholesky of H_tt, in place
      double dj = H[(size_t)j * row + j];
      for (uint32_t k = 0; k < j; ++k) dj -= H[(size_t)j * row + k] * H[(size_t)j * row + k];
      okh = okh && (dj > 0.0);
      const double ljj = __builtin_sqrt(dj);
      H[(size_t)j * row + j] = ljj;
      for (uint32_t i = j + 1u; i < nth; ++i) {
        double v = H[(size_t)i * row + j];
        for (uint32_t k = 0; k < j; ++k) v -= H[(size_t)i * row + k] * H[(size_t)j * row + k];
        H[(size_t)i * row + j] = v / ljj;
      }
    }
    for (uint32_t r = 0; r < pf; ++r)  // Y = Lt^-1 H_tb, over H_tb
      for (uint32_t i = 0; i < nth; ++i) {
        double v = H[(size_t)i * row + nth + r];
        for (uint32_t k = 0; k < i; ++k) v -= H[(size_t)i * row + k] * H[(size_t)k * row + nth + r];
        H[(size_t)i * row + nth + r] = v / H[(size_t)i * row + i];
      }
    for (uint32_t r = 0; r < pf; ++r)
      for (uint32_t c = 0; c <= r; ++c) {
        double v = H[(size_t)(nth + r) * row + nth + c];
        for (uint32_t k = 0; k < nth; ++k) v -= H[(size_t)k * row + nth + r] * H[(size_t)k * row + nth + c];
        H[(size_t)(nth + r) * row + nth + c] = 0.5 * v;
      }
    pt[0] = okh ? 1.0 : 0.0;
  }
  gq_sync();
  const bool okh = pt[0] != 0.0;
  gq_sync();
  return okh;
}
