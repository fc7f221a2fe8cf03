// mcp_glm_family.hpp -- the GLM family kernel shared by the IRLS kernel (glm_kernels.hip) and the GLMM kernel
// (glmm_kernels.hip): IRLS weight, weight x working response and deviance contribution of one observation.
#pragma once
#include "mcp_device.hpp"

namespace mcp {

constexpr double W_FLOOR = 1e-10;

// ---- short f64 kernels for the logistic row: e^{-a} (a >= 0), 1 / u and log(u) (u in (1, 2]).  ocml's exp and
// log1p are correctly-rounded-grade double-double expansions (~150 instructions each here, ISA count) and made
// the IRLS row 4x the cost of its algebra; IRLS needs ~1e-15, not the last bit (the stopping rule is 1e-8 on the
// deviance).  Maximum relative error against libm: exp 2.3e-16 on [0, 700], log1p 5.6e-16 (checked in numpy
// with the same operations; tests/test_gpu_parity.py compares the fits with the oracle's glibc path). ----
// log(1 + x) for x in [0, 1] given u = 1 + x and ru = 1 / u: log(u) by the atanh series on m = u or u / 2 in
// (1/sqrt2, sqrt2], plus the rounding of 1 + x carried as (x - (u - 1)) / u
__device__ __forceinline__ double log1p_unit(double x, double u, double ru) {
  const double c = x - (u - 1.0);
  const bool big = u > 1.4142135623730951;
  const double m = big ? u * 0.5 : u;
  const double s = (m - 1.0) * fast_rcp(m + 1.0);
  const double z = s * s;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, z, 1.0 / 19.0);
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  p = __builtin_fma(p, z, 1.0);
  return __builtin_fma(2.0 * s, p, (big ? 0.6931471805599453 : 0.0)) + c * ru;
}

// IRLS weight w, the product w * z of weight and working response, and the deviance contribution of one
// observation (R glm.fit forms).  X'Wz only ever needs w * z = w * eta + (y - mu) * (w / mu'), which for the
// canonical links is w * eta + (y - mu): no division by the (possibly floored) weight.
__device__ __forceinline__ void fam_eval(uint32_t fam, double eta, double yi, double& w, double& wz, double& dev_i) {
  if (fam == 0u) {
    const double ea = exp_neg(__builtin_fabs(eta));  // e^{-|eta|} in (0, 1]
    const double u = 1.0 + ea;
    const double inv = fast_rcp(u);
    const double pr = eta >= 0.0 ? inv : ea * inv;  // stable sigmoid
    w = pr * (1.0 - pr);
    w = w < W_FLOOR ? W_FLOOR : w;
    wz = __builtin_fma(w, eta, yi - pr);
    // -2 log-likelihood of a Bernoulli observation: log p = -log1p(e^-eta), log(1-p) = -eta - log1p(e^-eta) (eta >= 0)
    const double l1p = log1p_unit(ea, u, inv);
    const double ll = yi > 0.5 ? (eta >= 0.0 ? -l1p : eta - l1p) : (eta >= 0.0 ? -eta - l1p : -l1p);
    dev_i = -2.0 * ll;
  } else if (fam == 1u) {  // probit: eta clamped to +-qnorm(eps), mu' floored at eps (R's probit link)
    const double thresh = 8.125890664701906;
    const double e = eta < -thresh ? -thresh : (eta > thresh ? thresh : eta);
    const double mu = 0.5 * erfc(-e * 0.70710678118654752440), omu = 0.5 * erfc(e * 0.70710678118654752440);
    double dmu = exp(-0.5 * e * e) * 0.39894228040143267794;
    dmu = dmu < 2.220446049250313e-16 ? 2.220446049250313e-16 : dmu;
    w = dmu * dmu / (mu * omu);
    w = w < W_FLOOR ? W_FLOOR : w;
    wz = w * (eta + (yi - mu) / dmu);
    dev_i = -2.0 * (yi > 0.5 ? log(mu) : log(omu));
  } else {  // Poisson, log link
    const double mu = exp(eta);
    w = mu < W_FLOOR ? W_FLOOR : mu;
    wz = __builtin_fma(w, eta, yi - mu);
    dev_i = 2.0 * ((yi > 0.0 ? yi * log(yi / mu) : 0.0) - (yi - mu));
  }
}

}  // namespace mcp
