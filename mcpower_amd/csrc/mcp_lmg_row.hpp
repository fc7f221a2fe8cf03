// mcp_lmg_row.hpp -- the latent-scale random effect of ONE design row under the general random-effect structure
// (random slopes on the primary grouping, up to two extra groupings crossed with / nested in it, slopes on those),
// RE-DRAWN from the sequential stream on the spot: no per-replicate tables, so a lane-per-replicate kernel can use it.
// Only the design-capture kernel (introspect::capture_sim0_data twin) does; the fit kernels draw every level once into
// LDS (lmm_general_impl.hpp, glmm_general_kernels.hip) with these same formulas.
//   draw order on the sequential stream: primary intercepts, primary slopes, each extra grouping's intercepts, each extra
//   grouping's slopes (data_gen.rs:146-284; the lowering's lmg_seq offsets); slopes are correlated with the intercept
//   through chol(diag(tau) R diag(tau)), z_0 recovered from the f32 intercept effect; a row sums, in f64, the primary
//   intercept, every extra grouping's (intercept, slopes x column), then the primary slopes (data_gen.rs:812-842).
#pragma once
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_scenario.hpp"

namespace mcp {

// extra_level_of_row (generation.rs:402-427)
__device__ __forceinline__ uint32_t lmg_row_level(const SimParams& prm, uint32_t t, uint32_t i) {
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

template <int P, bool SCEN>
__device__ __forceinline__ double lmg_row_effect_redraw(const SimParams& prm, uint32_t k0, uint32_t k1, uint32_t lsim,
                                                        uint32_t rr, const float (&x)[P]) {
  const uint32_t nterms = prm.lmg_n_terms;
  double u_re = 0.0, prim_slopes = 0.0;
  for (uint32_t t = 0; t < nterms; ++t) {
    const uint32_t q = prm.lmg_q[t], nl = prm.lmg_levels[t];
    uint32_t l = lmg_row_level(prm, t, rr);
    l = l < nl ? l : 0u;
    double tau_t = prm.lmg_tau[t];
    if (SCEN && prm.scen_tau_sq) {  // realised tau per grouping (ICC jitter, data_gen.rs:1069-1092)
      const double t2 = prm.scen_tau_sq[(size_t)t * prm.scen_stride + lsim];
      tau_t = __builtin_sqrt(t2 > 0.0 ? t2 : 0.0);
    }
    const float e0 = (float)(draw_cluster_effect(prm, k0, k1, prm.lmg_seq[2u * t] + l) * tau_t);
    u_re += (double)e0;
    if (q > 1u) {
      double lt[9];
      if (!SCEN) {
        for (uint32_t i = 0; i < 9u; ++i) lt[i] = prm.lmg_chol[t][i];
      } else {  // chol_lower (data_gen.rs:55-75) of the replicate's own D
        const double tv[3] = {tau_t, prm.lmg_slope_sd[t][0], prm.lmg_slope_sd[t][1]};
        double dm[9];
        for (uint32_t i = 0; i < 9u; ++i) lt[i] = 0.0;
        for (uint32_t i = 0; i < q; ++i)
          for (uint32_t j = 0; j < q; ++j) dm[i * q + j] = tv[i] * prm.lmg_recorr[t][i * q + j] * tv[j];
        for (uint32_t j = 0; j < q; ++j) {
          double diag = dm[j * q + j];
          for (uint32_t k = 0; k < j; ++k) diag -= lt[j * q + k] * lt[j * q + k];
          const double ljj = __builtin_sqrt(diag > 0.0 ? diag : 0.0);
          lt[j * q + j] = ljj;
          for (uint32_t i = j + 1u; i < q; ++i)
            if (ljj > 0.0) {
              double sacc = dm[i * q + j];
              for (uint32_t k = 0; k < j; ++k) sacc -= lt[i * q + k] * lt[j * q + k];
              lt[i * q + j] = sacc / ljj;
            }
        }
      }
      double z[3];
      z[0] = tau_t > 0.0 ? (double)e0 / tau_t : 0.0;
      for (uint32_t j = 1; j < q; ++j) z[j] = draw_cluster_effect(prm, k0, k1, prm.lmg_seq[2u * t + 1u] + l * (q - 1u) + (j - 1u));
      for (uint32_t k = 1; k < q; ++k) {
        double uk = 0.0;
        for (uint32_t j = 0; j <= k; ++j) uk += lt[k * q + j] * z[j];
        float xv = 0.0f;
#pragma unroll
        for (int c = 1; c < P; ++c) xv = (uint32_t)c == prm.lmg_cols[t][k - 1u] ? x[c] : xv;
        const double term = (double)(float)uk * (double)xv;
        if (t == 0u) prim_slopes += term;  // the primary grouping's slopes come last in the row's sum
        else u_re += term;
      }
    }
  }
  return u_re + prim_slopes;
}

}  // namespace mcp
