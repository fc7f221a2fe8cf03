// mcp_ols_finish.hpp -- the end of an OLS fit, shared by the lane-per-replicate kernels (ols_kernels.hip) and the
// wave-pair kernel for wide designs (ols_wide_kernels.hip): decisions on the squared statistics (batch.rs:1089-1117,
// correction.rs:16-92), the overall F (batch.rs:1118-1142), the per-replicate tensors of BatchResult (spec.rs:553-600)
// and the additive counters (aggregation.rs:84-284: one atomic per wave per counter, into this wave's copy).
#pragma once
#include "mcp_params.hpp"
#include "mcp_decide.hpp"

#pragma clang fp contract(off)

namespace mcp {

// tsq: squared statistic per test (NaN = not testable); every lane of the calling wave holds one replicate
template <int MT>
__device__ __forceinline__ void ols_finish(const SimParams& prm, const double (&tsq)[MT], bool conv, double rss, double sst,
                                           double df, double dfn, uint32_t g, bool active, uint32_t out_row) {
  const double* crow = prm.crit + (size_t)g * prm.crit_stride;
  const uint32_t nt = prm.n_tests;
  // ---- decisions (batch.rs:1089-1117, correction.rs:16-92) ----
  uint32_t unc_bits = 0, cor_bits = 0;
  decide_tests(tsq, nt, crow, prm.correction, unc_bits, cor_bits);
  if (!conv) {
    unc_bits = 0;
    cor_bits = 0;
  }
  // overall F (batch.rs:1118-1142)
  bool ov = false;
  if (prm.report_overall && conv) {
    if (sst > 1e-10 && rss > 0.0 && rss < __builtin_inf() && sst < __builtin_inf()) {
      const double f = ((sst - rss) / dfn) / (rss / df);
      ov = (f < __builtin_inf()) && (f > crow[3 + nt]);
    }
  }

  // ---- per-replicate tensors (BatchResult, spec.rs:553-600) ----
  if (active) {
    const size_t cell = (size_t)out_row * prm.n_grid + g;
    if (prm.unc) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt) {
          prm.unc[cell * nt + t] = (uint8_t)((unc_bits >> t) & 1u);
          prm.cor[cell * nt + t] = (uint8_t)((cor_bits >> t) & 1u);
        }
    }
    if (prm.conv) prm.conv[cell] = conv ? 1 : 0;
    if (prm.overall) prm.overall[cell] = ov ? 1 : 0;
    if (prm.stat && g == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt) prm.stat[(size_t)out_row * nt + t] = tsq[t];
    }
  }

  // ---- additive counters (aggregation.rs:84-284): one atomic per wave per counter ----
  unsigned long long* ctr = prm.counters +
                            (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % CTR_COPIES) * prm.ctr_copy_stride +
                            (size_t)g * prm.ctr_stride;
  const bool lane0 = (threadIdx.x & 63u) == 0u;
  const uint32_t ub = active ? unc_bits : 0u, cb = active ? cor_bits : 0u;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if ((uint32_t)t < nt) {
      const unsigned long long mu = __ballot((ub >> t) & 1u);
      const unsigned long long mc = __ballot((cb >> t) & 1u);
      if (lane0) {
        if (mu) atomicAdd(ctr + prm.ctr_k_unc + t, (unsigned long long)__popcll(mu));
        if (mc) atomicAdd(ctr + prm.ctr_k_cor + t, (unsigned long long)__popcll(mc));
      }
    }
  }
  const uint32_t ku = __popc(ub), kc = __popc(cb);
#pragma unroll
  for (int k = 0; k <= MT; ++k) {
    if ((uint32_t)k <= nt) {
      const unsigned long long hu = __ballot(active && ku == (uint32_t)k);
      const unsigned long long hc = __ballot(active && kc == (uint32_t)k);
      if (lane0) {
        if (hu) atomicAdd(ctr + prm.ctr_hist_unc + k, (unsigned long long)__popcll(hu));
        if (hc) atomicAdd(ctr + prm.ctr_hist_cor + k, (unsigned long long)__popcll(hc));
      }
    }
  }
  {
    const unsigned long long mcv = __ballot(active && conv);
    const unsigned long long mov = __ballot(active && ov);
    const unsigned long long mac = __ballot(active);
    if (lane0) {
      if (mcv) atomicAdd(ctr + prm.ctr_conv, (unsigned long long)__popcll(mcv));
      if (mov) atomicAdd(ctr + prm.ctr_overall, (unsigned long long)__popcll(mov));
      if (mac) atomicAdd(ctr + prm.ctr_n_sims, (unsigned long long)__popcll(mac));
    }
  }
}

}  // namespace mcp
