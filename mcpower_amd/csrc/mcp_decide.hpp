// mcp_decide.hpp -- test decisions in squared-statistic space (correction.rs:16-115, batch.rs:1089-1117).
#pragma once
#include "mcp_params.hpp"

namespace mcp {

// Holm / BH need the descending order of t^2 (correction.rs:96-115): rank_i = number of tests ordered before i
template <int T>
__device__ __forceinline__ void ranks_desc(const double (&t)[T], uint32_t n, uint32_t (&rank)[T]) {
#pragma unroll
  for (int i = 0; i < T; ++i) {
    uint32_t rk = 0;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if ((uint32_t)j < n && j != i) {
        const bool in = t[i] != t[i], jn = t[j] != t[j];
        bool before;  // j sorts before i
        if (in && jn) before = j < i;
        else if (jn) before = false;
        else if (in) before = true;
        else before = (t[j] > t[i]) || (t[j] == t[i] && j < i);
        rk += before ? 1u : 0u;
      }
    }
    rank[i] = rk;
  }
}

// uncorrected + corrected decisions for the squared statistics of one fit (NaN never passes).
// crow = [crit_unc, per-rank / per-target corrected thresholds ...]
template <int MT>
__device__ __forceinline__ void decide_tests(const double (&tsq)[MT], uint32_t nt, const double* crow, int method,
                                             uint32_t& unc_bits, uint32_t& cor_bits) {
  const double crit_unc = crow[0];
  unc_bits = 0;
  cor_bits = 0;
#pragma unroll
  for (int t = 0; t < MT; ++t)
    if ((uint32_t)t < nt && tsq[t] > crit_unc) unc_bits |= 1u << t;  // NaN compares false
  if (method == 0) {
    cor_bits = unc_bits;
  } else if (method == 1 || method == 4) {  // Bonferroni / Tukey: per-target threshold
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if ((uint32_t)t < nt && tsq[t] > crow[1 + t]) cor_bits |= 1u << t;
  } else {
    uint32_t rank[MT];
    ranks_desc<MT>(tsq, nt, rank);
    uint32_t okb = 0;  // test t clears the threshold of its own rank
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if ((uint32_t)t < nt) {
        double cr = crow[1];
#pragma unroll
        for (int k = 1; k < MT; ++k) cr = (rank[t] == (uint32_t)k && (uint32_t)k < nt) ? crow[1 + k] : cr;
        if (tsq[t] > cr) okb |= 1u << t;
      }
    }
    if (method == 2) {  // Holm step-down: pass while every better-ranked test passed
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if ((uint32_t)t < nt) {
          bool pass = true;
#pragma unroll
          for (int j = 0; j < MT; ++j)
            if ((uint32_t)j < nt && rank[j] <= rank[t] && !((okb >> j) & 1u)) pass = false;
          if (pass) cor_bits |= 1u << t;
        }
      }
    } else {  // BH step-up: everything ranked at or above the last passing rank
      int last = -1;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt && ((okb >> t) & 1u)) last = (int)rank[t] > last ? (int)rank[t] : last;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if ((uint32_t)t < nt && (int)rank[t] <= last) cor_bits |= 1u << t;
    }
  }
}

}  // namespace mcp
