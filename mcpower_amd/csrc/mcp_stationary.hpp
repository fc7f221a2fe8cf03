// mcp_stationary.hpp -- stationarity check of a simplex result, shared by the mixed-model kernels (glmm_kernels.hip,
// glmm_slopes_kernels.hip, glmm_general_kernels.hip, lmm_general_impl.hpp); it replaces their UNCONDITIONAL simplex
// restart.  A restart re-shrinks a fresh simplex from edge 0.05 to the tolerance: ~160 evaluations of the objective
// that, in 2 000 of 2 000 replicates of the reference bench's glmm_intercept, moved no decision.  Here: central
// differences of the objective along every coordinate with lme4's deriv12 step h = 1e-4 (h = x_k for a bounded
// parameter closer than that to 0; the single probe +h for one pinned AT 0); the point is accepted when no probe is
// lower than f0 by more than 1e-8 and the Newton decrease g^2 / (2 H) each coordinate predicts stays below 1e-8 -- the
// 1e-8 the restart loop calls "no progress".  2 d evaluations; a rejected point takes the restart loop as before.
// Same probes in the same order (k ascending, +h then -h) as the checker restates them (oracle/oracle_internal.h).
// Included inside namespace mcp by each kernel file, after its LDS pointer typedefs (like mcp_glmm_hessian.hpp).
#pragma once

constexpr double MCP_STAT_H = 1e-4;
constexpr double MCP_STAT_TOL = 1e-8;

// f(point) -> objective, wave-uniform, may clamp the point in place.  sx: the simplex block in LDS (row 0 = the point,
// d entries); the probe is staged in row d + 3 (the simplex's expansion row, free between runs).  lower0: bit k set =
// parameter k is bounded below by 0.  One wave per block: __syncthreads() orders the LDS traffic.
template <class PT, class F>
__device__ __forceinline__ bool mixed_stationary(F& f, PT* sx, uint32_t row, uint32_t d, uint32_t lower0, double f0, uint32_t lane) {
  const double inf = __builtin_inf();
  if (!(f0 < inf && f0 > -inf)) return false;
  PT* pt = sx + (size_t)row * (d + 3u);
  for (uint32_t k = 0; k < d; ++k) {
    __syncthreads();
    const double pk = sx[k];
    const bool bounded = (lower0 >> k) & 1u;
    double h = MCP_STAT_H;
    if (bounded && pk == 0.0) {
      if (lane < d) pt[lane] = lane == k ? h : sx[lane];
      __syncthreads();
      const double fp = f(pt);
      if (!(fp >= f0 - MCP_STAT_TOL)) return false;
      continue;
    }
    if (bounded && pk < h) h = pk;
    if (lane < d) pt[lane] = lane == k ? sx[lane] + h : sx[lane];
    __syncthreads();
    const double fp = f(pt);
    __syncthreads();
    if (lane < d) pt[lane] = lane == k ? sx[lane] - h : sx[lane];
    __syncthreads();
    const double fm = f(pt);
    if (!(fp < inf && fp > -inf && fm < inf && fm > -inf) || fp < f0 - MCP_STAT_TOL || fm < f0 - MCP_STAT_TOL) return false;
    const double g = (fp - fm) / (2.0 * h), H = (fp - 2.0 * f0 + fm) / (h * h);
    if (H > 0.0 ? !(g * g <= 2.0 * MCP_STAT_TOL * H) : !(__builtin_fabs(fp - fm) <= 2.0 * MCP_STAT_TOL)) return false;
  }
  return true;
}
