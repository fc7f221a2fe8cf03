// mcp_suff.hpp -- lane-resident OLS sufficient statistics shared by the OLS kernels
// (replaces glmm 0.1.2's OlsSuffStats::add_rows; semantics from the call sites batch.rs:884-968).
#pragma once
#include "mcp_params.hpp"

#pragma clang fp contract(off)

namespace mcp {

constexpr int BLOCK = 256;
constexpr double EPS_RANK = 1e-12;         // batch.rs:41
constexpr double FLOAT_NEAR_ZERO = 1e-30;  // lib.rs:46


template <int P>
struct Suff {
  double sxx[(P - 1) * P / 2];  // x_a.x_b for 1 <= b <= a < P
  double sx[P - 1];             // column sums (= X'X[a][0], the intercept column is 1)
  double sxy[P - 1];
  double sy, syy;
};

template <int P>
__device__ __forceinline__ void suff_reset(Suff<P>& s) {
#pragma unroll
  for (int i = 0; i < (P - 1) * P / 2; ++i) s.sxx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < P - 1; ++i) {
    s.sx[i] = 0.0;
    s.sxy[i] = 0.0;
  }
  s.sy = 0.0;
  s.syy = 0.0;
}

// one design row into the sufficient statistics (replaces glmm add_rows; f64, fma)
template <int P>
__device__ __forceinline__ void suff_add_row(Suff<P>& s, const float (&x)[P], float yf) {
  const double y = (double)yf;
  double xd[P];
#pragma unroll
  for (int a = 1; a < P; ++a) xd[a] = (double)x[a];
  int idx = 0;
#pragma unroll
  for (int a = 1; a < P; ++a) {
#pragma unroll
    for (int b = 1; b <= a; ++b) {
      s.sxx[idx] = __builtin_fma(xd[a], xd[b], s.sxx[idx]);
      ++idx;
    }
    s.sx[a - 1] += xd[a];
    s.sxy[a - 1] = __builtin_fma(xd[a], y, s.sxy[a - 1]);
  }
  s.sy += y;
  s.syy = __builtin_fma(y, y, s.syy);
}

}  // namespace mcp
