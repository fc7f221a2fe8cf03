// ols_kernels.hip -- gfx950 kernels of the OLS simulate path.
//
// Mapping: ONE LANE PER REPLICATE (64 replicates per wavefront).  A replicate's
// sufficient statistics (lower triangle of X'X, X'y, y'y) and its p x p Cholesky
// factor live in registers; the 4-row Philox block of every design column is
// staged in LDS ([slot][lane], conflict-free) so that the ~4.85 % of draws that
// need the tail branch of the inverse CDF are fixed up in a short wave-level
// loop instead of executing both branches for every draw.  No cross-lane
// reduction, no divergence on the integer/RNG path, which is what bounds this
// kernel.
//
// Reference functions restated here (crates/engine-core/src):
//   data_gen.rs:87-925 generate_sim_data (continuous block :316-363, residual
//     :412-421, factor walk :509-551, interactions :643-649, y assembly :781-857)
//   batch.rs:871-1142 OLS arm of run_one_sim (grid-incremental sufficient
//     statistics, t^2 per target, correction, overall F)
//   correction.rs:16-92 apply_correction
// The X'X accumulation and Cholesky solve replace glmm 0.1.2's
// OlsSuffStats::add_rows / fit_suff_stats_t_sq (out of tree; semantics from the
// call sites batch.rs:884-968 and posthoc.rs:62-149).
#include "ols_sim_impl.hpp"

namespace mcp {

// ---- introspect::fit_provided_data for the OLS arm: one lane per provided design ----
// design: [fit][column][row] f32 (column 0 must be the intercept), outcome: [fit][row]
template <int P>
__global__ __launch_bounds__(64) void ols_fit_provided_kernel(const float* __restrict__ design,
                                                              const float* __restrict__ outcome, uint32_t n_fits,
                                                              uint32_t n, double* __restrict__ beta_out,
                                                              double* __restrict__ se_out, double* __restrict__ tsq_out,
                                                              double* __restrict__ rss_out, double* __restrict__ sst_out,
                                                              uint8_t* __restrict__ conv_out) {
  const uint32_t fit = blockIdx.x * blockDim.x + threadIdx.x;
  if (fit >= n_fits) return;
  const float* X = design + (size_t)fit * n * P;
  const float* Y = outcome + (size_t)fit * n;
  Suff<P> s;
  suff_reset(s);
  for (uint32_t i = 0; i < n; ++i) {
    float x[P];
#pragma unroll
    for (int c = 1; c < P; ++c) x[c] = X[(size_t)c * n + i];
    suff_add_row<P>(s, x, Y[i]);
  }
  double A[P][P], rhs[P], Lm[P][P], invd[P];
  A[0][0] = (double)n;
  rhs[0] = s.sy;
  {
    int idx = 0;
#pragma unroll
    for (int a = 1; a < P; ++a) {
      A[a][0] = s.sx[a - 1];
      rhs[a] = s.sxy[a - 1];
#pragma unroll
      for (int b = 1; b <= a; ++b) A[a][b] = s.sxx[idx++];
    }
  }
  bool ok = true;
  double dmin = __builtin_inf(), dmax = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int t = 0; t < j; ++t) d = __builtin_fma(-Lm[j][t], Lm[j][t], d);
    ok = ok && (d > 0.0) && (d < __builtin_inf());
    const double ljj = __builtin_sqrt(d);
    Lm[j][j] = ljj;
    invd[j] = 1.0 / ljj;
    dmin = ljj < dmin ? ljj : dmin;
    dmax = ljj > dmax ? ljj : dmax;
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int t = 0; t < j; ++t) v = __builtin_fma(-Lm[i][t], Lm[j][t], v);
      Lm[i][j] = v * invd[j];
    }
  }
  const bool conv = ok && (dmin / dmax >= EPS_RANK);
  double zf[P], beta[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = rhs[i];
#pragma unroll
    for (int t = 0; t < i; ++t) v = __builtin_fma(-Lm[i][t], zf[t], v);
    zf[i] = v * invd[i];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = zf[i];
#pragma unroll
    for (int t = i + 1; t < P; ++t) v = __builtin_fma(-Lm[t][i], beta[t], v);
    beta[i] = v * invd[i];
  }
  double bxty = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) bxty = __builtin_fma(beta[i], rhs[i], bxty);
  const double nd = (double)n;
  const double rss = s.syy - bxty;
  const double sigma_sq = rss / (nd - (double)P);
  rss_out[fit] = rss;
  sst_out[fit] = s.syy - s.sy * s.sy / nd;
  conv_out[fit] = conv ? 1 : 0;
  // (X'X)^-1_jj = ||L^-1 e_j||^2
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double u[P], nq = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lm[i][q], u[q], acc);
      u[i] = acc * invd[i];
      nq = __builtin_fma(u[i], u[i], nq);
    }
    const double se_sq = sigma_sq * nq;
    beta_out[(size_t)fit * P + j] = conv ? beta[j] : __builtin_nan("");
    se_out[(size_t)fit * P + j] = conv ? __builtin_sqrt(se_sq) : __builtin_nan("");
    tsq_out[(size_t)fit * P + j] = (conv && se_sq > FLOAT_NEAR_ZERO) ? beta[j] * beta[j] / se_sq : __builtin_nan("");
  }
}

// ---- exhaustive check kernel for the frozen inverse-CDF: out[i] = icdf(u32_to_unit_f32(first + i * stride)) ----
__global__ void icdf_probe_kernel(uint32_t first_word, uint32_t stride, uint32_t n, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = norm_inv_cdf_f32(u32_to_unit_f32(first_word + i * stride));
}

}  // namespace mcp

// ---- launchers (C++ linkage, called from host_api.hip) ----
namespace mcp {

hipError_t launch_ols_scen(int mode, int P, const SimParams* d_prm, uint32_t n_sims, uint32_t n_grid, hipStream_t st);  // ols_scen_kernels.hip

// flavor: 0 continuous-only, 1 factors, 2 general (scenario) path -- the general path has no GRID1 specialisation
template <int MODE>
static hipError_t launch_sim(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, int flavor, hipStream_t st, uint32_t n_grid = 1u) {
  if (MODE == MODE_FIT) flavor = 0;  // the fit kernel never builds rows
  if (flavor == 2) return launch_ols_scen(MODE, P, d_prm, n_sims, n_grid, st);  // ols_scen_kernels.hip
  if (grid1) return flavor ? launch_sim_pgf<MODE, true, 1>(P, d_prm, n_sims, st) : launch_sim_pgf<MODE, true, 0>(P, d_prm, n_sims, st);
  return flavor ? launch_sim_pgf<MODE, false, 1>(P, d_prm, n_sims, st, n_grid) : launch_sim_pgf<MODE, false, 0>(P, d_prm, n_sims, st, n_grid);
}

hipError_t launch_ols_fused(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, int flavor, uint32_t n_grid, hipStream_t st) {
  return launch_sim<MODE_FUSED>(P, d_prm, n_sims, grid1, flavor, st, n_grid);
}
size_t ols_rec_fields(int P) { return (size_t)(P - 1) * P / 2 + 2u * (size_t)(P - 1) + 2u; }
hipError_t launch_ols_generate(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, hipStream_t st) {
  return flavor == 2 ? launch_ols_scen(MODE_GENERATE, P, d_prm, n_sims, 1u, st)
                     : launch_sim<MODE_GENERATE>(P, d_prm, n_sims, true, flavor, st);
}
hipError_t launch_ols_fit(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, hipStream_t st) {
  return launch_sim<MODE_FIT>(P, d_prm, n_sims, grid1, 0, st);
}

hipError_t launch_ols_fit_provided(int P, const float* design, const float* outcome, uint32_t n_fits, uint32_t n,
                                   double* beta, double* se, double* tsq, double* rss, double* sst, uint8_t* conv,
                                   hipStream_t st) {
  const dim3 block(64), grid((n_fits + 63u) / 64u);
#define LFP(PP) \
  case PP: hipLaunchKernelGGL((ols_fit_provided_kernel<PP>), grid, block, 0, st, design, outcome, n_fits, n, beta, se, tsq, rss, sst, conv); break;
  switch (P) {
    LFP(2) LFP(3) LFP(4) LFP(5) LFP(6) LFP(7) LFP(8) LFP(9) LFP(10) LFP(11) LFP(12) LFP(13) LFP(14) LFP(15) LFP(16)
    default: return hipErrorInvalidValue;
  }
#undef LFP
  return hipGetLastError();
}

hipError_t launch_icdf_probe(uint32_t first_word, uint32_t stride, uint32_t n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(icdf_probe_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, first_word, stride, n, out);
  return hipGetLastError();
}

}  // namespace mcp
