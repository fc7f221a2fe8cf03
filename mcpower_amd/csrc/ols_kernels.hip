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
#include <hip/hip_runtime.h>
#include "mcp_device.hpp"
#include "mcp_params.hpp"
#include "mcp_rows.hpp"
#include "mcp_decide.hpp"
#include "mcp_suff.hpp"
#include "mcp_ols_finish.hpp"

#pragma clang fp contract(off)

namespace mcp {

enum : int { MODE_FUSED = 0, MODE_GENERATE = 1, MODE_FIT = 2 };

// Fit the accumulated prefix of n rows and decide every test (batch.rs:956-1142).
template <int P>
__device__ __forceinline__ void fit_decide(const SimParams& prm, const Suff<P>& s, uint32_t g, uint32_t n,
                                           bool active, uint32_t out_row) {
  // ---- X'X (lower) and X'y ----
  // P <= 8: the factor gets its own registers; P = 9..12 (MAX_P_FUSED): it overwrites X'X entry by entry -- an entry of
  // X'X is read for the last time when the factor's entry at the same place is formed -- so the live set stays one
  // triangle (78 doubles at P = 12) and the kernel keeps two waves per SIMD
  constexpr bool INPLACE = P > MAX_P_WAVE;
  constexpr int MT = P > MAX_P_WAVE ? MAX_P_FUSED : MAX_T_WAVE;  // tests per fit
  double A[P][P];
  double rhs[P];
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
  // ---- Cholesky, converged = min|L_ii| / max|L_ii| >= EPS_RANK (batch.rs:40-41) ----
  double Lsep[INPLACE ? 1 : P][INPLACE ? 1 : P];
#define Lm(i, j) (*(INPLACE ? &A[i][j] : &Lsep[INPLACE ? 0 : (i)][INPLACE ? 0 : (j)]))
  double invd[P];
  bool ok = true;
  double dmin = __builtin_inf(), dmax = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int t = 0; t < j; ++t) d = __builtin_fma(-Lm(j, t), Lm(j, t), d);
    ok = ok && (d > 0.0) && (d < __builtin_inf());
    const double ljj = __builtin_sqrt(d);
    Lm(j, j) = ljj;
    invd[j] = 1.0 / ljj;
    dmin = ljj < dmin ? ljj : dmin;
    dmax = ljj > dmax ? ljj : dmax;
#pragma unroll
    for (int i = j + 1; i < P; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int t = 0; t < j; ++t) v = __builtin_fma(-Lm(i, t), Lm(j, t), v);
      Lm(i, j) = v * invd[j];
    }
  }
  const bool conv = ok && (dmin / dmax >= EPS_RANK);
  // ---- beta = (L L')^-1 X'y ----
  double zf[P], beta[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    double v = rhs[i];
#pragma unroll
    for (int t = 0; t < i; ++t) v = __builtin_fma(-Lm(i, t), zf[t], v);
    zf[i] = v * invd[i];
  }
#pragma unroll
  for (int i = P - 1; i >= 0; --i) {
    double v = zf[i];
#pragma unroll
    for (int t = i + 1; t < P; ++t) v = __builtin_fma(-Lm(t, i), beta[t], v);
    beta[i] = v * invd[i];
  }
  double bxty = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) bxty = __builtin_fma(beta[i], rhs[i], bxty);
  const double nd = (double)n;
  const double rss = s.syy - bxty;
  const double df = nd - (double)P;
  const double sigma_sq = rss / df;
  const double sst = s.syy - s.sy * s.sy / nd;

  // ---- squared statistic per test: (c'beta)^2 / (sigma^2 ||L^-1 c||^2)  (posthoc.rs:86-99) ----
  const double* crow = prm.crit + (size_t)g * prm.crit_stride;
  const uint32_t nt = prm.n_tests;
  double tsq[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    tsq[t] = __builtin_nan("");
    if ((uint32_t)t < nt) {
      const double* c = prm.contrast + t * MAX_P;
      double est = 0.0, nq = 0.0;
      double u[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        est = __builtin_fma(c[i], beta[i], est);
        double acc = c[i];
#pragma unroll
        for (int q = 0; q < i; ++q) acc = __builtin_fma(-Lm(i, q), u[q], acc);
        u[i] = acc * invd[i];
        nq = __builtin_fma(u[i], u[i], nq);
      }
      const double se_sq = sigma_sq * nq;
      if (conv && se_sq > FLOAT_NEAR_ZERO && se_sq < __builtin_inf()) tsq[t] = est * est / se_sq;
    }
  }

  // ---- decisions, per-replicate tensors, additive counters (mcp_ols_finish.hpp) ----
  ols_finish<MT>(prm, tsq, conv, rss, sst, df, (double)(P - 1), g, active, out_row);
#undef Lm
}

// tile layout of the staged pipeline: [block of 256 replicates][column: x1..x_{P-1}, y][row quad][256 lanes][4 rows]
// f32 -- a lane owns 16 contiguous bytes per (column, row quad), so generator stores and fit loads are
// dwordx4 (1 KiB per wave instruction) and a workgroup's whole design is one contiguous region.
__device__ __forceinline__ size_t tile_quad_index(uint32_t P, uint32_t n_quads, uint32_t blk, uint32_t c, uint32_t quad) {
  return ((((size_t)blk * P + c) * n_quads + quad) * BLOCK) * 4u;  // + tid * 4 + r
}

// MODE_FUSED   : generate + accumulate + fit, nothing but flags/counters leaves the chip
// MODE_GENERATE: generate and store the design tile to HBM
// MODE_FIT     : stream the tile back (coalesced across replicates) and fit
#ifndef MCP_LB_W
#define MCP_LB_W 1
#endif
// GRID1: a single sample size (the common find_power call): the grid loop disappears and the sufficient
// statistics are dead once X'X is formed, which frees their registers for the solve.
// FLAVOR 0: continuous-only design (the factor walk / categorical code is compiled out);
// FLAVOR 1: + factor dummies; FLAVOR 2: general path (per-replicate scenario design, non-normal marginals /
// residuals, per-study effects, heteroskedastic scaling) on top of FLAVOR 1.
template <int P, int MODE, bool GRID1, int FLAVOR>
__global__ __launch_bounds__(BLOCK, MCP_LB_W) void ols_sim_kernel(const SimParams* __restrict__ pp) {
  constexpr bool FACTORS = FLAVOR >= 1;
  constexpr bool SCEN = FLAVOR == 2;
  extern __shared__ float stage_lds[];
  const SimParams& prm = *pp;
  const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = local < prm.n_sims;
  // general path, fused mode: lanes take the replicates of the launch in transform-class order (scenario_kernels.hip)
  const uint32_t lsim = active ? ((SCEN && MODE == MODE_FUSED && prm.scen_perm) ? prm.scen_perm[local] : local) : (prm.n_sims - 1u);
  const uint64_t sim_id = prm.sim_begin + lsim;
  const uint64_t key = pcg_mix64(prm.base_seed, sim_id);  // rng.rs:61-69
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  const uint32_t out_row = prm.out_base + lsim;
  const uint32_t max_n = prm.max_n;
  float* st = stage_lds + threadIdx.x;
  // dense tail queue of this wave, behind the staging area
  uint32_t* tq = reinterpret_cast<uint32_t*>(stage_lds + (size_t)4 * (P + (SCEN ? 1 : 0)) * BLOCK) + (threadIdx.x >> 6) * TAIL_QUEUE;
  RowCtx<P> cx;
  if (MODE != MODE_FIT) row_ctx_init<P, FACTORS>(prm, cx);
  ScenLane<P> sl;
  if (SCEN && MODE != MODE_FIT) scen_lane_load<P>(prm, lsim, sl);

  const uint32_t n_quads = (max_n + 3u) >> 2;
  if (MODE == MODE_GENERATE) {
    for (uint32_t ctr = 0; ctr < n_quads; ++ctr) {
      stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, ctr, st, SCEN ? sl.meta : 0u, tq);
      float q[P][4];  // [column: x1..x_{P-1}, y][row of the quad]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x[P], y;
        build_row<P, FACTORS, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, 4u * ctr + r, x, y, 0.0, &sl);
#pragma unroll
        for (int c = 1; c < P; ++c) q[c - 1][r] = x[c];
        q[P - 1][r] = y;
      }
#pragma unroll
      for (int c = 0; c < P; ++c)
        *reinterpret_cast<float4*>(prm.tile + tile_quad_index(P, n_quads, blockIdx.x, c, ctr) + threadIdx.x * 4u) =
            make_float4(q[c][0], q[c][1], q[c][2], q[c][3]);
    }
    return;
  }

  Suff<P> s;
  suff_reset(s);
  uint32_t row = 0;
  const uint32_t n_grid = GRID1 ? 1u : prm.n_grid;
  for (uint32_t g = 0; g < n_grid; ++g) {
    const uint32_t n_end = prm.ns[g];
    if (MODE == MODE_FUSED) {
      while (row < n_end) {
        const uint32_t base = row & ~3u;
        if ((row & 3u) == 0u) stage_group<P, BLOCK, 0, SCEN>(prm, k0, k1, row >> 2, st, SCEN ? sl.meta : 0u, tq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t rr = base + (uint32_t)r;
          if (rr >= row && rr < n_end) {
            float x[P], y;
            build_row<P, FACTORS, BLOCK, 0, SCEN>(prm, cx, k0, k1, st, r, rr, x, y, 0.0, &sl);
            suff_add_row<P>(s, x, y);
          }
        }
        const uint32_t nxt = base + 4u;
        row = nxt < n_end ? nxt : n_end;
      }
    } else {  // MODE_FIT: dwordx4 loads, lane = replicate; whole row quads take the vector path
      while (row < n_end) {
        if ((row & 3u) == 0u && row + 4u <= n_end) {
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          f32x4 v[P];
#pragma unroll
          for (int c = 0; c < P; ++c)
            v[c] = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4*>(prm.tile + tile_quad_index(P, n_quads, blockIdx.x, c, row >> 2) + threadIdx.x * 4u));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x[P];
#pragma unroll
            for (int c = 1; c < P; ++c) x[c] = r == 0 ? v[c - 1].x : (r == 1 ? v[c - 1].y : (r == 2 ? v[c - 1].z : v[c - 1].w));
            const float y = r == 0 ? v[P - 1].x : (r == 1 ? v[P - 1].y : (r == 2 ? v[P - 1].z : v[P - 1].w));
            suff_add_row<P>(s, x, y);
          }
          row += 4u;
        } else {  // ragged grid boundary: one row
          float x[P];
#pragma unroll
          for (int c = 1; c < P; ++c)
            x[c] = prm.tile[tile_quad_index(P, n_quads, blockIdx.x, c - 1, row >> 2) + threadIdx.x * 4u + (row & 3u)];
          const float y = prm.tile[tile_quad_index(P, n_quads, blockIdx.x, P - 1, row >> 2) + threadIdx.x * 4u + (row & 3u)];
          suff_add_row<P>(s, x, y);
          row += 1u;
        }
      }
    }
    fit_decide<P>(prm, s, g, n_end, active, out_row);
  }
}

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

hipError_t launch_ols_wide(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, hipStream_t st);  // ols_wide_kernels.hip

// fused kernel, P = 9..12 (MAX_P_FUSED): the staging area passes 48 KB at P = 12
template <bool GRID1, int FLAVOR>
static hipError_t launch_fused_wide(int P, const SimParams* d_prm, uint32_t n_sims, size_t lds, hipStream_t st) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1u) / BLOCK);
#define LFW(PP)                                                                                                         \
  case PP: {                                                                                                            \
    if (lds > 48 * 1024) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)ols_sim_kernel<PP, MODE_FUSED, GRID1, FLAVOR>,                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      if (e != hipSuccess) return e;                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL((ols_sim_kernel<PP, MODE_FUSED, GRID1, FLAVOR>), grid, block, lds, st, d_prm);                   \
  } break;
  switch (P) {
    LFW(9) LFW(10) LFW(11) LFW(12)
    // P = 13 .. 16, one sample size, plain data path: two waves share 64 replicates (ols_wide_kernels.hip)
    default: return (GRID1 && FLAVOR != 2 && P >= 13 && P <= MAX_P) ? launch_ols_wide(P, d_prm, n_sims, FLAVOR, st) : hipErrorInvalidValue;
  }
#undef LFW
  return hipGetLastError();
}

template <int MODE, bool GRID1, int FLAVOR>
static hipError_t launch_sim_pgf(int P, const SimParams* d_prm, uint32_t n_sims, hipStream_t st) {
  const dim3 block(BLOCK), grid((n_sims + BLOCK - 1u) / BLOCK);
  // staging area: 4 rows x (K + 1 <= P) columns x 256 lanes of f32 (+1 scratch column on the general path;
  // none for the fit-only kernel)
  const size_t lds = MODE == MODE_FIT ? 0 : (size_t)4 * (P + (FLAVOR == 2 ? 1 : 0)) * BLOCK * sizeof(float) +
                                                (BLOCK / 64) * TAIL_QUEUE * sizeof(uint32_t);
  switch (P) {
    case 2: hipLaunchKernelGGL((ols_sim_kernel<2, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 3: hipLaunchKernelGGL((ols_sim_kernel<3, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 4: hipLaunchKernelGGL((ols_sim_kernel<4, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 5: hipLaunchKernelGGL((ols_sim_kernel<5, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 6: hipLaunchKernelGGL((ols_sim_kernel<6, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 7: hipLaunchKernelGGL((ols_sim_kernel<7, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    case 8: hipLaunchKernelGGL((ols_sim_kernel<8, MODE, GRID1, FLAVOR>), grid, block, lds, st, d_prm); break;
    default:
      if (MODE == MODE_FUSED) return launch_fused_wide<GRID1, FLAVOR>(P, d_prm, n_sims, lds, st);
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// flavor: 0 continuous-only, 1 factors, 2 general (scenario) path -- the general path has no GRID1 specialisation
template <int MODE>
static hipError_t launch_sim(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, int flavor, hipStream_t st) {
  if (MODE == MODE_FIT) flavor = 0;  // the fit kernel never builds rows
  if (flavor == 2) return launch_sim_pgf<MODE, false, 2>(P, d_prm, n_sims, st);
  if (grid1) return flavor ? launch_sim_pgf<MODE, true, 1>(P, d_prm, n_sims, st) : launch_sim_pgf<MODE, true, 0>(P, d_prm, n_sims, st);
  return flavor ? launch_sim_pgf<MODE, false, 1>(P, d_prm, n_sims, st) : launch_sim_pgf<MODE, false, 0>(P, d_prm, n_sims, st);
}

hipError_t launch_ols_fused(int P, const SimParams* d_prm, uint32_t n_sims, bool grid1, int flavor, hipStream_t st) {
  return launch_sim<MODE_FUSED>(P, d_prm, n_sims, grid1, flavor, st);
}
hipError_t launch_ols_generate(int P, const SimParams* d_prm, uint32_t n_sims, int flavor, hipStream_t st) {
  return flavor == 2 ? launch_sim_pgf<MODE_GENERATE, false, 2>(P, d_prm, n_sims, st)
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
